"""Thin tensor-level wrappers over the C ABI: allocate outputs with torch, launch on the current stream of the
device that owns the tensors, return device tensors.  These wrappers do no arithmetic of their own (the few
mesh-sized torch expressions of the path live in pipeline.py and are named there)."""
import os
import threading

import ctypes

import torch

from . import _hip as H


def _f(t):
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.contiguous().float()


class _Built:
    """A device buffer built once by kernels on some stream (packed filters, cached inverses, constant meshes) and then
    read by launches on any stream: remembers the stream it was built on and an event behind the build, so that a first
    use from ANOTHER stream waits for it.  Building inside a HIP-graph capture is refused (the buffer would come from the
    graph's private pool and be replayed over by later eager launches): warm the path up eagerly before capturing."""
    __slots__ = ('value', 'stream', 'event', 'tag')

    def __init__(self, build, device, what, tag=None):
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError('stabstitch2_amd: %s would be built inside a HIP-graph capture; run the path once eagerly '
                               '(warm-up) before capturing' % what)
        self.value = build()
        self.stream = torch.cuda.current_stream(device)
        self.event = torch.cuda.Event()
        self.event.record(self.stream)
        self.tag = tag

    def get(self, device):
        if self.event is None or torch.cuda.is_current_stream_capturing():
            # (inside a capture no event may be queried; the eager warm-up that built the buffer was submitted before the
            # capture began, and torch.cuda.graph synchronises the device on entry)
            return self.value
        cur = torch.cuda.current_stream(device)
        if cur != self.stream:
            if self.event.query():
                self.event = None               # finished long ago: nothing to wait for any more
            else:
                cur.wait_event(self.event)
        return self.value


# ------------------------------------------------------------------ layout
def nchw_to_nhwc(x, c_pad=None, out=None):
    n, c, h, w = x.shape
    c_pad = c_pad or ((c + 3) // 4) * 4
    if out is None:
        out = torch.empty((n, h, w, c_pad), device=x.device, dtype=torch.float32)
    assert tuple(out.shape) == (n, h, w, c_pad)
    H.call('ss_nchw_to_nhwc', H.dptr(_f(x)), H.dptr(out), n, c, h, w, c_pad, H.stream())
    return out


def nhwc_to_nchw(x, c=None):
    n, h, w, cs = x.shape
    c = c or cs
    out = torch.empty((n, c, h, w), device=x.device, dtype=torch.float32)
    H.call('ss_nhwc_to_nchw', H.dptr(x), H.dptr(out), n, c, h, w, cs, H.stream())
    return out


def adjacent(a, b):
    """Do two contiguous fp32 tensors lie back to back in memory (b right behind a)?  Then a launch over both reads them as one."""
    return (a.is_contiguous() and b.is_contiguous() and a.dtype == b.dtype == torch.float32 and a.device == b.device and
            b.data_ptr() == a.data_ptr() + a.numel() * 4)


def stem_input(xs):
    """NCHW frames (one tensor [n,3,h,w] or a list of them, laid back to back) -> [N,h,w+8,3]: the stem convolution's
    input layout (3 zero pixels left, 5 right; ss_nchw_to_nhwc3_padded)."""
    xs = xs if isinstance(xs, (list, tuple)) else [xs]
    total = sum(x.shape[0] for x in xs)
    h, w = xs[0].shape[2], xs[0].shape[3]
    buf = torch.empty((total, h, w + 8, 3), device=xs[0].device, dtype=torch.float32)
    o, i = 0, 0
    xs = [_f(x) for x in xs]
    while i < len(xs):
        x = xs[i]
        assert x.shape[1] == 3 and tuple(x.shape[2:]) == (h, w), x.shape
        n = x.shape[0]
        while i + 1 < len(xs) and adjacent(xs[i], xs[i + 1]):        # frames that lie back to back in memory: one launch
            i += 1
            assert xs[i].shape[1] == 3 and tuple(xs[i].shape[2:]) == (h, w), xs[i].shape
            n += xs[i].shape[0]
        H.call('ss_nchw_to_nhwc3_padded', H.dptr(x), H.dptr(buf[o:o + n]), n, h, w, H.stream())
        o += n
        i += 1
    return buf


def conv_stem(buf, wgt, bias=None, relu=True):
    """7x7 / stride 2 / pad 3 stem on `stem_input` frames.  wgt [cout,7,24] (layers.pack_stem3) -> [n,ho,wo,cout];
    wgt [g,cout,7,24] / bias [g,cout]: g stems reading the SAME frames in one launch -> [g,n,ho,wo,cout]."""
    n, h, wp, _ = buf.shape
    w = wp - 8
    grouped = wgt.dim() == 4
    g = wgt.shape[0] if grouped else 1
    cout = wgt.shape[-3]
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    out = torch.empty(((g, n, ho, wo, cout) if grouped else (n, ho, wo, cout)), device=buf.device, dtype=torch.float32)
    H.call('ss_conv_stem3', H.dptr(buf), H.dptr(wgt), H.dptr(bias, True), H.dptr(out), n, h, w, cout, int(relu), cout, g,
           0, wgt[0].numel() if grouped else 0, out[0].numel() if grouped else 0, H.stream())
    return out


# The whole stem (conv 7x7/2 + folded BN + ReLU + max-pool 3/2/1) in one kernel (csrc/stem.hip); SS_STEM_FUSED=0 runs the
# two-kernel path (ss_conv_stem3 + ss_maxpool_nhwc) for A/B measurements and the equivalence test.
STEM_FUSED = os.environ.get('SS_STEM_FUSED', '1') == '1'


def stem_pool_packed(wgt):
    """Filters of ss_stem_pool in its register layout, built on first use from a `layers.pack_stem3` tensor ([64,7,24], or
    [g*64,7,24] / [g,64,7,24] for g banks) and kept on it like the Winograd packs."""
    g = wgt.numel() // (64 * 168)
    assert wgt.numel() == g * 64 * 168, wgt.shape
    ent = getattr(wgt, '_stem_packed', None)
    tag = (wgt._version, wgt.data_ptr())
    if ent is None or ent.tag != tag:
        def build():
            pk = torch.empty(int(H.lib().ss_stem_pool_packed_floats(g)), device=wgt.device, dtype=torch.float32)
            H.call('ss_stem_pool_pack', H.dptr(wgt), H.dptr(pk), g, H.stream())
            return pk
        ent = _Built(build, wgt.device, 'the stem filter pack', tag)
        wgt._stem_packed = ent
    return ent.get(wgt.device), g


STEM_POOL_MAX_BYTES = 1 << 32        # bytes of row-packed input one ss_stem_pool launch can address


def stem_pool(buf, wgt, bias=None):
    """Conv2d(3,64,7,2,3) + folded BN + ReLU + MaxPool2d(3,2,1) on `stem_input` frames [n,h,w+8,3] in one kernel.
    wgt [g*64,7,24] (+ bias [g*64]): g filter banks reading the same frames -> [g,n,hp,wp,64] (each bank contiguous)."""
    n, h, wp8, _ = buf.shape
    w = wp8 - 8
    pk, g = stem_pool_packed(wgt)
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    hp, wp = (ho - 1) // 2 + 1, (wo - 1) // 2 + 1
    out = torch.empty((g, n, hp, wp, 64), device=buf.device, dtype=torch.float32)
    # one launch addresses its frames with 32-bit byte offsets (< 4 GiB of row-packed input, ss_stem_pool returns
    # SS_ERR_UNSUPPORTED beyond: ~2000 LR frames): larger batches go in several launches, each bank's slice written in place
    per = max(1, (STEM_POOL_MAX_BYTES - 1) // (h * wp8 * 12))
    for s in range(0, n, per):
        e = min(s + per, n)
        H.call('ss_stem_pool', H.dptr(buf[s:e]), H.dptr(pk), H.dptr(bias, True), H.dptr(out[0, s:e]), e - s, h, w, g,
               out[0].numel(), H.stream())
    return out


# ------------------------------------------------------------------ conv / pool / fc
# Deterministic kernel policy.  By default the conv engine picks a layer's kernel by LAUNCH SIZE as well as by
# geometry (F(4x4,3x3) needs >= 512 workgroups, F(2x2,3x3) >= 96, small launches are cut along K, FC layers with >= 16 rows run on
# the conv engine): a frame's last digits depend on how many frames share its launches (~1e-5 px between a 32-frame clip, two
# 16-frame passes and a stream).  Under the policy every choice follows the layer's GEOMETRY alone -- the Winograd rules as if the
# launch were large, no split-K (and no pool-in-reduction), FC always on the one-wave-per-neuron kernel -- so a frame's result is
# bit-identical whatever the batch: resident clip == chunked passes == streamed (tests/test_gpu_round6.py).  Price: small launches
# lose split-K's parallelism (batch-1 streaming ~1.6x slower, clips unchanged to ~1 %).
#     with ops.deterministic(): ...          or        pipeline.run_two_view(..., deterministic=True), OnlineStitcher(..., deterministic=True)
DETERMINISTIC = os.environ.get('SS_DETERMINISTIC', '0') == '1'        # the process default; `with ops.deterministic():` overrides it per THREAD
_PIN_IMAGES = 1 << 20
_policy_tls = threading.local()


def is_deterministic():
    """Is the geometry-only kernel policy in force for the calling thread (a `with ops.deterministic():` block, else the process
    default SS_DETERMINISTIC)?"""
    return getattr(_policy_tls, 'on', DETERMINISTIC)


class deterministic:
    """Context manager: the geometry-only kernel policy inside the block, for the calling thread (flag=False: leave the policy in
    force as it is)."""

    def __init__(self, flag=True):
        self.flag = bool(flag)
        self.old = None

    def __enter__(self):
        self.old = getattr(_policy_tls, 'on', None)
        if self.flag:
            _policy_tls.on = True
        return self

    def __exit__(self, *exc):
        if self.old is None:
            if hasattr(_policy_tls, 'on'):
                del _policy_tls.on
        else:
            _policy_tls.on = self.old
        return False


def conv_workspace(device, floats):
    """Split-K scratch for ONE launch, sized by ss_conv_workspace_need and taken from torch's caching allocator on the
    launch stream (stream-ordered: the block may be reused as soon as the reduce kernel behind it has been enqueued,
    and inside a HIP-graph capture it comes from the graph's own pool).  Most launches need none."""
    if floats <= 0:
        return None
    return torch.empty(int(floats), device=device, dtype=torch.float32)


def _conv_ws_need(n, t, h, w, c, cout, kt, kh, kw, stride, pt, ph, pw, groups):
    if is_deterministic():     # no workspace = no split-K (ss_conv_nhwc: "a NULL workspace disables splitting")
        return 0
    return int(H.lib().ss_conv_workspace_need(n, t, h, w, c, cout, kt, kh, kw, stride, pt, ph, pw, groups))


# Winograd F(2x2,3x3) for the stride-1 3x3 layers (csrc/wino.hip).  Host-level switch for A/B measurements and the
# parity tests; the dispatch rule itself is the library's (ss_conv_uses_winograd).
WINOGRAD = os.environ.get('SS_WINOGRAD', '1') == '1'


# Winograd F(4x4,3x3) (csrc/wino43.hip) where its 8 x 60-pixel tile blocks fit the map and the launch is deep enough:
# 'auto' = the rule below, '0' = never, '1' = wherever the kernel's geometry constraints hold (tests / A-B runs)
WINO43 = os.environ.get('SS_WINO43', 'auto')
WINO43_MIN_CIN = int(os.environ.get('SS_WINO43_MIN_CIN', '0'))      # 0 = the library's default (64)
WINO43_MIN_WGS = int(os.environ.get('SS_WINO43_MIN_WGS', '0'))      # 0 = the library's default (512); 1 pins the kernel choice
                                                                    # per layer whatever the batch (reproducible runs)


def _uses_wino43(kt, kh, kw, stride, pad, cin, cout, ho, wo, images, groups=1):
    """The library's rule (ss_conv_uses_wino43) behind the host switches: SS_WINO43 = 0 never, 1 wherever the kernel's geometry
    constraints hold (all thresholds 1), auto = the library's thresholds (or SS_WINO43_MIN_WGS / SS_WINO43_MIN_CIN)."""
    if WINO43 == '0' or not WINOGRAD or WINO_MATH != 'f32' or tuple(pad) != (0, 1, 1):
        return False
    forced = WINO43 == '1'
    return bool(H.lib().ss_conv_uses_wino43(int(kt), int(kh), int(kw), int(stride), int(cin), int(cout), int(ho), int(wo),
                                            int(images), int(groups), 1 if (forced or is_deterministic()) else WINO43_MIN_WGS,
                                            1 if forced else WINO43_MIN_CIN, 1 if forced else 0))


_WINO43_REFUSED = set()      # device indices whose first F(4x4,3x3) launch answered SS_ERR_DEVICE (no 144 KB of LDS per workgroup)


# F(4x4,3x3) with persistent workgroups (csrc/wino43.hip, conv_wino43p_kernel; bit-identical to the one-block-per-workgroup kernel).
# SS_WINO43_PERSIST=0 = the latter, for A/B runs; the knob is the library's process-wide, output-neutral ss_wino43_set_persistent.
WINO43_PERSIST = os.environ.get('SS_WINO43_PERSIST', '1') == '1'
_w43_persist_set = [None]


def _try_wino43(x, wgt, bias, res, relu, out):
    """ops.conv's use of the F(4x4,3x3) kernel.  The kernel needs 144 KB of LDS per workgroup; a device that cannot give it (not
    gfx950) makes the library answer SS_ERR_DEVICE at the first launch, nothing has been launched then: THAT device is remembered
    and its layers go to F(2x2,3x3) / the implicit GEMM from now on (other devices of the process keep the kernel).  A launch whose
    sizes the kernel cannot address (SS_ERR_UNSUPPORTED) falls through for this launch only.  -> the result, or None."""
    if x.device.index in _WINO43_REFUSED:
        return None
    if _w43_persist_set[0] != WINO43_PERSIST:
        H.lib().ss_wino43_set_persistent(int(WINO43_PERSIST))
        _w43_persist_set[0] = WINO43_PERSIST
    try:
        return conv_winograd43(x, wgt, bias, res, relu, out)
    except H.HipError as e:
        if e.code not in (-3, -4) or WINO43 == '1':          # (SS_WINO43=1 = "force it": then the error is the answer)
            raise
        if e.code == -4:
            _WINO43_REFUSED.add(x.device.index)
        return None


def _uses_winograd(kt, kh, kw, stride, pad, cin, cout, ho, wo, images):
    return bool(WINOGRAD and kt == 1 and tuple(pad) == (0, 1, 1) and
                H.lib().ss_conv_uses_winograd(int(kt), int(kh), int(kw), int(stride), int(cin), int(cout), int(ho),
                                              int(wo), _PIN_IMAGES if is_deterministic() else int(images)))


# which kernel the most recent ops.conv / ops.conv_grouped / ops.conv_winograd call launched ('wino' | 'igemm'): read by
# bench.ConvProbe for its executed-flop accounting (the dispatch is not re-derived there)
last_conv_path = None


def conv_executed_flop_ratio(kt, kh, kw, stride, cin, cout, out_shape):
    """Executed MFMA flop / direct-convolution flop of the launch the engine picks for this geometry (the library's own dispatch
    rules, as ops.conv / ops.conv_grouped apply them): 36/144 on F(4x4,3x3), 16/36 on F(2x2,3x3), else 1."""
    if len(out_shape) == 5 and kt == 1:          # grouped 2-D launch [g,n,ho,wo,c]
        groups, per_group = out_shape[0], out_shape[1]
    else:
        groups, per_group = 1, out_shape[0]
    ho, wo = out_shape[-3], out_shape[-2]
    if _uses_wino43(kt, kh, kw, stride, (0, 1, 1), cin, cout, ho, wo, per_group, groups):
        return 36.0 / 144.0
    return 16.0 / 36.0 if _uses_winograd(kt, kh, kw, stride, (0, 1, 1), cin, cout, ho, wo, per_group * groups) else 1.0


# Arithmetic of the Winograd layers' GEMMs.  'f32' (default): v_mfma_f32_32x32x2_f32.  'bf16x9' (opt-in, SS_WINO_MATH):
# every fp32 x fp32 product formed exactly from three bf16 slices per operand (nine slice products) on the bf16 matrix
# pipe, fp32 accumulation -- agrees with 'f32' to fp32 rounding, not bit for bit (csrc/wino.hip, SLICED).
WINO_MATH = os.environ.get('SS_WINO_MATH', 'f32')


def wino_packed(wgt, groups, sliced=False):
    """Transformed + packed filters of a 3x3 weight tensor ([cout,1,3,3,cin] or [g,cout,1,3,3,cin]), built on first use
    by ss_wino_pack / ss_wino_pack3 and kept on the tensor (prepared weights are rebuilt, hence re-packed, whenever a net
    is reloaded; an in-place edit of the tensor bumps its version counter and re-packs too)."""
    attr = '_wino_packed3' if sliced else '_wino_packed'
    ent = getattr(wgt, attr, None)
    tag = (wgt._version, wgt.data_ptr())
    if ent is None or ent.tag != tag:
        cout, cin = wgt.shape[-5], wgt.shape[-1]
        per = int((H.lib().ss_wino_packed3_floats if sliced else H.lib().ss_wino_packed_floats)(cout, cin))

        def build():
            pk = torch.empty((groups, per), device=wgt.device, dtype=torch.float32)
            H.call('ss_wino_pack3' if sliced else 'ss_wino_pack', H.dptr(wgt), H.dptr(pk), cout, cin, groups, H.stream())
            return pk
        ent = _Built(build, wgt.device, 'the Winograd filter pack', tag)
        setattr(wgt, attr, ent)
    return ent.get(wgt.device)


def wino43_packed(wgt, groups):
    """F(4x4,3x3) transformed + packed filters of a 3x3 weight tensor (kept on the tensor like wino_packed)."""
    ent = getattr(wgt, '_wino43_packed', None)
    tag = (wgt._version, wgt.data_ptr())
    if ent is None or ent.tag != tag:
        cout, cin = wgt.shape[-5], wgt.shape[-1]
        per = int(H.lib().ss_wino43_packed_floats(cout, cin))
        assert per > 0, (cout, cin)

        def build():
            pk = torch.empty((groups, per), device=wgt.device, dtype=torch.float32)
            H.call('ss_wino43_pack', H.dptr(wgt), H.dptr(pk), cout, cin, groups, H.stream())
            return pk
        ent = _Built(build, wgt.device, 'the F(4x4,3x3) filter pack', tag)
        setattr(wgt, '_wino43_packed', ent)
    return ent.get(wgt.device)


def conv_winograd43(x, wgt, bias=None, res=None, relu=False, out=None):
    """3x3 / stride 1 / pad 1 convolution on the fused Winograd F(4x4,3x3) kernel (csrc/wino43.hip), unconditionally.
    x nhwc [n,h,w,c] (or [g,n,h,w,c] with wgt [g,cout,1,3,3,c]); cin % 16 == 0, cout % 64 == 0."""
    grouped = wgt.dim() == 6
    g = wgt.shape[0] if grouped else 1
    cout, cin = wgt.shape[-5], wgt.shape[-1]
    assert tuple(wgt.shape[-4:-1]) == (1, 3, 3) and x.shape[-1] == cin, (wgt.shape, x.shape)
    shared = grouped and x.dim() == 4
    n, h, w = x.shape[-4], x.shape[-3], x.shape[-2]
    global last_conv_path
    last_conv_path = 'wino43'
    if out is None:
        out = torch.empty(((g, n, h, w, cout) if grouped else (n, h, w, cout)), device=x.device, dtype=torch.float32)
    pk = wino43_packed(wgt, g)
    H.call('ss_conv3x3_wino43_nhwc', H.dptr(x), H.dptr(pk), H.dptr(bias, True), H.dptr(res, True), H.dptr(out),
           n, h, w, cin, cout, int(relu), out.shape[-1], g, 0 if (shared or not grouped) else x[0].numel(),
           pk.shape[1], out[0].numel() if grouped else 0, H.stream())
    return out


POOL_FUSED = os.environ.get('SS_POOL_FUSED', '1') == '1'      # the regressors' 2x2 max-pool inside the Winograd epilogue


def conv_winograd(x, wgt, bias=None, res=None, relu=False, out=None, pool2=False):
    """3x3 / stride 1 / pad 1 convolution on the fused Winograd F(2x2,3x3) kernel, unconditionally (ops.conv applies the
    library's dispatch rule).  x nhwc [n,h,w,c] (or [g,n,h,w,c] with wgt [g,cout,1,3,3,c]: grouped launch).
    pool2: followed by MaxPool2d(2, 2) in the same kernel -> [.., h // 2, w // 2, cout] (no residual; fp32 MFMA arithmetic only)."""
    grouped = wgt.dim() == 6
    g = wgt.shape[0] if grouped else 1
    cout, cin = wgt.shape[-5], wgt.shape[-1]
    assert tuple(wgt.shape[-4:-1]) == (1, 3, 3) and x.shape[-1] == cin, (wgt.shape, x.shape)
    shared = grouped and x.dim() == 4
    n, h, w = x.shape[-4], x.shape[-3], x.shape[-2]
    sliced = WINO_MATH == 'bf16x9'
    global last_conv_path
    last_conv_path = 'wino'
    in_gs = 0 if (shared or not grouped) else x[0].numel()
    if pool2:
        assert res is None and not sliced
        if out is None:
            out = torch.empty(((g, n, h // 2, w // 2, cout) if grouped else (n, h // 2, w // 2, cout)), device=x.device,
                              dtype=torch.float32)
        pk = wino_packed(wgt, g, False)
        H.call('ss_conv3x3_wino_pool2_nhwc', H.dptr(x), H.dptr(pk), H.dptr(bias, True), H.dptr(out), n, h, w, cin, cout,
               int(relu), out.shape[-1], g, in_gs, pk.shape[1], out[0].numel() if grouped else 0, H.stream())
        return out
    if out is None:
        out = torch.empty(((g, n, h, w, cout) if grouped else (n, h, w, cout)), device=x.device, dtype=torch.float32)
    pk = wino_packed(wgt, g, sliced)
    H.call('ss_conv3x3_wino3_nhwc' if sliced else 'ss_conv3x3_wino_nhwc', H.dptr(x), H.dptr(pk), H.dptr(bias, True), H.dptr(res, True), H.dptr(out),
           n, h, w, cin, cout, int(relu), out.shape[-1], g, in_gs,
           pk.shape[1], out[0].numel() if grouped else 0, H.stream())
    return out


def pool2_is_fused(x, wgt, stride=1, pad=(0, 1, 1)):
    """Does ops.conv / ops.conv_grouped (pool2=True) run convolution + MaxPool2d(2, 2) as ONE kernel for these operands?
    (Where the Winograd kernel is dispatched; elsewhere the two are separate launches.)"""
    cout, kt, kh, kw, cin = wgt.shape[-5:]
    groups = wgt.shape[0] if wgt.dim() == 6 else 1
    n, h, w = x.shape[-4], x.shape[-3], x.shape[-2]
    return bool(POOL_FUSED and WINO_MATH != 'bf16x9' and _uses_winograd(kt, kh, kw, stride, pad, cin, cout, h, w, n * groups))


POOL_SPLITK = os.environ.get('SS_POOL_SPLITK', '1') == '1'      # the 2x2 max-pool inside the split-K reduction of small launches


def pool2_in_reduce(x, wgt, stride=1, pad=(0, 1, 1)):
    """Does a pool2 convolution of these operands that is NOT on the Winograd kernel take its pool inside the split-K reduction
    (one conv launch + one reduce launch), rather than in a max-pool launch of its own?"""
    cout, kt, kh, kw, cin = wgt.shape[-5:]
    groups = wgt.shape[0] if wgt.dim() == 6 else 1
    n, h, w, c = x.shape[-4:]
    return bool(POOL_SPLITK and kt == 1 and pad[0] == 0 and
                _conv_ws_need(n, 1, h, w, c, cout, 1, kh, kw, stride, 0, pad[1], pad[2], groups) > 0)


def _conv_pool2_splitk(x, wgt, bias, stride, pad, relu, groups):
    """conv + ReLU + MaxPool2d(2, 2) as ONE implicit-GEMM launch + its split-K reduction (which takes the pool), for launches
    that split along K; None when this launch would not (the caller then pools in a launch of its own).
    x [n,h,w,c] / [g,n,h,w,c] (or 4-D shared by g groups); wgt [cout,1,kh,kw,cin] / [g,cout,1,kh,kw,cin]."""
    if not POOL_SPLITK:
        return None
    grouped = wgt.dim() == 6
    cout, kt, kh, kw, cin = wgt.shape[-5:]
    n, h, w, c = x.shape[-4:]
    pt, ph, pw = pad
    if kt != 1 or pt != 0:
        return None
    need = _conv_ws_need(n, 1, h, w, c, cout, 1, kh, kw, stride, 0, ph, pw, groups)
    if need <= 0:
        return None
    ho = (h + 2 * ph - kh) // stride + 1
    wo = (w + 2 * pw - kw) // stride + 1
    shape = (groups, n, ho // 2, wo // 2, cout) if grouped else (n, ho // 2, wo // 2, cout)
    out = torch.empty(shape, device=x.device, dtype=torch.float32)
    ws = conv_workspace(x.device, need)
    shared = grouped and x.dim() == 4
    global last_conv_path
    last_conv_path = 'igemm'
    H.call('ss_conv_pool2_nhwc', H.dptr(x), H.dptr(wgt), H.dptr(bias, True), H.dptr(out), n, h, w, c, cout, kh, kw, stride, ph, pw,
           int(relu), cout, groups, 0 if (shared or not grouped) else x[0].numel(), wgt[0].numel() if grouped else 0,
           out[0].numel() if grouped else 0, H.dptr(ws), ws.numel(), H.stream())
    return out


def conv(x, wgt, bias=None, res=None, stride=1, pad=(0, 1, 1), relu=False, out=None, pool2=False):
    """x nhwc [n,h,w,c] or [n,t,h,w,c]; wgt [cout,kt,kh,kw,cin] (cin == x channels).
    pool2: the convolution is followed by MaxPool2d(2, 2) -- inside the Winograd kernel where that kernel runs, else as a
    second launch."""
    if pool2:
        assert x.dim() == 4 and out is None and res is None
        if pool2_is_fused(x, wgt, stride, pad):
            return conv_winograd(x, wgt, bias, None, relu, None, pool2=True)
        y = _conv_pool2_splitk(x, wgt, bias, stride, pad, relu, 1)
        return y if y is not None else maxpool(conv(x, wgt, bias, None, stride, pad, relu), 2, 2, 0)
    five = x.dim() == 5
    if five:
        n, t, h, w, c = x.shape
    else:
        n, h, w, c = x.shape
        t = 1
    cout, kt, kh, kw, cin = wgt.shape
    assert cin == c, (cin, c)
    pt, ph, pw = pad
    to = t + 2 * pt - kt + 1
    ho = (h + 2 * ph - kh) // stride + 1
    wo = (w + 2 * pw - kw) // stride + 1
    if out is None:
        shape = (n, to, ho, wo, cout) if five else (n, ho, wo, cout)
        out = torch.empty(shape, device=x.device, dtype=torch.float32)
    if not five and _uses_wino43(kt, kh, kw, stride, pad, c, cout, ho, wo, n):
        y = _try_wino43(x, wgt, bias, res, relu, out)
        if y is not None:
            return y
    if not five and _uses_winograd(kt, kh, kw, stride, pad, c, cout, ho, wo, n):
        return conv_winograd(x, wgt, bias, res, relu, out)
    global last_conv_path
    last_conv_path = 'igemm'
    ws = conv_workspace(x.device, _conv_ws_need(n, t, h, w, c, cout, kt, kh, kw, stride, pt, ph, pw, 1))
    H.call('ss_conv_nhwc', H.dptr(x), H.dptr(wgt), H.dptr(bias, True), H.dptr(res, True), H.dptr(out),
           n, t, h, w, c, cout, kt, kh, kw, stride, pt, ph, pw, int(relu), out.shape[-1],
           1, 0, 0, 0, H.dptr(ws, True), 0 if ws is None else ws.numel(), H.stream())
    return out


def conv_grouped(x, wgt, bias=None, res=None, stride=1, pad=(0, 1, 1), relu=False, pool2=False, out=None):
    """G independent convolutions of identical geometry in ONE launch: x [G,n,h,w,c], wgt [G,cout,kt,kh,kw,cin],
    bias [G,cout] | None, res [G,n,ho,wo,cout] | None -> [G,n,ho,wo,cout]; a 4-D x [n,h,w,c] is shared by all groups.  Used where the reference runs twin
    sub-networks (regressNet2 ref/tgt, the SpatialNet and TemporalNet trunks in streaming mode)."""
    g2, cout, kt, kh, kw, cin = wgt.shape
    shared = x.dim() == 4                # one input read by every group (group stride 0)
    if shared:
        n, h, w, c = x.shape
        g = g2
    else:
        g, n, h, w, c = x.shape
    assert g2 == g and cin == c and kt == 1, (wgt.shape, x.shape)
    pt, ph, pw = pad
    ho = (h + 2 * ph - kh) // stride + 1
    wo = (w + 2 * pw - kw) // stride + 1
    if pool2:      # conv + MaxPool2d(2, 2): one kernel on the Winograd path, two launches otherwise -> [g,n,ho//2,wo//2,cout]
        assert res is None and out is None
        if pool2_is_fused(x, wgt, stride, pad):
            return conv_winograd(x, wgt, bias, None, relu, None, pool2=True)
        y = _conv_pool2_splitk(x, wgt, bias, stride, pad, relu, g)
        if y is not None:
            return y
        y = conv_grouped(x, wgt, bias, None, stride, pad, relu)
        return maxpool(y.view(g * n, ho, wo, cout), 2, 2, 0).view(g, n, ho // 2, wo // 2, cout)
    if out is None:
        out = torch.empty((g, n, ho, wo, cout), device=x.device, dtype=torch.float32)
    assert tuple(out.shape) == (g, n, ho, wo, cout) and out.is_contiguous()
    if _uses_wino43(kt, kh, kw, stride, pad, c, cout, ho, wo, n, g):
        y = _try_wino43(x, wgt, bias, res, relu, out)
        if y is not None:
            return y
    if _uses_winograd(kt, kh, kw, stride, pad, c, cout, ho, wo, n * g):
        return conv_winograd(x, wgt, bias, res, relu, out)
    global last_conv_path
    last_conv_path = 'igemm'
    ws = conv_workspace(x.device, _conv_ws_need(n, 1, h, w, c, cout, kt, kh, kw, stride, pt, ph, pw, g))
    H.call('ss_conv_nhwc', H.dptr(x), H.dptr(wgt), H.dptr(bias, True), H.dptr(res, True), H.dptr(out),
           n, 1, h, w, c, cout, kt, kh, kw, stride, pt, ph, pw, int(relu), cout,
           g, 0 if shared else x[0].numel(), wgt[0].numel(), out[0].numel(), H.dptr(ws, True),
           0 if ws is None else ws.numel(), H.stream())
    return out


def maxpool(x, k, stride, pad=0, out=None):
    n, h, w, c = x.shape
    ho = (h + 2 * pad - k) // stride + 1
    wo = (w + 2 * pad - k) // stride + 1
    if out is None:
        out = torch.empty((n, ho, wo, c), device=x.device, dtype=torch.float32)
    H.call('ss_maxpool_nhwc', H.dptr(x), H.dptr(out), n, h, w, c, k, stride, pad, H.stream())
    return out


def maxpool_split(x, k, stride, pad, out0, out1):
    """Pool x [n,h,w,c]; channels [0,c/2) -> out0, [c/2,c) -> out1 (each [n,ho,wo,c/2], preallocated)."""
    n, h, w, c = x.shape
    H.call('ss_maxpool_nhwc_split', H.dptr(x), H.dptr(out0), H.dptr(out1), n, h, w, c, k, stride, pad, H.stream())
    return out0, out1


def linear(x, w, b=None, relu=False, out=None):
    m, k = x.shape
    nout = w.shape[0]
    assert w.shape[1] == k
    if out is None:
        out = torch.empty((m, nout), device=x.device, dtype=torch.float32)
    assert out.numel() == m * nout
    H.call('ss_linear', H.dptr(x), H.dptr(w), H.dptr(b, True), H.dptr(out), m, k, nout, int(relu), H.stream())
    return out


# ------------------------------------------------------------------ correlation
def linear_grouped(x, w, b=None, relu=False, outs=None):
    """G fully connected layers of identical shape in one launch: x [G,m,k], w [G,nout,k], b [G,nout] | None ->
    [G,m,nout]; outs: list of G contiguous destination tensors of m * nout floats each (any addresses) instead."""
    g, m, k = x.shape
    nout = w.shape[1]
    assert w.shape[0] == g and w.shape[2] == k and x.is_contiguous()
    res = None
    if outs is None:
        res = torch.empty((g, m, nout), device=x.device, dtype=torch.float32)
        outs = [res[i] for i in range(g)]
    assert len(outs) == g and all(o.numel() == m * nout for o in outs)
    arr = H.ptr_array(outs)
    H.call('ss_linear_grouped', H.dptr(x), x[0].numel(), H.dptr(w), H.dptr(b, True), arr, g, m, k, nout, int(relu), H.stream())
    return res


def ccl(f1, f2, scale=10.0, want_nchw=True, want_nhwc4=True):
    """f1, f2 nhwc [n,h,w,c] -> (flow NCHW [n,2,h,w] | None, flow nhwc4 [n,h,w,4] | None)."""
    n, h, w, c = f1.shape
    ws = torch.empty(int(H.lib().ss_ccl_workspace_floats(n, h, w, c)), device=f1.device, dtype=torch.float32)
    a = torch.empty((n, 2, h, w), device=f1.device, dtype=torch.float32) if want_nchw else None
    b = torch.empty((n, h, w, 4), device=f1.device, dtype=torch.float32) if want_nhwc4 else None
    H.call('ss_ccl', H.dptr(f1), H.dptr(f2), H.dptr(a, True), H.dptr(b, True), n, h, w, c, float(scale),
           H.dptr(ws), H.stream())
    return a, b


def l2norm(x):
    """F.normalize(x, p=2, dim=channels) on an nhwc tensor."""
    out = torch.empty_like(x)
    H.call('ss_l2norm_nhwc', H.dptr(x), H.dptr(out), x.numel() // x.shape[-1], x.shape[-1], H.stream())
    return out


def cost_volume(x1, x2, r, out=None, chain=0):
    """nhwc in -> nhwc [n,h,w,pad4((2r+1)^2)] (padding channels are zero).
    chain = S > 0: x1 / x2 hold the S + 1 views of a chain of S pairs ONCE; -> the 2 S volumes [first views | second views] of the
    pairs (ss_cost_volume_shifted), equal to the volumes of cat(x[:S], x[1:])."""
    n, h, w, c = x1.shape
    d = (2 * r + 1) ** 2
    cs = ((d + 3) // 4) * 4
    if chain:
        assert n == chain + 1 and tuple(x2.shape) == tuple(x1.shape)
        n = 2 * chain
    if out is None:
        out = torch.empty((n, h, w, cs), device=x1.device, dtype=torch.float32)
    assert tuple(out.shape) == (n, h, w, cs)
    if chain:
        H.call('ss_cost_volume_shifted', H.dptr(x1), H.dptr(x2), H.dptr(out), n, h, w, c, r, cs, chain, 1 - chain, H.stream())
    else:
        H.call('ss_cost_volume', H.dptr(x1), H.dptr(x2), H.dptr(out), n, h, w, c, r, cs, H.stream())
    return out


def cost_volume_bidir(x1, x2, r, out=None):
    """Both directions in one launch: nhwc in -> [2,n,h,w,pad4((2r+1)^2)] = (cost_volume(x1, x2), cost_volume(x2, x1))."""
    n, h, w, c = x1.shape
    cs = (((2 * r + 1) ** 2 + 3) // 4) * 4
    if out is None:
        out = torch.empty((2, n, h, w, cs), device=x1.device, dtype=torch.float32)
    assert tuple(out.shape) == (2, n, h, w, cs) and tuple(x2.shape) == tuple(x1.shape)
    H.call('ss_cost_volume_bidir', H.dptr(x1), H.dptr(x2), H.dptr(out), n, h, w, c, r, cs, H.stream())
    return out


# ------------------------------------------------------------------ geometry
def tensor_dlt(src, dst):
    n = src.shape[0]
    out = torch.empty((n, 3, 3), device=src.device, dtype=torch.float32)
    H.call('ss_tensor_dlt', H.dptr(_f(src)), H.dptr(_f(dst)), H.dptr(out), n, H.stream())
    return out


def spatial_decompose(offset8, img_h, img_w):
    n = offset8.shape[0]
    ab = torch.empty((2, n, 3, 3), device=offset8.device, dtype=torch.float32)      # (back to back: homo_warp_pair reads them as one)
    a, b = ab[0], ab[1]
    H.call('ss_spatial_decompose', H.dptr(offset8), H.dptr(a), H.dptr(b), n, float(img_h), float(img_w), H.stream())
    return a, b


def spatial_meshes(offset8, off_ref, off_tgt, img_h, img_w, out=None):
    n = offset8.shape[0]
    if out is not None:
        m1, m2 = out
        assert m1.numel() == n * 126 and m2.numel() == n * 126
    else:
        m1 = torch.empty((n, 7, 9, 2), device=offset8.device, dtype=torch.float32)
        m2 = torch.empty((n, 7, 9, 2), device=offset8.device, dtype=torch.float32)
    H.call('ss_spatial_meshes', H.dptr(offset8), H.dptr(off_ref), H.dptr(off_tgt), H.dptr(m1), H.dptr(m2), n,
           float(img_h), float(img_w), H.stream())
    return m1, m2


def homo_warp_nhwc(x, theta, out_h, out_w):
    n, h, w, c = x.shape
    out = torch.empty((n, out_h, out_w, c), device=x.device, dtype=torch.float32)
    H.call('ss_homo_warp_nhwc', H.dptr(x), H.dptr(theta), H.dptr(out), n, h, w, c, out_h, out_w, H.stream())
    return out


def homo_warp_pair(x1, x2, th1, th2, out_h, out_w):
    """(homo_warp_nhwc(x1, th1), homo_warp_nhwc(x2, th2)) -- as ONE launch when both the maps and the transforms lie back to back in
    memory (the two views' halves of a trunk output, spatial_decompose's pair), else two."""
    if adjacent(x1, x2) and adjacent(th1, th2) and x1.shape[1:] == x2.shape[1:]:
        n1, n2 = x1.shape[0], x2.shape[0]
        out = torch.empty((n1 + n2, out_h, out_w, x1.shape[3]), device=x1.device, dtype=torch.float32)
        H.call('ss_homo_warp_nhwc', H.dptr(x1), H.dptr(th1), H.dptr(out), n1 + n2, x1.shape[1], x1.shape[2], x1.shape[3], out_h, out_w,
               H.stream())
        return out[:n1], out[n1:]
    if (adjacent(th1, th2) and x1.shape == x2.shape and x1.is_contiguous() and x2.is_contiguous() and x1.device == x2.device):
        # views of ONE tensor a whole number of images apart (a chain of pairs: views [0:n] and [1:n+1]): still one launch
        img = x1[0].numel() * 4
        d = x2.data_ptr() - x1.data_ptr()
        if 0 <= d <= x1.numel() * 4 and d % img == 0 and x1.untyped_storage().data_ptr() == x2.untyped_storage().data_ptr():
            n = x1.shape[0]
            out = torch.empty((2 * n, out_h, out_w, x1.shape[3]), device=x1.device, dtype=torch.float32)
            H.call('ss_homo_warp_pair_nhwc', H.dptr(x1), H.dptr(x2), H.dptr(th1), H.dptr(out), n, x1.shape[1], x1.shape[2], x1.shape[3],
                   out_h, out_w, H.stream())
            return out[:n], out[n:]
    return homo_warp_nhwc(x1, th1, out_h, out_w), homo_warp_nhwc(x2, th2, out_h, out_w)


def homo_warp_nchw(x, theta, out_h, out_w):
    n, c, h, w = x.shape
    out = torch.empty((n, c, out_h, out_w), device=x.device, dtype=torch.float32)
    H.call('ss_homo_warp_nchw', H.dptr(x), H.dptr(theta), H.dptr(out), n, c, h, w, out_h, out_w, H.stream())
    return out


def tps_solve(source, target):
    n = source.shape[0]
    T = torch.empty((n, 2, 66), device=source.device, dtype=torch.float32)
    H.call('ss_tps_solve', H.dptr(_f(source)), H.dptr(_f(target)), H.dptr(T), n, H.stream())
    return T


def tps_solve_shared(source, target):
    """n control-point sets [n,63,2] against ONE target [63,2] / [1,63,2] -> T [n,2,66] (no broadcast copy)."""
    n = source.shape[0]
    assert target.numel() == 126
    T = torch.empty((n, 2, 66), device=source.device, dtype=torch.float32)
    H.call('ss_tps_solve_shared_target', H.dptr(_f(source)), H.dptr(_f(target)), H.dptr(T), n, H.stream())
    return T


def tps_points(point, source, T):
    n, q, _ = point.shape
    out = torch.empty((n, q, 2), device=point.device, dtype=torch.float32)
    H.call('ss_tps_points', H.dptr(_f(point)), H.dptr(_f(source)), H.dptr(T), H.dptr(out), n, q, H.stream())
    return out


_rigid_winv = {}
RIGID_INVERSE_CACHE = True      # host switch for A/B runs and the equivalence test


def rigid_winv(img_h, img_w, device):
    """fp64 W^-1 of the TPS system whose control points are the normalised RIGID mesh of an (img_h, img_w) image: a
    constant of the tsmotion composition, computed once per (size, device) by ss_tps_inverse."""
    device = torch.device(device)
    key = (int(img_h), int(img_w), str(device))
    ent = _rigid_winv.get(key)
    if ent is None:
        def build():
            xs = torch.linspace(0.0, float(img_w), 9)
            ys = torch.linspace(0.0, float(img_h), 7)
            m = torch.stack((xs.view(1, -1).expand(7, -1), ys.view(-1, 1).expand(-1, 9)), 2).reshape(63, 2)
            src = torch.stack((m[:, 0] * 2. / float(img_w) - 1., m[:, 1] * 2. / float(img_h) - 1.), 1).contiguous().to(device)
            w = torch.empty((66, 66), device=device, dtype=torch.float64)
            H.call('ss_tps_inverse', H.dptr(src), H.dptr(w, dtype=torch.float64), H.stream())
            return w
        ent = _Built(build, device, 'the cached rigid-mesh TPS inverse')
        _rigid_winv[key] = ent
    return ent.get(device)


def tsmotion(smotion, tmotion, img_h=360, img_w=480, out=None, lag=1):
    """smotion, tmotion [n,7,9,2] -> (smesh, tsmotion) [n,7,9,2] (out: the two preallocated result tensors).
    lag: frame k pairs with frame k - lag (S interleaved streams advancing together: lag = S)."""
    n = smotion.shape[0]
    ws = torch.empty(int(H.lib().ss_tsmotion_workspace_floats(n)), device=smotion.device, dtype=torch.float32)
    if out is None:
        smesh = torch.empty((n, 7, 9, 2), device=smotion.device, dtype=torch.float32)
        tsm = torch.empty((n, 7, 9, 2), device=smotion.device, dtype=torch.float32)
    else:
        smesh, tsm = out
        assert smesh.numel() == n * 126 and tsm.numel() == n * 126 and smesh.is_contiguous() and tsm.is_contiguous()
    winv = rigid_winv(img_h, img_w, smotion.device) if RIGID_INVERSE_CACHE else None
    H.call('ss_tsmotion_lag', H.dptr(_f(smotion)), H.dptr(_f(tmotion)), H.dptr(smesh), H.dptr(tsm), n, int(lag), float(img_h),
           float(img_w), H.dptr(winv, True, dtype=torch.float64), H.dptr(ws), H.stream())
    return smesh, tsm


def window_push(ring, src, src_off, state=None, blocks=0, block=0, stride=0, delta=0, per=1):
    """Streaming mode: ring [R,W,...] (contiguous, fixed address) drops slot 0 of every ring and appends the row at
    src.flatten()[src_off[r]:][:E]; optionally moves `blocks` blocks of `block` floats inside `state` (block b at b * stride
    <- the floats `delta` further) in the same launch (ss_window_push).
    per > 1: ring [G, per, W, ...] -- G ring kinds x `per` streams, ring (g, j) takes src.flatten()[src_off[g] + j * E:][:E]."""
    import ctypes
    if per > 1:
        r, w = ring.shape[0], ring.shape[2]
        e = ring[0, 0, 0].numel()
        assert ring.shape[1] == per
    else:
        r, w = ring.shape[0], ring.shape[1]
        e = ring[0, 0].numel()
    assert ring.is_contiguous() and src.is_contiguous() and len(src_off) == r
    assert all(0 <= o and o + per * e <= src.numel() for o in src_off)
    if blocks:
        assert state.is_contiguous() and (blocks - 1) * stride + delta + block <= state.numel()
        assert delta >= block and (blocks == 1 or stride >= delta + block), 'window_push: overlapping state blocks'
    offs = (ctypes.c_longlong * r)(*src_off)
    H.call('ss_window_push_groups', H.dptr(ring), H.dptr(src), ctypes.cast(offs, ctypes.c_void_p), r, per, w, e,
           H.dptr(state, True), blocks, block, stride, delta, H.stream())
    return ring


# ------------------------------------------------------------------ render
MODES = {'NORMAL': 0, 'FAST': 1}
# Opt-in (SS_RENDER_EPS_FOLD=1): the fused AVERAGE renders fold the reference's + 1e-6 into their row table
# (SS_WARP_EPS_FOLD): not the reference's arithmetic (~1e-3 px), never the default.
RENDER_EPS_FOLD = os.environ.get('SS_RENDER_EPS_FOLD', '0') == '1'


def _avg_mode(mode):
    return MODES[mode] | (16 if RENDER_EPS_FOLD else 0)


def tps_warp(U, source, T, hc, wc, mode='NORMAL', with_mask=False):
    b, c, h, w = U.shape
    out = torch.empty((b, c + int(with_mask), hc, wc), device=U.device, dtype=torch.float32)
    H.call('ss_tps_warp_mask_nchw' if with_mask else 'ss_tps_warp_nchw', H.dptr(_f(U)), H.dptr(_f(source)),
           H.dptr(T), H.dptr(out), b, c, h, w, hc, wc, MODES[mode], H.stream())
    return out


def render_footprints(source, T, h, w, hc, wc, watch=None):
    """source [n,V,63,2], T [n,V,2,66], images h x w -> footprints [n, ss_render_footprint_floats] of every (frame, view) on
    the hc x wc canvas (tile-corner sampling coordinates, mesh hulls, tile order), two small launches for the whole clip.
    watch = (guard, watch_i [n,4], watch_f [n,4]): the streaming overflow watcher (ops.canvas_watch on `source`, frame = stream) inside
    the same launches."""
    n, v = source.shape[0], source.shape[1]
    per = int(H.lib().ss_render_footprint_floats(v, hc, wc))
    fp = torch.empty((n, per), device=source.device, dtype=torch.float32)
    if watch is None:
        H.call('ss_render_footprints', H.dptr(_f(source)), H.dptr(_f(T)), H.dptr(fp), n, v, h, w, hc, wc, H.stream())
    else:
        guard, wi, wf = watch
        assert tuple(wi.shape) == (n, 4) and tuple(wf.shape) == (n, 4)
        H.call('ss_render_footprints_watch', H.dptr(_f(source)), H.dptr(_f(T)), H.dptr(fp), n, v, h, w, hc, wc, float(guard),
               H.dptr(wi, dtype=torch.int32), H.dptr(wf), H.stream())
    return fp


def _fp_args(footprint):
    return H.dptr(footprint, True), (0 if footprint is None else footprint.numel())


def render_average(imgs, source, T, hc, wc, mode='NORMAL', out=None, footprint=None):
    """imgs: list of 2|3 device tensors [1,3,h,w] / [3,h,w]; source [V,63,2]; T [V,2,66] -> [3,hc,wc].
    footprint: this frame's row of `render_footprints` (views that cannot reach a tile are skipped there and count as
    exactly 0 -- a deliberate deviation from the reference, whose clamped sampler returns a rounding residue of up to
    ~1e-2 grey levels outside a view's image; the skip test samples each 64 x 8 tile at six points and is not a proof),
    or None (every view evaluated everywhere: the reference's arithmetic at every pixel)."""
    v = len(imgs)
    imgs = [_f(i) for i in imgs]
    h, w = imgs[0].shape[-2:]
    arr = H.ptr_array(imgs)
    if out is None:
        out = torch.empty((3, hc, wc), device=imgs[0].device, dtype=torch.float32)
    fp, fpn = _fp_args(footprint)
    H.call('ss_render_average', arr, H.dptr(_f(source)), H.dptr(T), fp, fpn, H.dptr(out), v, h, w, hc, wc,
           _avg_mode(mode), H.stream())
    return out


def render_average_u8(frames, source, T, hc, wc, mode='NORMAL', out=None, footprint=None):
    """The fused AVERAGE render straight from decoded uint8 frames to the uint8 video frame: frames = list of 2|3 device
    tensors [h,w,3] uint8 (cv2 channel order); -> uint8 [hc,wc,3] = `.astype(np.uint8)` of the fused values.  Equal,
    bit for bit, to ingest (uint8 -> fp32 planes) + render_average + canvas_to_u8; the fp32 planes and canvas are never
    written."""
    v = len(frames)
    h, w = frames[0].shape[0], frames[0].shape[1]
    arr = H.ptr_array(frames, dtype=torch.uint8)
    if out is None:
        out = torch.empty((hc, wc, 3), device=frames[0].device, dtype=torch.uint8)
    fp, fpn = _fp_args(footprint)
    H.call('ss_render_average_u8', arr, H.dptr(_f(source)), H.dptr(T), fp, fpn, _u8ptr(out), v, h, w, hc, wc,
           _avg_mode(mode), H.stream())
    return out


def render_average_clip(views, source, T, hc, wc, mode='NORMAL', out=None, footprint=None):
    """A whole clip in one launch: views = list of 2|3 contiguous device tensors [n,3,h,w]; source [n,V,63,2];
    T [n,V,2,66]; footprint [n, ss_render_footprint_floats] | None -> [n,3,hc,wc].  Bit-identical to n render_average calls."""
    v = len(views)
    n, _, h, w = views[0].shape
    assert all(tuple(t.shape) == (n, 3, h, w) for t in views) and source.shape[0] == n and T.shape[0] == n
    arr = H.ptr_array(views)
    if out is None:
        out = torch.empty((n, 3, hc, wc), device=views[0].device, dtype=torch.float32)
    assert tuple(out.shape) == (n, 3, hc, wc)
    fp, fpn = H.dptr(footprint, True), (0 if footprint is None else footprint.shape[-1])
    H.call('ss_render_average_clip', arr, H.dptr(_f(source)), H.dptr(T), fp, fpn, H.dptr(out), n, v, h, w, hc, wc,
           _avg_mode(mode), H.stream())
    return out


def render_average_clip_u8(views, source, T, hc, wc, mode='NORMAL', out=None, footprint=None):
    """The same from decoded uint8 clips: views = list of 2|3 contiguous device tensors [n,h,w,3] uint8 -> uint8 [n,hc,wc,3]."""
    v = len(views)
    n, h, w, _ = views[0].shape
    assert all(tuple(t.shape) == (n, h, w, 3) for t in views) and source.shape[0] == n and T.shape[0] == n
    arr = H.ptr_array(views, dtype=torch.uint8)
    if out is None:
        out = torch.empty((n, hc, wc, 3), device=views[0].device, dtype=torch.uint8)
    assert tuple(out.shape) == (n, hc, wc, 3)
    fp, fpn = H.dptr(footprint, True), (0 if footprint is None else footprint.shape[-1])
    H.call('ss_render_average_clip_u8', arr, H.dptr(_f(source)), H.dptr(T), fp, fpn, _u8ptr(out), n, v, h, w, hc, wc,
           _avg_mode(mode), H.stream())
    return out


def tps_warp_views(imgs, source, T, hc, wc, mode='NORMAL'):
    """imgs: list of V <= 3 device tensors [1,3,h,w] / [3,h,w] -> [V,4,hc,wc] (3 colour planes + ones-mask plane)."""
    v = len(imgs)
    imgs = [_f(i) for i in imgs]
    h, w = imgs[0].shape[-2:]
    arr = H.ptr_array(imgs)
    out = torch.empty((v, 4, hc, wc), device=imgs[0].device, dtype=torch.float32)
    H.call('ss_tps_warp_views', arr, H.dptr(_f(source)), H.dptr(T), H.dptr(out), v, h, w, hc, wc, MODES[mode],
           H.stream())
    return out


def add_mul(x, add, mul):
    out = torch.empty_like(x)
    H.call('ss_add_mul', H.dptr(x), H.dptr(out), float(add), float(mul), x.numel(), H.stream())
    return out


def mask_union(a, b):
    out = torch.empty_like(a)
    H.call('ss_mask_union', H.dptr(a), H.dptr(b), H.dptr(out), a.numel(), H.stream())
    return out


def _u8ptr(t):
    return H.dptr(t, dtype=torch.uint8)


def ingest_u8(frames, lr_h=360, lr_w=480, want_hr=True, hr_out=None, lr_out=None):
    """frames uint8 [n,h,w,3] (decoded, cv2 channel order) -> (hr [n,3,h,w] in 0..255 | None,
    lr [n,3,lr_h,lr_w] = cv2.resize(...)/127.5-1).  Replaces test_online_tra.py:252-278."""
    n, h, w, c = frames.shape
    if c != 3:
        raise ValueError('frames must be [n,h,w,3] uint8')
    dev = frames.device
    hr = None
    if want_hr:
        hr = hr_out if hr_out is not None else torch.empty((n, 3, h, w), device=dev, dtype=torch.float32)
    lr = lr_out if lr_out is not None else torch.empty((n, 3, lr_h, lr_w), device=dev, dtype=torch.float32)
    H.call('ss_ingest_u8', _u8ptr(frames), H.dptr(hr, allow_none=True), H.dptr(lr), n, h, w, lr_h, lr_w, H.stream())
    return hr, lr


def canvas_to_u8(canvas, out=None):
    """fp32 [n,3,h,w] -> uint8 [n,h,w,3] with `.astype(np.uint8)` semantics (test_online_tra.py:413)."""
    n, c, h, w = canvas.shape
    if c != 3:
        raise ValueError('canvas must be [n,3,h,w]')
    if out is None:
        out = torch.empty((n, h, w, 3), device=canvas.device, dtype=torch.uint8)
    H.call('ss_canvas_to_u8', H.dptr(canvas), _u8ptr(out), n, h, w, H.stream())
    return out


def linear_blend(ref, tgt, ref_m, tgt_m, want_mask=False, out=None):
    """ref,tgt [3,hc,wc]; ref_m,tgt_m [hc,wc] -> fused [3,hc,wc] (or mask1 [hc,wc])."""
    hc, wc = ref_m.shape[-2:]
    ws = torch.empty(int(H.lib().ss_linear_blend_workspace_floats(hc, wc)), device=ref_m.device, dtype=torch.float32)
    if not want_mask and out is None:
        out = torch.empty((3, hc, wc), device=ref_m.device, dtype=torch.float32)
    if want_mask:
        out = None
    mk = torch.empty((hc, wc), device=ref_m.device, dtype=torch.float32) if want_mask else None
    H.call('ss_linear_blend', H.dptr(ref, True), H.dptr(tgt, True), H.dptr(_f(ref_m)), H.dptr(_f(tgt_m)),
           H.dptr(out, True), H.dptr(mk, True), hc, wc, H.dptr(ws), H.stream())
    return mk if want_mask else out


def render_linear_clip(views, source, T, hc, wc, mode='NORMAL', out=None, want_masks=False):
    """LINEAR fusion of a whole clip in three launches (four with three views): views = list of 2|3 contiguous device tensors
    [n,3,h,w] fp32 -> [n,3,hc,wc]; or [n,h,w,3] uint8 -> the uint8 video frames [n,hc,wc,3].  source [n,V,63,2]; T [n,V,2,66].
    want_masks: also return the blender's mask1 of every pass [n,V-1,hc,wc].  Bit-identical to the per-frame chain
    tps_warp_views + linear_blend (+ mask_union + linear_blend)."""
    v = len(views)
    u8 = views[0].dtype == torch.uint8
    if u8:
        n, h, w, _ = views[0].shape
        assert all(tuple(t.shape) == (n, h, w, 3) for t in views)
    else:
        n, _, h, w = views[0].shape
        assert all(tuple(t.shape) == (n, 3, h, w) for t in views)
    assert source.shape[0] == n and T.shape[0] == n
    dev = views[0].device
    arr = H.ptr_array(views, dtype=torch.uint8 if u8 else torch.float32)
    shape = (n, hc, wc, 3) if u8 else (n, 3, hc, wc)
    if out is None:
        out = torch.empty(shape, device=dev, dtype=views[0].dtype)
    assert tuple(out.shape) == shape and out.dtype == views[0].dtype
    ws = torch.empty(int(H.lib().ss_linear_clip_workspace_floats(n, v, hc, wc)), device=dev, dtype=torch.float32)
    mk = torch.empty((n, v - 1, hc, wc), device=dev, dtype=torch.float32) if want_masks else None
    H.call('ss_render_linear_clip_u8' if u8 else 'ss_render_linear_clip', arr, H.dptr(_f(source)), H.dptr(T),
           _u8ptr(out) if u8 else H.dptr(out), H.dptr(mk, True), n, v, h, w, hc, wc, MODES[mode], H.dptr(ws), H.stream())
    return (out, mk) if want_masks else out


def mesh_bbox(meshes, img_h, img_w, bbox=None):
    """meshes: list of LR-scale tensors [...,2] -> device tensor [4] = wmin, wmax, hmin, hmax (HR px).
    bbox: an existing box to fold these meshes into (in place)."""
    acc = bbox is not None
    if bbox is None:
        bbox = torch.empty(4, device=meshes[0].device, dtype=torch.float32)
    for i, m in enumerate(meshes):
        m = _f(m)
        H.call('ss_mesh_bbox', H.dptr(m), m.numel() // 2, float(img_h), float(img_w), H.dptr(bbox), int(acc or i > 0),
               H.stream())
    return bbox


def mesh_normalize(mesh, bbox, img_h, img_w):
    """LR-scale mesh [...,2] -> canvas-normalised [N,63,2]."""
    m = _f(mesh)
    out = torch.empty((m.numel() // 126, 63, 2), device=m.device, dtype=torch.float32)
    H.call('ss_mesh_normalize', H.dptr(m), H.dptr(bbox), H.dptr(out), m.numel() // 2, float(img_h), float(img_w),
           H.stream())
    return out


def mesh_normalize_views(meshes, bbox, img_h, img_w):
    """meshes: V tensors [..., N,7,9,2] (N frames each) -> the render's control points [N,V,63,2] on the canvas `bbox`."""
    v = len(meshes)
    n = meshes[0].numel() // 126
    out = torch.empty((n, v, 63, 2), device=meshes[0].device, dtype=torch.float32)
    for k, m in enumerate(meshes):
        m = _f(m)
        assert m.numel() == n * 126
        H.call('ss_mesh_normalize_views', H.dptr(m), H.dptr(bbox), H.dptr(out), n, k, v, float(img_h), float(img_w), H.stream())
    return out


def canvas_watch_state(streams, device):
    """Fresh (watch_i [S,4] int32, watch_f [S,4] fp32) for ops.canvas_watch."""
    wi = torch.tensor([[0, 0, -1, 0]] * streams, dtype=torch.int32, device=device)
    inf = float('inf')
    wf = torch.tensor([[inf, -inf, inf, -inf]] * streams, dtype=torch.float32, device=device)
    return wi, wf


def canvas_watch(src, watch_i, watch_f, guard):
    """src [S,V,63,2] canvas-normalised control points of one push -> updates the streams' overflow state (ss_canvas_watch)."""
    s, v = src.shape[0], src.shape[1]
    assert src.is_contiguous() and tuple(watch_i.shape) == (s, 4) and tuple(watch_f.shape) == (s, 4)
    H.call('ss_canvas_watch', H.dptr(src), s, v, float(guard), H.dptr(watch_i, dtype=torch.int32), H.dptr(watch_f), H.stream())


def stream_normalize_watch(meshes, frame_stride, bboxes, img_h, img_w, guard=0.0, watch_i=None, watch_f=None):
    """One push's render control points of all views + the overflow watcher in ONE launch: meshes = V tensors whose stream s starts
    `frame_stride` floats after stream s - 1; bboxes [4] (one canvas) or [S,4] (a canvas per stream) -> [S,V,63,2]; equal, bit for
    bit, to mesh_normalize_views(_boxes) (+ canvas_watch when the watcher state is given)."""
    v = len(meshes)
    s = 1 if bboxes.dim() == 1 else bboxes.shape[0]
    for m in meshes:
        assert m.is_contiguous() and m.dtype == torch.float32
    out = torch.empty((s, v, 63, 2), device=bboxes.device, dtype=torch.float32)
    arr = H.ptr_array(list(meshes))
    H.call('ss_stream_normalize_watch', arr, v, int(frame_stride), H.dptr(bboxes), 0 if bboxes.dim() == 1 else 4, H.dptr(out), s,
           float(img_h), float(img_w), float(guard), H.dptr(watch_i, True, dtype=torch.int32), H.dptr(watch_f, True), H.stream())
    return out


def stream_splines(meshes, frame_stride, bboxes, nrigid, img_h, img_w):
    """One push's render control points AND their splines in ONE launch (ss_stream_splines): arguments as stream_normalize_watch
    -> (src [S,V,63,2], T [S,V,2,66]), equal bit for bit to stream_normalize_watch + tps_solve_shared.  The overflow watcher is not
    touched: render_footprints(..., watch=...) or canvas_watch(src, ...)."""
    v = len(meshes)
    s = 1 if bboxes.dim() == 1 else bboxes.shape[0]
    for m in meshes:
        assert m.is_contiguous() and m.dtype == torch.float32
    assert nrigid.numel() == 126 and nrigid.is_contiguous()
    src = torch.empty((s, v, 63, 2), device=bboxes.device, dtype=torch.float32)
    T = torch.empty((s, v, 2, 66), device=bboxes.device, dtype=torch.float32)
    arr = H.ptr_array(list(meshes))
    H.call('ss_stream_splines', arr, v, int(frame_stride), H.dptr(bboxes), 0 if bboxes.dim() == 1 else 4, H.dptr(nrigid), H.dptr(src),
           H.dptr(T), s, float(img_h), float(img_w), H.stream())
    return src, T


def mesh_normalize_views_boxes(meshes, frame_stride, bboxes, img_h, img_w):
    """Per-frame canvases: meshes = list of V tensors whose frame f starts `frame_stride` floats after frame f - 1 (the tensor
    handed in starts at frame 0's mesh); bboxes [n,4] -> [n,V,63,2]."""
    n, v = bboxes.shape[0], len(meshes)
    out = torch.empty((n, v, 63, 2), device=bboxes.device, dtype=torch.float32)
    for k, m in enumerate(meshes):
        assert m.is_contiguous() and m.dtype == torch.float32
        H.call('ss_mesh_normalize_views_boxes', H.dptr(m), int(frame_stride), H.dptr(bboxes), H.dptr(out), n, k, v, float(img_h),
               float(img_w), H.stream())
    return out


def h2mesh(Hm, mesh):
    """H [n,3,3], mesh [n,...,2] -> persp_divide(H^-1 [x y 1]^T), same shape as mesh (spatial_network.py:20-36)."""
    n = Hm.shape[0]
    m = _f(mesh)
    out = torch.empty_like(m)
    H.call('ss_h2mesh', H.dptr(_f(Hm)), H.dptr(m), H.dptr(out), n, m.numel() // (2 * n), H.stream())
    return out


def three_view_align(w12_m1, w12_m2, w23_m1, w23_m2, img_h, img_w):
    """Four LR meshes [1,N,7,9,2] -> (a1, a2, b1, b2, mid) [1,N,7,9,2] in HR pixels: pair (2,3) shifted by the per-frame mean
    offset, middle plane (threeview:345-380)."""
    n = w12_m1.numel() // 126
    outs = [torch.empty((1, n, 7, 9, 2), device=w12_m1.device, dtype=torch.float32) for _ in range(5)]
    H.call('ss_three_view_align', H.dptr(_f(w12_m1)), H.dptr(_f(w12_m2)), H.dptr(_f(w23_m1)), H.dptr(_f(w23_m2)),
           *[H.dptr(o) for o in outs], n, float(img_h), float(img_w), H.stream())
    return outs


def three_view_normalize(a1, a2, b1, b2, mid, bbox):
    """The five aligned meshes [1,N,7,9,2] (HR pixels) normalised on the first canvas in one launch -> [6, N, 63, 2] =
    {a1, b2 | a2, b1 | mid, mid}: points [0:2], sources [2:4], targets [4:6] of the two re-projections' batched TPS solve."""
    n = mid.numel() // 126
    out = torch.empty((6, n, 63, 2), device=mid.device, dtype=torch.float32)
    H.call('ss_three_view_normalize', H.dptr(_f(a1)), H.dptr(_f(a2)), H.dptr(_f(b1)), H.dptr(_f(b2)), H.dptr(_f(mid)), H.dptr(bbox),
           H.dptr(out), n * 63, H.stream())
    return out


def three_view_finish(n1, n3, mid, bbox):
    """Re-projected outer meshes n1 / n3 [N,63,2] (normalised on the first canvas `bbox`) and the untranslated middle mesh
    -> (mesh1, middle, mesh3) [1,N,7,9,2] in first-canvas pixels."""
    n = mid.numel() // 126
    outs = [torch.empty((1, n, 7, 9, 2), device=mid.device, dtype=torch.float32) for _ in range(3)]
    H.call('ss_three_view_finish', H.dptr(_f(n1)), H.dptr(_f(n3)), H.dptr(_f(mid)), H.dptr(bbox), *[H.dptr(o) for o in outs],
           n * 63, H.stream())
    return outs


def three_view_splines(m12_1, m12_2, m23_1, m23_2, first_box, out_box, nrigid, img_h, img_w):
    """The streaming three-view push between the chains' smoothed meshes and the render in ONE launch (ss_three_view_splines =
    three_view_align -> three_view_normalize -> tps_solve -> tps_points -> three_view_finish on the first canvas, then the final
    meshes normalised on the output canvas and tps_solve_shared onto the rigid mesh; bit-identical to those launches).
    m*: [k,7,9,2] contiguous, LR scale -> ((mesh1, middle, mesh3) [1,k,7,9,2] first-canvas pixels, src [k,3,63,2], T [k,3,2,66]).
    The overflow watcher is not touched: render_footprints(..., watch=...) or canvas_watch(src, ...)."""
    k = m12_1.numel() // 126
    for m in (m12_1, m12_2, m23_1, m23_2):
        assert m.is_contiguous() and m.dtype == torch.float32 and m.numel() == k * 126
    assert nrigid.numel() == 126 and nrigid.is_contiguous()
    d = m12_1.device
    outs = [torch.empty((1, k, 7, 9, 2), device=d, dtype=torch.float32) for _ in range(3)]
    src = torch.empty((k, 3, 63, 2), device=d, dtype=torch.float32)
    T = torch.empty((k, 3, 2, 66), device=d, dtype=torch.float32)
    H.call('ss_three_view_splines', H.dptr(m12_1), H.dptr(m12_2), H.dptr(m23_1), H.dptr(m23_2), 126, H.dptr(first_box), H.dptr(out_box),
           H.dptr(nrigid), *[H.dptr(o) for o in outs], H.dptr(src), H.dptr(T), k, float(img_h), float(img_w), H.stream())
    return tuple(outs), src, T


def fill(t, value=0.0):
    """t[...] = value in place (t contiguous fp32)."""
    H.call('ss_fill_f32', H.dptr(t), float(value), t.numel(), H.stream())
    return t


# ------------------------------------------------------------------ smooth glue
def smooth_embed(sm1, sm2, ts1, ts2, e1w, e1b, e3w, e3b, nw, t, wstride, zero_first):
    hidden = torch.empty((nw, t, 7, 9, 128), device=sm1.device, dtype=torch.float32)
    H.call('ss_smooth_embed', H.dptr(sm1), H.dptr(sm2), H.dptr(ts1), H.dptr(ts2), H.dptr(e1w), H.dptr(e1b),
           H.dptr(e3w), H.dptr(e3b), H.dptr(hidden), nw, t, wstride, int(zero_first), H.stream())
    return hidden


def smooth_finalize(sm1, sm2, ts1, ts2, delta, nw, t, wstride, zero_first):
    names = ('ori_mesh1', 'ori_mesh2', 'ori_path1', 'ori_path2', 'smooth_mesh1', 'smooth_mesh2', 'smooth_path1',
             'smooth_path2')
    outs = {k: torch.empty((nw, t, 7, 9, 2), device=sm1.device, dtype=torch.float32) for k in names}
    H.call('ss_smooth_finalize', H.dptr(sm1), H.dptr(sm2), H.dptr(ts1), H.dptr(ts2), H.dptr(delta),
           *[H.dptr(outs[k]) for k in names], nw, t, wstride, int(zero_first), H.stream())
    return outs



def smooth_stitch(sm1, sm2, ts1, ts2, delta, nw, t, want_paths=True):
    """The clip's tensors straight from the sliding windows (ss_smooth_stitch): sm*/ts* [n,7,9,2] with n = nw + t - 1,
    delta [nw,t,7,9,4] -> dict(ori_mesh1/2, smooth_mesh1/2 [1,n,7,9,2] (+ ori_path2, smooth_path2))."""
    n = nw + t - 1
    assert sm1.shape[0] == n and delta.numel() == nw * t * 63 * 4
    names = ['ori_mesh1', 'ori_mesh2', 'smooth_mesh1', 'smooth_mesh2'] + (['ori_path2', 'smooth_path2'] if want_paths else [])
    outs = {k: torch.empty((1, n, 7, 9, 2), device=sm1.device, dtype=torch.float32) for k in names}
    H.call('ss_smooth_stitch', H.dptr(sm1), H.dptr(sm2), H.dptr(ts1), H.dptr(ts2), H.dptr(delta),
           H.dptr(outs['ori_mesh1']), H.dptr(outs['ori_mesh2']), H.dptr(outs['smooth_mesh1']), H.dptr(outs['smooth_mesh2']),
           H.dptr(outs.get('ori_path2'), True), H.dptr(outs.get('smooth_path2'), True), nw, t, H.stream())
    return outs
