"""Parameter containers with the reference checkpoint layout + weight preparation for the HIP engine.

The nn.Conv2d / nn.BatchNorm2d / nn.Linear / nn.Conv3d objects below are never *called*: they only
hold parameters under the reference's state-dict keys so that
`net.load_state_dict(torch.load('spatial_warp.pth')['model'])` works with strict=True.  The forward
passes run on libstabstitch_hip.so with weights repacked once per load:
  * eval-mode BatchNorm folded into the preceding conv (scale into the filter, shift into a bias);
  * filters [cout,cin,kh,kw] -> [cout,1,kh,kw,cin_pad4] (NHWC taps, zero taps for padded channels);
  * first FC of every regressor re-indexed from the reference's NCHW flatten to NHWC flatten.
Architecture restated from torchvision 0.14.1 resnet18 (spatial_network.py:123-139) and the
reference's regressors (spatial_network.py:147-259, temporal_network.py:65-105).
"""
import os

import torch
import torch.nn as nn

from . import ops


def pad4(c):
    return (c + 3) // 4 * 4


class BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))
        self.stride = stride


def res_layer(cin, cout, stride):
    return nn.Sequential(BasicBlock(cin, cout, stride), BasicBlock(cout, cout, 1))


def make_trunk():
    """(feature_extractor_stage1, feature_extractor_stage2) with the reference's Sequential indices."""
    stage1 = nn.Sequential(nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True),
                           nn.MaxPool2d(3, 2, 1), res_layer(64, 64, 1), res_layer(64, 128, 2))
    stage2 = nn.Sequential(res_layer(128, 256, 2))
    return stage1, stage2


def regress_convs(cin, widths):
    mods, c = [], cin
    for wd in widths:
        mods += [nn.Conv2d(c, wd, 3, padding=1, bias=False), nn.ReLU(inplace=True),
                 nn.Conv2d(wd, wd, 3, padding=1, bias=False), nn.ReLU(inplace=True), nn.MaxPool2d(2, 2)]
        c = wd
    return nn.Sequential(*mods)


def regress_fc(fin, h1, h2, fout):
    return nn.Sequential(nn.Linear(fin, h1), nn.ReLU(inplace=True), nn.Linear(h1, h2), nn.ReLU(inplace=True),
                         nn.Linear(h2, fout))


# --------------------------------------------------------------------------- weight preparation
def pack_conv2d(conv, bn=None):
    """-> (wgt [cout,1,kh,kw,cin_pad4], bias [cout] | None) on the conv's device."""
    w = conv.weight.detach().float()
    bias = None
    if bn is not None:
        scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
        w = w * scale.view(-1, 1, 1, 1)
        bias = (bn.bias.detach().float() - bn.running_mean.detach().float() * scale).contiguous()
    cout, cin, kh, kw = w.shape
    cp = pad4(cin)
    out = torch.zeros((cout, 1, kh, kw, cp), device=w.device, dtype=torch.float32)
    out[:, 0, :, :, :cin] = w.permute(0, 2, 3, 1)
    return out.contiguous(), bias


def pack_stem3(conv, bn):
    """The 7x7 / stride-2 stem of the trunk for ss_conv_stem3: [cout,3,7,7] (+ folded BN) -> (wgt [cout,7,24], bias):
    wgt[co][dh][3 dw + c] = w[co][c][dh][dw]; entries 21..23 of every filter row are zero."""
    w = conv.weight.detach().float()
    scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
    w = w * scale.view(-1, 1, 1, 1)
    bias = (bn.bias.detach().float() - bn.running_mean.detach().float() * scale).contiguous()
    cout = w.shape[0]
    assert tuple(w.shape[1:]) == (3, 7, 7)
    out = torch.zeros((cout, 7, 24), device=w.device, dtype=torch.float32)
    out[:, :, :21] = w.permute(0, 2, 3, 1).reshape(cout, 7, 21)         # [co][dh][dw][c] -> dw-major, channel fastest
    return out.contiguous(), bias


def pack_conv3d(conv):
    w = conv.weight.detach().float()          # [cout,cin,kt,kh,kw]
    return w.permute(0, 2, 3, 4, 1).contiguous(), conv.bias.detach().float().contiguous()


def pack_fc_first(lin, c, hw):
    """nn.Linear over an NCHW flatten (index c*hw + p) -> columns re-ordered for an NHWC flatten (p*c + ch)."""
    w = lin.weight.detach().float()
    n = w.shape[0]
    w = w.view(n, c, hw).permute(0, 2, 1).reshape(n, hw * c).contiguous()
    return w, lin.bias.detach().float().contiguous()


def pack_fc(lin):
    return lin.weight.detach().float().contiguous(), lin.bias.detach().float().contiguous()


_version_counter = [0]


class PreparedMixin:
    """Lazy, invalidating cache of repacked weights.  `weights_version` changes whenever the parameters may have
    changed (load_state_dict, .to(), .cuda(), ...): caches DERIVED from two nets' prepared weights (shared stems, twin
    trunks, captured HIP graphs) are keyed on the versions of both and rebuilt on a mismatch."""

    def _prepared(self):
        dev = next(self.parameters()).device
        cache = getattr(self, '_prep_cache', None)
        if cache is None or cache[0] != dev:
            with torch.no_grad():
                cache = (dev, self._prepare())
            object.__setattr__(self, '_prep_cache', cache)
        return cache[1]

    @property
    def weights_version(self):
        v = getattr(self, '_weights_version', None)
        if v is None:
            v = self._bump()
        return v

    def _bump(self):
        _version_counter[0] += 1
        object.__setattr__(self, '_weights_version', _version_counter[0])
        return _version_counter[0]

    def _invalidate(self):
        object.__setattr__(self, '_prep_cache', None)
        self._bump()

    def load_state_dict(self, *a, **k):
        self._invalidate()
        return super().load_state_dict(*a, **k)

    def _apply(self, fn, *a, **k):
        self._invalidate()
        return super()._apply(fn, *a, **k)


# --------------------------------------------------------------------------- forward helpers (HIP)
def prep_trunk_stage1(stage1):
    p = {'conv1': pack_stem3(stage1[0], stage1[1])}
    p['layer1'] = [prep_block(b) for b in stage1[4]]
    p['layer2'] = [prep_block(b) for b in stage1[5]]
    return p


def prep_trunk_stage2(stage2):
    return {'layer3': [prep_block(b) for b in stage2[0]]}


def prep_block(blk):
    d = {'c1': pack_conv2d(blk.conv1, blk.bn1), 'c2': pack_conv2d(blk.conv2, blk.bn2), 'stride': blk.stride,
         'ds': None}
    if blk.downsample is not None:
        d['ds'] = pack_conv2d(blk.downsample[0], blk.downsample[1])
    return d


def run_block(x, p):
    y = ops.conv(x, p['c1'][0], p['c1'][1], stride=p['stride'], pad=(0, 1, 1), relu=True)
    idt = x if p['ds'] is None else ops.conv(x, p['ds'][0], p['ds'][1], stride=p['stride'], pad=(0, 0, 0))
    return ops.conv(y, p['c2'][0], p['c2'][1], res=idt, stride=1, pad=(0, 1, 1), relu=True)


STAGE1_CHUNK = int(os.environ.get('SS_STAGE1_CHUNK', '64'))      # images per trunk pass
# conv1 + max-pool run in sub-chunks so that conv1's output (11 MB per image) is still in the 256 MB MALL when the
# pool reads it; the rest of the trunk runs on the whole chunk (large launches)
CONV1_CHUNK = int(os.environ.get('SS_CONV1_CHUNK', '16'))


def run_stage1(x_nchw, p, chunk=None):
    """NCHW input(s) in [-1,1] -> nhwc [n,H/8,W/8,128].  `x_nchw` may be a list of [n_i,3,H,W] tensors: they are
    laid out back to back in one NHWC buffer (no torch.cat of the inputs) and run as one batch."""
    chunk = chunk or STAGE1_CHUNK
    buf = ops.stem_input(x_nchw)                     # [total, h, w + 8, 3]
    total, h, w = buf.shape[0], buf.shape[1], buf.shape[2] - 8
    whole = None
    for s in range(0, total, chunk):
        m = min(chunk, total - s)
        if ops.STEM_FUSED and p['conv1'][0].shape[0] == 64:
            x = ops.stem_pool(buf[s:s + m], p['conv1'][0], p['conv1'][1])[0]      # conv + BN + ReLU + pool, one kernel
        elif CONV1_CHUNK > 0 and m > CONV1_CHUNK and h % 2 == 0 and w % 2 == 0:
            x = torch.empty((m, (h // 2 + 1) // 2, (w // 2 + 1) // 2, 64), device=buf.device, dtype=torch.float32)
            for c0 in range(0, m, CONV1_CHUNK):
                y = ops.conv_stem(buf[s + c0:s + min(c0 + CONV1_CHUNK, m)], p['conv1'][0], p['conv1'][1], relu=True)
                ops.maxpool(y, 3, 2, 1, out=x[c0:c0 + y.shape[0]])
        else:
            x = ops.conv_stem(buf[s:s + chunk], p['conv1'][0], p['conv1'][1], relu=True)
            x = ops.maxpool(x, 3, 2, 1)
        for b in p['layer1']:
            x = run_block(x, b)
        for b in p['layer2']:
            x = run_block(x, b)
        if total <= chunk:
            return x
        if whole is None:                # several trunk passes: their features land in one preallocated tensor (no torch.cat)
            whole = torch.empty((total,) + tuple(x.shape[1:]), device=x.device, dtype=torch.float32)
        whole[s:s + x.shape[0]].copy_(x)
    return whole


def run_stage2(x, p):
    for b in p['layer3']:
        x = run_block(x, b)
    return x


def prep_regressor(convs, fcs, last_c, last_hw):
    cw = [pack_conv2d(m)[0] for m in convs if isinstance(m, nn.Conv2d)]
    lin = [m for m in fcs if isinstance(m, nn.Linear)]
    return {'convs': cw, 'fc': [pack_fc_first(lin[0], last_c, last_hw), pack_fc(lin[1]), pack_fc(lin[2])]}


# The regressors and the SmoothNet windows run in batch chunks: one conv launch addresses its input with 32-bit byte
# offsets (< 2 GiB per group, ss_conv_nhwc returns SS_ERR_UNSUPPORTED beyond), and the reference handles clips of any
# length frame by frame.  512 cost volumes of 45x60x124 floats are 0.69 GB.
REG_CHUNK = int(os.environ.get('SS_REG_CHUNK', '512'))


def _fc_tail(flat, fc, out=None, out_slices=None, row0=0):
    """The three FC layers of a regressor on flattened NHWC features.  out: [rows, nout] destination of the last layer;
    out_slices: list of (r0, r1, dst) -- rows [r0, r1) of the WHOLE batch (this call covers rows row0 .. row0 + rows) go to
    dst[r - r0] (dst [r1 - r0, nout] contiguous): the last layer then runs once per overlapping slice and writes in place
    (e.g. every view's motions behind a zero first frame, no torch.cat)."""
    if flat.shape[1] != fc[0][0].shape[1]:
        raise ValueError('regressor expects %d features, got %d: the reference hard-wires 360x480 inputs'
                         % (fc[0][0].shape[1], flat.shape[1]))
    y = ops.linear(flat, fc[0][0], fc[0][1], relu=True)
    y = ops.linear(y, fc[1][0], fc[1][1], relu=True)
    if out_slices is None:
        return ops.linear(y, fc[2][0], fc[2][1], relu=False, out=out)
    rows = flat.shape[0]
    for r0, r1, dst in out_slices:
        a, b = max(r0, row0), min(r1, row0 + rows)
        if a < b:
            ops.linear(y[a - row0:b - row0], fc[2][0], fc[2][1], relu=False, out=dst[a - r0:b - r0])
    return None


def run_regressor(x, p, chunk=None, out=None, out_slices=None, row0=0):
    """x nhwc; pairs of 3x3 conv+ReLU then 2x2 max-pool; NHWC flatten; 3 FC.  out / out_slices: see _fc_tail."""
    chunk = chunk or REG_CHUNK
    if x.shape[0] > chunk:
        nout = p['fc'][2][0].shape[0]
        if out is None and out_slices is None:
            out = torch.empty((x.shape[0], nout), device=x.device, dtype=torch.float32)
        for s in range(0, x.shape[0], chunk):
            e = min(s + chunk, x.shape[0])
            run_regressor(x[s:e], p, chunk, None if out is None else out[s:e], out_slices, row0 + s)
        return out
    for i, w in enumerate(p['convs']):          # conv, ReLU, conv, ReLU, MaxPool2d(2, 2): the pool rides in the second conv's kernel
        x = ops.conv(x, w, None, stride=1, pad=(0, 1, 1), relu=True, pool2=bool(i & 1))
    return _fc_tail(x.reshape(x.shape[0], -1), p['fc'], out, out_slices, row0)


def pair_regressors(pa, pb):
    """Stack the conv weights of two regressors of identical architecture (regressNet2 ref / tgt) for grouped launches."""
    return {'convs': [torch.stack((a, b), 0).contiguous() for a, b in zip(pa['convs'], pb['convs'])],
            'fc': (pa['fc'], pb['fc'])}


def run_regressor_pair(x, pp, chunk=None, outs=None):
    """x [2,n,h,w,c] nhwc (one input per regressor) -> (out_a, out_b): the eight convs of both regressors run as
    eight grouped launches instead of sixteen, the pools on the joint batch; the FC stacks stay per regressor."""
    g, n = x.shape[0], x.shape[1]
    chunk = chunk or REG_CHUNK
    if n > chunk:
        if outs is None:
            outs = [torch.empty((n, pp['fc'][k][2][0].shape[0]), device=x.device, dtype=torch.float32) for k in range(g)]
        for s in range(0, n, chunk):
            e = min(s + chunk, n)
            run_regressor_pair(x[:, s:e].contiguous(), pp, chunk, [o[s:e] for o in outs])
        return outs
    for i, w in enumerate(pp['convs']):
        x = ops.conv_grouped(x, w, None, None, stride=1, pad=(0, 1, 1), relu=True, pool2=bool(i & 1))
    return [_fc_tail(x[k].reshape(n, -1), pp['fc'][k], None if outs is None else outs[k]) for k in range(g)]


# --------------------------------------------------------------------------- four heads, one set of launches (round 4)
# SpatialNet's two stage-2 regressors (regressNet2 ref / tgt, spatial_network.py:197-259) and TemporalNet's regressor applied
# to two views (temporal_network.py:65-105) have the same architecture behind their first convolution (only its input width
# differs: 121 / 49 displacement channels).  With equally many images per head -- S pairs per push in streaming mode, a chunk
# of b pairs offline -- conv2 .. conv8 and the three FC layers of all four heads run as ONE grouped launch each instead of
# two sets (11 launches fewer per pass; on the <= 11 x 15 maps the merged launches fill the chip where each set alone left
# half of it idle).  Per image the arithmetic is the separate heads'; the conv engine's kernel choice follows the launch size.
QUAD = os.environ.get('SS_QUAD_REGRESSOR', '1') == '1'
FC_GEMM_ROWS = int(os.environ.get('SS_FC_GEMM_ROWS', '16'))      # rows per head from which FC1 / FC2 of the heads run on the conv engine


def quad_regressors(pp_pair, p_temp, gt=2):
    """pp_pair: pair_regressors(r2_ref, r2_tgt); p_temp: TemporalNet's prepared regressor -> weights of the (2 + gt)-head
    launches (heads: ref, tgt, then TemporalNet's on gt = 2 views -- its filters appear gt times -- or on one view)."""
    convs = [torch.cat((wp,) + (wt[None],) * gt, 0).contiguous() for wp, wt in zip(pp_pair['convs'][1:], p_temp['convs'][1:])]
    fa, fb, ft = pp_pair['fc'][0], pp_pair['fc'][1], p_temp['fc']
    fc = [(torch.stack((fa[l][0], fb[l][0]) + (ft[l][0],) * gt, 0).contiguous(),
           torch.stack((fa[l][1], fb[l][1]) + (ft[l][1],) * gt, 0).contiguous()) for l in range(3)]
    return {'conv1_pair': pp_pair['convs'][0], 'conv1_t': p_temp['convs'][0], 'convs': convs, 'fc': fc}


def get_quad(spatial_net, temporal_net, gt=2):
    """The (2 + gt)-head weights of a (SpatialNet, TemporalNet) pair, cached on SpatialNet's prepared dict and keyed on both
    nets' weight versions (rebuilt when either was reloaded / moved)."""
    sp, tp = spatial_net._prepared(), temporal_net._prepared()
    ver = (spatial_net.weights_version, temporal_net.weights_version)
    if sp.get('quad_version') != ver:
        sp['quad'] = {}
        sp['quad_version'] = ver
    if gt not in sp['quad']:
        sp['quad'][gt] = quad_regressors(sp['r2_pair'], tp['r2'], gt)
    return sp['quad'][gt]


def run_regressor_quad(cv_s, cv_t, q, outs):
    """cv_s [2,b,h,w,124] (both directions of SpatialNet's stage 2), cv_t [gt,b,h,w,52] (TemporalNet, gt = 2 views or 1);
    outs: 2 + gt contiguous destinations of b * 126 floats (offset_2_ref, offset_2_tgt, temporal motions per view)."""
    b, gt = cv_s.shape[1], cv_t.shape[0]
    g = 2 + gt
    assert cv_t.shape[1] == b and b <= REG_CHUNK and q['convs'][0].shape[0] == g and len(outs) == g
    y = torch.empty((g, b) + tuple(cv_s.shape[2:4]) + (q['conv1_pair'].shape[1],), device=cv_s.device, dtype=torch.float32)
    ops.conv_grouped(cv_s, q['conv1_pair'], None, None, stride=1, pad=(0, 1, 1), relu=True, out=y[0:2])
    ops.conv(cv_t.view((gt * b,) + tuple(cv_t.shape[2:])), q['conv1_t'], None, stride=1, pad=(0, 1, 1), relu=True,
             out=y[2:g].view((gt * b,) + tuple(y.shape[2:])))
    x = y
    for i, w in enumerate(q['convs']):          # conv2 .. conv8: a 2x2 max-pool rides behind every second convolution
        x = ops.conv_grouped(x, w, None, None, stride=1, pad=(0, 1, 1), relu=True, pool2=not (i & 1))
    h = x.reshape(g, b, -1)
    if h.shape[2] != q['fc'][0][0].shape[2]:
        raise ValueError('regressor expects %d features, got %d: the reference hard-wires 360x480 inputs'
                         % (q['fc'][0][0].shape[2], h.shape[2]))
    for l in (0, 1):
        w, bias = q['fc'][l]
        if b >= FC_GEMM_ROWS and not ops.is_deterministic():
            # enough rows for a matrix-core tile: the grouped product as a 1x1 convolution on the conv engine (tools/ab_fc_grouped.py,
            # 4 heads x 32 rows: 1536 -> 1024 in 14.9 us against 34.8 for the one-wave-per-neuron kernel, 1024 -> 512 in 9.0 against 13.9)
            h = ops.conv_grouped(h.view(g, 1, 1, b, h.shape[2]), w.view(g, w.shape[1], 1, 1, 1, w.shape[2]), bias, None, stride=1,
                                 pad=(0, 0, 0), relu=True).view(g, b, w.shape[1])
        else:
            h = ops.linear_grouped(h, w, bias, relu=True)
    ops.linear_grouped(h, q['fc'][2][0], q['fc'][2][1], relu=False, outs=outs)


# --------------------------------------------------------------------------- twin trunks (streaming mode)
def _stack2(a, b):
    return None if a is None else torch.stack((a, b), 0).contiguous()


def pair_trunks(pa, pb):
    """Stack two stage-1 trunks of identical architecture (SpatialNet / TemporalNet) for grouped launches."""
    def blk(x, y):
        d = {'c1': (_stack2(x['c1'][0], y['c1'][0]), _stack2(x['c1'][1], y['c1'][1])),
             'c2': (_stack2(x['c2'][0], y['c2'][0]), _stack2(x['c2'][1], y['c2'][1])), 'stride': x['stride'], 'ds': None}
        if x['ds'] is not None:
            d['ds'] = (_stack2(x['ds'][0], y['ds'][0]), _stack2(x['ds'][1], y['ds'][1]))
        return d
    return {'conv1': (_stack2(pa['conv1'][0], pb['conv1'][0]), _stack2(pa['conv1'][1], pb['conv1'][1])),
            'layer1': [blk(x, y) for x, y in zip(pa['layer1'], pb['layer1'])],
            'layer2': [blk(x, y) for x, y in zip(pa['layer2'], pb['layer2'])]}


def run_stage1_pair(xs, pp):
    """The same NCHW inputs through TWO trunks (weights stacked by pair_trunks) -> nhwc [2,n,H/8,W/8,128]:
    every layer is one grouped launch; conv1 reads the shared input once per group (group stride 0)."""
    buf = ops.stem_input(xs)
    total = buf.shape[0]
    if ops.STEM_FUSED:
        y = ops.stem_pool(buf, pp['conv1'][0], pp['conv1'][1])                   # [2, total, H/4, W/4, 64], one kernel
    else:
        y = ops.conv_stem(buf, pp['conv1'][0], pp['conv1'][1], relu=True)       # [2, total, H/2, W/2, 64]
        y = ops.maxpool(y.view(2 * total, *y.shape[2:]), 3, 2, 1)
        y = y.view(2, total, *y.shape[1:])
    for b in pp['layer1'] + pp['layer2']:
        t = ops.conv_grouped(y, b['c1'][0], b['c1'][1], None, stride=b['stride'], pad=(0, 1, 1), relu=True)
        idt = y if b['ds'] is None else ops.conv_grouped(y, b['ds'][0], b['ds'][1], None, stride=b['stride'], pad=(0, 0, 0))
        y = ops.conv_grouped(t, b['c2'][0], b['c2'][1], idt, stride=1, pad=(0, 1, 1), relu=True)
    return y


# --------------------------------------------------------------------------- shared stem (offline 2-view path)
def pair_stems(pa, pb):
    """conv1 of two trunks as ONE convolution with 2 x 64 filters (same input, same geometry)."""
    return (torch.cat((pa['conv1'][0], pb['conv1'][0]), 0).contiguous(),
            torch.cat((pa['conv1'][1], pb['conv1'][1]), 0).contiguous())


def run_stem_shared(xs, stem):
    """NCHW inputs (list) -> (pool_a, pool_b) nhwc [n,H/4,W/4,64]: conv1 + ReLU + max-pool of two trunks that read the
    same frames, computed by one 128-filter conv1 (the input tile, the row bookkeeping and the NHWC conversion are
    shared) and a pool that splits the channels; 16-image sub-chunks keep conv1's output MALL-resident."""
    buf = ops.stem_input(xs)
    total, h, w = buf.shape[0], buf.shape[1], buf.shape[2] - 8
    if ops.STEM_FUSED and stem[0].shape[0] == 128:
        y = ops.stem_pool(buf, stem[0], stem[1])             # [2,total,H/4,W/4,64]: both stems + pools in one kernel
        return y[0], y[1]
    dev = buf.device
    ho, wo = ((h - 1) // 2 + 2) // 2, ((w - 1) // 2 + 2) // 2
    pa = torch.empty((total, ho, wo, 64), device=dev, dtype=torch.float32)
    pb = torch.empty((total, ho, wo, 64), device=dev, dtype=torch.float32)
    step = CONV1_CHUNK if CONV1_CHUNK > 0 else total
    for c0 in range(0, total, step):
        c1 = min(c0 + step, total)
        y = ops.conv_stem(buf[c0:c1], stem[0], stem[1], relu=True)                          # [m,H/2,W/2,128]
        ops.maxpool_split(y, 3, 2, 1, pa[c0:c1], pb[c0:c1])
    return pa, pb


def run_trunk_body(x, p):
    """layer1 + layer2 of a stage-1 trunk on its pooled stem output."""
    for b in p['layer1']:
        x = run_block(x, b)
    for b in p['layer2']:
        x = run_block(x, b)
    return x
