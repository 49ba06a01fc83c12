"""Online 2-view / 3-view stitching pipeline on the MI355X HIP engine.

Tensor-level counterpart of `test()` in Full_model_inference/Codes/test_online_tra.py:158-426 and
test_online_tra_threeview.py:95-519 (file / video I/O excluded), with the reference's helper names
(`get_stable_sqe`, `linear_blender`, `recover_mesh`, `get_rigid_mesh`, `get_norm_mesh`).

The reference walks the clip frame by frame with batch 1.  Here every stage is one batched pass
over the whole clip (the clip is resident in HBM): SpatialNet over all frame pairs, TemporalNet
over all frames of a view, one batched tsmotion composition per view, all sliding SmoothNet windows
as one batch, one batched TPS solve for every (frame, view), then one fused warp+blend launch for the
whole clip.  The only host round trip is the data-dependent canvas size (test_online_tra.py:122-123).

Everything tensor-sized (frames, feature maps, cost volumes, canvases) is computed by the HIP kernels, and in the 2-view
path so is the mesh-sized bookkeeping (the clip's meshes and the metric harness's stitched paths straight from the sliding
windows: `ops.smooth_stitch`; the render's control points: `ops.mesh_normalize_views`): a steady-state 2-view clip issues
no torch (aten) kernel at all (tools/trace_torch_ops.py); the three-view mesh alignment (`three_view_compose`: scale, mean
offset, middle mesh, first canvas, TPS re-projection) runs on `ss_three_view_align / _finish` + the mesh kernels as well.

Long videos: `run_two_view_long` keeps the reference's ONE global canvas (test_online_tra.py:106-120) in bounded device
memory -- pass 1 estimates the meshes chunk by chunk from the LR frames only, pass 2 renders chunk by chunk onto the
shared canvas.
"""
import glob
import os

import torch

from . import grid_res, ops
from .spatial_network import get_rigid_mesh, get_norm_mesh, build_SpatialNet  # noqa: F401 (reference names)

grid_h = grid_res.GRID_H
grid_w = grid_res.GRID_W
LR_H, LR_W = 360, 480
WINDOW = 7


def recover_mesh(norm_mesh, height, width):
    """test_online_tra.py:61-69."""
    b = norm_mesh.size()[0]
    x = (norm_mesh[..., 0] + 1) * float(width) / 2.
    y = (norm_mesh[..., 1] + 1) * float(height) / 2.
    return torch.stack([x, y], 2).reshape([b, grid_h + 1, grid_w + 1, 2])


def linear_blender(ref, tgt, ref_m, tgt_m, mask=False):
    """test_online_tra.py:34-58; ref,tgt [1,3,H,W], ref_m,tgt_m [1,1,H,W]."""
    if mask:
        return ops.linear_blend(None, None, ref_m[0, 0].contiguous(), tgt_m[0, 0].contiguous(), True)[None, None]
    return ops.linear_blend(ref[0].contiguous(), tgt[0].contiguous(), ref_m[0, 0].contiguous(),
                            tgt_m[0, 0].contiguous())[None]


# ------------------------------------------------------------------ stages (batched over the clip)
def _stack(lst, dev):
    return torch.stack([t.to(dev, non_blocking=True).float() for t in lst], 0)


OVERLAP_STREAMS = os.environ.get('SS_OVERLAP_STREAMS', '0') == '1'
_side = {}


def _side_stream(dev):
    s = _side.get(dev)
    if s is None:
        s = torch.cuda.Stream(device=dev)
        _side[dev] = s
    return s


SPATIAL_CHUNK = int(os.environ.get('SS_SPATIAL_CHUNK', '32'))     # frame pairs per SpatialNet pass


def _mesh_pair(n, dev):
    return (torch.empty((n, grid_h + 1, grid_w + 1, 2), device=dev, dtype=torch.float32),
            torch.empty((n, grid_h + 1, grid_w + 1, 2), device=dev, dtype=torch.float32))


@torch.no_grad()
def spatial_stage(spatial_net, lr1, lr2, chunk=None, cache1=None):
    """lr1, lr2 [N,3,360,480] device -> smotion1, smotion2 [N,7,9,2].
    cache1: per chunk the (f64, f32) trunk features of view 1 kept by an earlier pass (then only view 2 goes through
    the trunk)."""
    chunk = chunk or SPATIAL_CHUNK
    n = lr1.shape[0]
    m1, m2 = _mesh_pair(n, lr1.device)                   # every chunk writes its rows in place (no torch.cat)
    for i, s in enumerate(range(0, n, chunk)):
        e = min(s + chunk, n)
        if cache1 is None:
            off = spatial_net(lr1[s:e], lr2[s:e])
        else:
            f64_2, f32_2 = spatial_net.trunk_features([lr2[s:e]])
            off = spatial_net.forward_pair(cache1[i][0], f64_2, cache1[i][1], f32_2, LR_H, LR_W)
        ops.spatial_meshes(off[0], off[1], off[2], LR_H, LR_W, out=(m1[s:e], m2[s:e]))
    return m1, m2


@torch.no_grad()
def temporal_stage(temporal_net, lr):
    """lr [N,3,360,480] device -> tmotion [N,7,9,2] (frame 0 = 0)."""
    return temporal_stage_views(temporal_net, [lr])[0]


@torch.no_grad()
def temporal_stage_views(temporal_net, lrs):
    """Both (all) views of a clip in ONE batched pass: lrs = list of V tensors [N,3,360,480]
    -> list of V tmotion tensors [N,7,9,2] (frame 0 = 0, written in place)."""
    n = lrs[0].shape[0]
    f = temporal_net.features(lrs)                                   # [V*N,45,60,128], view-major
    return temporal_net.motions_from_view_features([f[i * n:(i + 1) * n] for i in range(len(lrs))], zero_first=True)


SHARED_STEM = os.environ.get('SS_SHARED_STEM', '1') == '1'
# SpatialNet and TemporalNet behind the shared stem on two HIP streams (measured); off by default
JOINT_OVERLAP = os.environ.get('SS_JOINT_OVERLAP', '0') == '1'
# the 4-head path: TemporalNet's trunk beside SpatialNet's small-launch chain (second stream; fork AFTER SpatialNet's trunk:
# forked before it, two full-chip trunks only get in each other's way, measured -2 %; forked behind it +1.1 % frames/s, bit-identical).
# Opt-in (SS_QUAD_OVERLAP=1), not the headline: with kernels of two streams sharing the chip a launch's duration is no longer
# its own, and the conv engine's roofline fraction -- flop / summed launch durations -- reads 0.59 instead of 0.64 for the
# same work done sooner; bench.py reports the overlapped run under other_configs.
QUAD_OVERLAP = os.environ.get('SS_QUAD_OVERLAP', '0') == '1'


class JointEstimator:
    """SpatialNet and TemporalNet of a 2-view stream, chunk by chunk, in memory that does not grow with the stream (beyond
    the [N,7,9,2] motions themselves).  Both nets start with the same 7x7/2 conv + pool on the same LR frames
    (spatial_network.py:127-130 and temporal_network.py:47-50 build identical stems), so the stem runs ONCE with 2 x 64
    filters and each net continues from its half of the channels.  TemporalNet's cost volumes pair every frame with its
    predecessor; across a chunk boundary the predecessor's stage-1 features are carried over (one [1,45,60,128] map per
    view, copied into a buffer of its own), so feeding a video in chunks launches exactly the kernels a resident clip of the same chunking launches:
    the motions are bit-identical however the frames arrive.
        est = JointEstimator(spatial_net, temporal_net, n_frames, device); est.push(lr1[s:e], lr2[s:e]) ...; est.result()
    tmotion1: view 1's temporal motions when a previous pass already produced them (three-view: the middle view is view 2
    of pair (1,2) and view 1 of pair (2,3)); its TemporalNet trunk is then skipped.  cache2: list that receives view 2's
    SpatialNet trunk features per chunk (for a later pair that starts with this view; resident clips only)."""

    def __init__(self, spatial_net, temporal_net, n_frames, device, tmotion1=None, cache2=None):
        from . import layers as L
        self.L = L
        self.spatial_net, self.temporal_net = spatial_net, temporal_net
        self.sp, self.tp = spatial_net._prepared(), temporal_net._prepared()
        # shared-stem filters derive from BOTH nets' weights: keyed on both versions, one live entry (replaced on a miss)
        ver = (spatial_net.weights_version, temporal_net.weights_version)
        if self.sp.get('stem_pair_version') != ver:
            self.sp['stem_pair'] = L.pair_stems(self.sp['s1'], self.tp['s1'])
            self.sp['stem_pair_version'] = ver
        self.n, self.pos = n_frames, 0
        self.m1, self.m2 = _mesh_pair(n_frames, device)
        self.tmotion1, self.cache2 = tmotion1, cache2
        self.views = (1,) if tmotion1 is not None else (0, 1)
        self.tm = torch.empty((len(self.views), n_frames, grid_h + 1, grid_w + 1, 2), device=device, dtype=torch.float32)
        for i in range(len(self.views)):
            ops.fill(self.tm[i, 0])                      # the zero motion of frame 0 (temporal_network.py:31-33)
        self.carry = None

    @torch.no_grad()
    def push(self, lr1, lr2, cache1=None):
        """The next b frames of both views, [b,3,360,480] device tensors.  cache1 = (f64, f32): view 1's SpatialNet trunk
        features for exactly these frames from an earlier pair's `cache2` (needs tmotion1; lr1 is not read then -- only
        view 2 goes through the stem and the trunks)."""
        L, sp, tp = self.L, self.sp, self.tp
        b = lr2.shape[0]
        s, e = self.pos, self.pos + b
        if e > self.n:
            raise ValueError('more frames pushed (%d) than announced (%d)' % (e, self.n))
        if cache1 is not None and self.tmotion1 is None:
            raise ValueError('cache1 needs tmotion1: view 1 is skipped entirely')
        xa, xb = L.run_stem_shared([lr2] if cache1 is not None else [lr1, lr2], sp['stem_pair'])
        if cache1 is None and self.tmotion1 is not None:
            xb = xb[b:]                                     # TemporalNet continues on view 2 only
        if L.QUAD and (cache1 is None) == (self.tmotion1 is None) and b <= L.REG_CHUNK:
            self._quad(xa, xb, b, s, e, cache1)
        elif JOINT_OVERLAP:
            # SpatialNet (main stream) and TemporalNet (side stream) are independent behind the shared stem: two chains of
            # launches whose partially filled last rounds top each other up
            main = torch.cuda.current_stream(xb.device)
            side = _side_stream(xb.device)
            side.wait_stream(main)
            xb.record_stream(side)
            with torch.cuda.stream(side):
                self._temporal(xb, b, s, e)
            self._spatial(xa, b, s, e, cache1)
            main.wait_stream(side)
        else:
            self._spatial(xa, b, s, e, cache1)
            self._temporal(xb, b, s, e)
        self.pos = e

    def _quad(self, xa, xb, b, s, e, cache1=None):
        """Both nets of a 2-view chunk with the regressor heads in shared launches (layers.run_regressor_quad): SpatialNet's
        ref / tgt heads on the b pairs, TemporalNet's head on b consecutive-frame pairs per view -- the first chunk has b - 1
        (frame 0 has no predecessor): its head runs on a zero cost volume in row 0, whose result lands on frame 0's motion
        and is replaced by the zero motion of temporal_network.py:31-33.  With `cache1` (second pair of a three-view clip:
        view 1's trunk features and temporal motions come from the first pair) only view 2 goes through the trunks and
        TemporalNet's head runs on ONE view: three heads per launch."""
        L, sp, tp = self.L, self.sp, self.tp
        lead = 0 if s == 0 else 1
        nv = len(self.views)

        def temporal_side():
            f = L.run_trunk_body(xb, tp['s1'])                # TemporalNet features [nv*b,45,60,128], view-major
            cv_t = torch.empty((nv, b, f.shape[1], f.shape[2], 52), device=f.device, dtype=torch.float32)
            for i in range(nv):
                fi = f[i * b:(i + 1) * b]
                if lead:
                    ops.cost_volume(self.carry[i], fi[0:1], 3, out=cv_t[i, 0:1])
                else:
                    ops.fill(cv_t[i, 0])
                if b > 1:
                    ops.cost_volume(fi[:b - 1], fi[1:], 3, out=cv_t[i, 1:])
            return f, cv_t

        def spatial_trunk():
            f64 = L.run_trunk_body(xa, sp['s1'])
            f32 = L.run_stage2(f64, sp['s2'])
            if self.cache2 is not None:
                self.cache2.append((f64[b:], f32[b:]) if cache1 is None else (f64, f32))
            return f64, f32

        def spatial_chain(f64, f32):
            if cache1 is not None:                           # view 1's features from the earlier pair, view 2's from this pass
                return self.spatial_net.forward_pair_cv(cache1[0], f64, cache1[1], f32, LR_H, LR_W)
            return self.spatial_net.forward_pair_cv(f64[:b], f64[b:], f32[:b], f32[b:], LR_H, LR_W)

        if QUAD_OVERLAP:
            # SpatialNet's chain behind its trunk is a string of small dependent launches (CCL, regressNet1 on 23 x 30 .. 5 x 7
            # maps, decomposition, feature warps) that leave most of the chip idle; TemporalNet's trunk -- full-chip launches
            # with no dependence on them -- runs beside it on a second HIP stream (its own hardware queue: GPU_MAX_HW_QUEUES)
            main = torch.cuda.current_stream(xb.device)
            side = _side_stream(xb.device)
            f64, f32 = spatial_trunk()                     # (both trunks are full-chip launches: one after the other)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                f, cv_t = temporal_side()
            off1, cv_s = spatial_chain(f64, f32)
            main.wait_stream(side)
            xb.record_stream(side)                         # allocated on the main stream, read on the side stream
            f.record_stream(main)                          # allocated on the side stream, read on the main stream from here on
            cv_t.record_stream(main)
        else:
            off1, cv_s = spatial_chain(*spatial_trunk())
            f, cv_t = temporal_side()
        off_ref = torch.empty((b, 126), device=f.device, dtype=torch.float32)
        off_tgt = torch.empty((b, 126), device=f.device, dtype=torch.float32)
        L.run_regressor_quad(cv_s, cv_t, L.get_quad(self.spatial_net, self.temporal_net, nv),
                             [off_ref, off_tgt] + [self.tm[i, e - b:e].view(b, -1) for i in range(nv)])
        if not lead:
            for i in range(nv):
                ops.fill(self.tm[i, 0])
        ops.spatial_meshes(off1, off_ref, off_tgt, LR_H, LR_W, out=(self.m1[s:e], self.m2[s:e]))
        self._carry(f, b, nv, e)

    def _carry(self, f, b, nv, e):
        # the last frame's features of every view in a small buffer of their own: a slice of `f` would keep the whole chunk's
        # [nv*b,45,60,128] features alive through the next push
        if e < self.n:                                   # (the last chunk carries nothing)
            if self.carry is None:
                self.carry = [torch.empty((1,) + tuple(f.shape[1:]), device=f.device, dtype=torch.float32) for _ in range(nv)]
            for i in range(nv):
                self.carry[i].copy_(f[i * b + b - 1:(i + 1) * b])

    def _spatial(self, xa, b, s, e, cache1=None):
        L, sp = self.L, self.sp
        f64 = L.run_trunk_body(xa, sp['s1'])
        f32 = L.run_stage2(f64, sp['s2'])
        if cache1 is None:
            f64_1, f32_1, f64_2, f32_2 = f64[:b], f32[:b], f64[b:], f32[b:]
        else:
            (f64_1, f32_1), f64_2, f32_2 = cache1, f64, f32
        off1, off_ref, off_tgt = self.spatial_net.forward_pair(f64_1, f64_2, f32_1, f32_2, LR_H, LR_W)
        if self.cache2 is not None:
            self.cache2.append((f64_2, f32_2))
        ops.spatial_meshes(off1, off_ref, off_tgt, LR_H, LR_W, out=(self.m1[s:e], self.m2[s:e]))

    def _temporal(self, xb, b, s, e):
        """xb: the TemporalNet stem output of the views in self.views, view-major [nv*b,90,120,64]."""
        L, tp = self.L, self.tp
        nv = len(self.views)
        f = L.run_trunk_body(xb, tp['s1'])              # [nv*b,45,60,128], view-major
        lead = 0 if s == 0 else 1                       # the pair (last frame of the previous chunk, first of this one)
        rows = b - 1 + lead
        if rows > 0:
            cv = torch.empty((nv * rows, f.shape[1], f.shape[2], 52), device=f.device, dtype=torch.float32)
            slices = []
            for i in range(nv):
                fi = f[i * b:(i + 1) * b]
                if lead:
                    ops.cost_volume(self.carry[i], fi[0:1], 3, out=cv[i * rows:i * rows + 1])
                if b > 1:
                    ops.cost_volume(fi[:b - 1], fi[1:], 3, out=cv[i * rows + lead:(i + 1) * rows])
                slices.append((i * rows, (i + 1) * rows, self.tm[i, e - rows:e].view(rows, -1)))
            L.run_regressor(cv, tp['r2'], out_slices=slices)
        self._carry(f, b, nv, e)

    def result(self):
        """-> (smotion1, smotion2, tmotion1, tmotion2), each [N,7,9,2] (tmotion frame 0 = 0)."""
        if self.pos != self.n:
            raise ValueError('%d of %d frames pushed' % (self.pos, self.n))
        self.carry = None
        if self.tmotion1 is not None:
            return self.m1, self.m2, self.tmotion1, self.tm[0]
        return self.m1, self.m2, self.tm[0], self.tm[1]


@torch.no_grad()
def joint_stage(spatial_net, temporal_net, lr1, lr2, chunk=None, tmotion1=None, cache2=None, cache1=None):
    """SpatialNet and TemporalNet of a resident 2-view clip in one sweep (JointEstimator fed in chunks of `chunk` frames).
    cache1: per chunk (same chunking) view 1's SpatialNet trunk features kept by an earlier pair (with tmotion1).
    -> (smotion1, smotion2, tmotion1, tmotion2), each [N,7,9,2] (tmotion frame 0 = 0)."""
    chunk = chunk or SPATIAL_CHUNK
    n = lr2.shape[0]
    est = JointEstimator(spatial_net, temporal_net, n, lr2.device, tmotion1, cache2)
    for i, s in enumerate(range(0, n, chunk)):
        est.push(None if cache1 is not None else lr1[s:s + chunk], lr2[s:s + chunk], None if cache1 is None else cache1[i])
    return est.result()


@torch.no_grad()
def smooth_stage(smooth_net, s1, s2, t1, t2):
    """Stage 3 of test() (test_online_tra.py:309-392) on the whole stream's motions [N,7,9,2]: tsmotion composition, all
    sliding SmoothNet windows, the clip's tensors (mesh-sized; the windows run in chunks of smooth_network.WINDOW_CHUNK)."""
    n = s1.shape[0]
    smesh1, tsm1 = ops.tsmotion(s1, t1, LR_H, LR_W)
    smesh2, tsm2 = ops.tsmotion(s2, t2, LR_H, LR_W)
    nw = n - (WINDOW - 1)
    delta = smooth_net.window_deltas(smesh1, smesh2, tsm1, tsm2, nw, WINDOW, 1, 1)
    out = ops.smooth_stitch(smesh1, smesh2, tsm1, tsm2, delta, nw, WINDOW)
    out['smotion1'], out['smotion2'], out['tmotion1'], out['tmotion2'] = s1, s2, t1, t2
    out['tsmotion1'], out['tsmotion2'] = tsm1, tsm2
    return out


@torch.no_grad()
def estimate_meshes(nets, lr1, lr2, tmotion1=None, spatial_cache1=None, keep_spatial_cache2=False):
    """Stages 1-3 of test() (test_online_tra.py:284-392) for one clip.
    lr1, lr2: [N,3,360,480] device tensors (or lists of [1,3,360,480]).
    -> dict(smooth_mesh1/2, ori_mesh1/2 [1,N,7,9,2], ori_path2, smooth_path2 stitched as test_metric_ssd.py:433-436)."""
    spatial_net, temporal_net, smooth_net = nets
    dev = next(spatial_net.parameters()).device
    if isinstance(lr1, (list, tuple)):
        lr1 = torch.cat([t.to(dev) for t in lr1], 0)
        lr2 = torch.cat([t.to(dev) for t in lr2], 0)
    n = lr1.shape[0]
    if n < WINDOW:
        raise ValueError('need at least %d frames for the sliding smooth window, got %d' % (WINDOW, n))
    cache2 = [] if keep_spatial_cache2 else None
    if OVERLAP_STREAMS:
        # SpatialNet and TemporalNet are independent until tsmotion: run them on two HIP streams so that the
        # partially filled last round of one net's kernels is topped up with the other's workgroups
        main = torch.cuda.current_stream(dev)
        side = _side_stream(dev)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            t1, t2 = temporal_stage_views(temporal_net, [lr1, lr2])
        s1, s2 = spatial_stage(spatial_net, lr1, lr2)
        main.wait_stream(side)
        for t in (t1, t2):
            t.record_stream(main)
    elif SHARED_STEM and (tmotion1 is None) == (spatial_cache1 is None):
        # both nets behind one stem; with view 1's motions AND trunk features known (second pair of a three-view clip) only
        # view 2 goes through it
        s1, s2, t1, t2 = joint_stage(spatial_net, temporal_net, lr1, lr2, tmotion1=tmotion1, cache2=cache2, cache1=spatial_cache1)
    else:
        s1, s2 = spatial_stage(spatial_net, lr1, lr2, cache1=spatial_cache1)
        if tmotion1 is None:
            t1, t2 = temporal_stage_views(temporal_net, [lr1, lr2])
        else:
            t1, t2 = tmotion1, temporal_stage_views(temporal_net, [lr2])[0]
    out = smooth_stage(smooth_net, s1, s2, t1, t2)
    if cache2:
        out['spatial_cache2'] = cache2
    return out


def _stitch_windows(o):
    """torch restatement of ops.smooth_stitch on per-window SmoothNet outputs [nw,7,7,9,2] (kept for the tests: the
    kernel is checked against it).  Window 0 contributes its 7 frames, every later window its last frame
    (test_online_tra.py:377-392); the metric harness's paths are chained across windows as test_metric_ssd.py:427-436 does."""
    out = {}
    for k in ('ori_mesh1', 'ori_mesh2', 'smooth_mesh1', 'smooth_mesh2'):
        out[k] = torch.cat((o[k][0], o[k][1:, -1]), 0).unsqueeze(0)
    op, sp = o['ori_path2'], o['smooth_path2']
    inc = op[1:, -1] - op[1:, -2]
    ori_path = torch.cat((op[0], op[0, -1:] + torch.cumsum(inc, 0)), 0)
    smooth_path = torch.cat((sp[0], ori_path[WINDOW:] + (sp[1:, -1] - op[1:, -1])), 0)
    out['ori_path2'] = ori_path.unsqueeze(0)
    out['smooth_path2'] = smooth_path.unsqueeze(0)
    return out


# ------------------------------------------------------------------ checkpoints
def load_nets(model_dir, device='cuda'):
    """test_online_tra.py:164-201: build the three networks, require EXACTLY three `*.pth` files in `model_dir`
    (spatial_warp.pth, temporal_warp.pth, smooth_warp.pth), load `torch.load(p)['model']` strictly, eval mode.
    -> (spatial_net, temporal_net, smooth_net) on `device`.  Raises FileNotFoundError where the reference prints
    'No checkpoint found!' and exits."""
    from .spatial_network import SpatialNet
    from .temporal_network import TemporalNet
    from .smooth_network import SmoothNet
    ckpt_list = sorted(glob.glob(os.path.join(model_dir, '*.pth')))
    if len(ckpt_list) != 3:
        raise FileNotFoundError('No checkpoint found! %s holds %d *.pth files, expected spatial_warp.pth, '
                                'temporal_warp.pth and smooth_warp.pth' % (model_dir, len(ckpt_list)))
    nets = []
    for cls, name in ((SpatialNet, 'spatial_warp.pth'), (TemporalNet, 'temporal_warp.pth'), (SmoothNet, 'smooth_warp.pth')):
        net = cls()
        ck = torch.load(os.path.join(model_dir, name), map_location='cpu')
        net.load_state_dict(ck['model'])
        nets.append(net.to(device).eval())
    return tuple(nets)


def find_model_dir(root):
    """The reference's checkpoint locations relative to its tree (Full_model_inference/README.md:2-6):
    full_model_tra first (test_online_tra.py), then full_model_ssd; None when neither holds three *.pth files."""
    for sub in ('full_model_tra', 'full_model_ssd'):
        d = os.path.join(root, 'Full_model_inference', sub)
        if len(glob.glob(os.path.join(d, '*.pth'))) == 3:
            return d
    return None


_nrigid_cache = {}


def norm_rigid_mesh(img_h, img_w, device):
    """get_norm_mesh(get_rigid_mesh(1, h, w)) as a cached device constant ([1,63,2]; built once per (size, device)
    instead of on the host for every render)."""
    device = torch.device(device)
    key = (int(img_h), int(img_w), str(device))
    ent = _nrigid_cache.get(key)
    if ent is None:
        ent = ops._Built(lambda: get_norm_mesh(get_rigid_mesh(1, img_h, img_w, device=device), img_h, img_w).contiguous(),
                         device, 'the cached normalised rigid mesh')
        _nrigid_cache[key] = ent
    return ent.get(device)


# AVERAGE render: skip a view's 63-term spline on canvas tiles it provably cannot reach (ops.render_footprints) and take
# its contribution there as exactly 0 instead of the rounding residue of the reference's clamped sampler.
# SS_SKIP_OUTSIDE=0: evaluate every view at every pixel.
SKIP_OUTSIDE = os.environ.get('SS_SKIP_OUTSIDE', '1') == '1'


# ------------------------------------------------------------------ render
@torch.no_grad()
def canvas_bbox(meshes, img_h, img_w, prescaled=False, bbox=None):
    """(wmin, wmax, hmin, hmax) over `meshes` in HR pixels as a device tensor [4] (test_online_tra.py:103-120); `bbox`:
    an existing box to fold in (more chunks of the same video)."""
    sh, sw = (0.0, 0.0) if prescaled else (img_h, img_w)
    return ops.mesh_bbox(list(meshes), sh, sw, bbox)


def canvas_size(bbox):
    """bbox device tensor [4] -> (Hc, Wc): the path's one host sync (the data-dependent canvas size,
    test_online_tra.py:122-123: fp32 extents `wmax - wmin`, `hmax - hmin`, truncated by `.int()`)."""
    import numpy as np
    bb = bbox.cpu().numpy()                        # one 16-byte device -> host copy
    return int(np.float32(bb[3]) - np.float32(bb[2])), int(np.float32(bb[1]) - np.float32(bb[0]))


@torch.no_grad()
def render_plan(meshes, img_h, img_w, prescaled=False, bbox=None, size=None):
    """Canvas + TPS coefficients for every (frame, view).
    meshes: list of V tensors [1,N,7,9,2] (LR scale, or HR canvas pixels with prescaled=True).
    bbox: the canvas box to render onto (device [4]; default: the box of these meshes); size: its (Hc, Wc) when the
    caller already read it back (no host sync then).
    -> (Hc, Wc, source [N,V,63,2], T [N,V,2,66])."""
    dev = meshes[0].device
    v = len(meshes)
    n = meshes[0].shape[1]
    sh, sw = (0.0, 0.0) if prescaled else (img_h, img_w)
    if bbox is None:
        bbox = canvas_bbox(meshes, img_h, img_w, prescaled)
    hc, wc = size if size is not None else canvas_size(bbox)
    src = ops.mesh_normalize_views(meshes, bbox, sh, sw)                          # [N,V,63,2]
    T = ops.tps_solve_shared(src.view(n * v, 63, 2), norm_rigid_mesh(img_h, img_w, dev)).view(n, v, 2, 66)
    return hc, wc, src, T


def _as_clip(x, n):
    """A view's frames as ONE contiguous device tensor [n,3,h,w] when they already are one (no copy), else None."""
    if torch.is_tensor(x) and x.dim() == 4 and x.shape[0] == n and x.is_cuda and x.is_contiguous() and x.dtype == torch.float32:
        return x
    return None


# LINEAR fusion of clips held as tensors: three launches per clip (ops.render_linear_clip) instead of eight per frame.
# SS_LINEAR_CLIP=0: the per-frame chain (tps_warp_views + linear_blend), which lists of separate frames always take.
LINEAR_CLIP = os.environ.get('SS_LINEAR_CLIP', '1') == '1'
LINEAR_CLIP_FRAMES = int(os.environ.get('SS_LINEAR_CLIP_FRAMES', '64'))     # frames per launch group (workspace: 44 MB per 720p frame and view)


def _linear_clip(clips, src, T, hc, wc, warp_mode, out):
    n = src.shape[0]
    for s in range(0, n, LINEAR_CLIP_FRAMES):
        e = min(s + LINEAR_CLIP_FRAMES, n)
        ops.render_linear_clip([c[s:e] for c in clips], src[s:e], T[s:e], hc, wc, warp_mode, out=out[s:e])
    return out


@torch.no_grad()
def render_frames(img_lists, meshes, warp_mode='NORMAL', fusion_mode='AVERAGE', out=None, prescaled=False, bbox=None,
                  size=None):
    """img_lists: V lists (or tensors [N,3,H,W]) of HR frames (0..255); meshes: V tensors [1,N,7,9,2].
    -> (frames [N,3,Hc,Wc] device tensor, Hc, Wc).  AVERAGE fusion of clips held as tensors is ONE launch for the whole
    clip (ops.render_average_clip); lists of separate frames and LINEAR fusion go frame by frame.
    With SKIP_OUTSIDE (default) a view contributes exactly 0 on canvas tiles it cannot reach, where the reference's
    clamped sampler leaves a rounding residue of <~ 1e-2 grey levels (ops.render_average)."""
    v = len(img_lists)
    dev = meshes[0].device
    n = meshes[0].shape[1]
    first = img_lists[0][0]
    img_h, img_w = first.shape[-2:]
    hc, wc, src, T = render_plan(meshes, img_h, img_w, prescaled, bbox, size)
    if out is None or tuple(out.shape) != (n, 3, hc, wc) or not out.is_contiguous():
        out = torch.empty((n, 3, hc, wc), device=dev, dtype=torch.float32)      # (a buffer of another canvas is not reused)
    fp = ops.render_footprints(src, T, img_h, img_w, hc, wc) if (SKIP_OUTSIDE and fusion_mode == 'AVERAGE') else None
    clips = [_as_clip(x, n) for x in img_lists]
    if fusion_mode == 'AVERAGE' and all(c is not None for c in clips):
        ops.render_average_clip(clips, src, T, hc, wc, warp_mode, out=out, footprint=fp)
        return out, hc, wc
    if fusion_mode == 'LINEAR' and LINEAR_CLIP and all(c is not None for c in clips):
        _linear_clip(clips, src, T, hc, wc, warp_mode, out)
        return out, hc, wc
    for i in range(n):
        imgs = [img_lists[k][i].to(dev, non_blocking=True) for k in range(v)]
        if fusion_mode == 'AVERAGE':
            ops.render_average(imgs, src[i], T[i], hc, wc, warp_mode, out=out[i], footprint=None if fp is None else fp[i])
        else:
            w = ops.tps_warp_views(imgs, src[i], T[i], hc, wc, warp_mode)            # [V,4,Hc,Wc]
            if v == 2:
                ops.linear_blend(w[0, 0:3], w[1, 0:3], w[0, 3], w[1, 3], out=out[i])
            else:
                f = ops.linear_blend(w[0, 0:3], w[1, 0:3], w[0, 3], w[1, 3])
                ops.linear_blend(f, w[2, 0:3], ops.mask_union(w[0, 3], w[1, 3]), w[2, 3], out=out[i])
    return out, hc, wc


def get_stable_sqe(img1_list, img2_list, smooth_mesh1, smooth_mesh2, warp_mode, fusion_mode):
    """test_online_tra.py:96-154 -> (list of ndarray [Hc,Wc,3] fp32, Wc, Hc) like the reference."""
    frames, hc, wc = render_frames([img1_list, img2_list], [smooth_mesh1, smooth_mesh2], warp_mode, fusion_mode)
    host = frames.permute(0, 2, 3, 1).cpu().numpy()
    return [host[i] for i in range(host.shape[0])], torch.tensor(wc, dtype=torch.int32), \
        torch.tensor(hc, dtype=torch.int32)


@torch.no_grad()
def run_two_view(hr1, hr2, lr1, lr2, nets, warp_mode='NORMAL', fusion_mode='AVERAGE', to_host=False, out=None, deterministic=False):
    """-> (frames, Hc, Wc, smooth_mesh1, smooth_mesh2); frames = device tensor [N,3,Hc,Wc]
    (or list of HWC ndarrays with to_host=True).  out: the frames tensor of a previous call to render into; it is reused
    when the canvas still has that size (same-size clips of one stream), else a new one is allocated.
    deterministic: the conv engine's geometry-only kernel policy (ops.deterministic): a frame's bits do not depend on how many
    frames share its launches -- resident clip == chunked passes == stream."""
    with ops.deterministic(deterministic):
        acc = estimate_meshes(nets, lr1, lr2)
    frames, hc, wc = render_frames([hr1, hr2], [acc['smooth_mesh1'], acc['smooth_mesh2']], warp_mode, fusion_mode, out=out)
    if to_host:
        host = frames.permute(0, 2, 3, 1).cpu().numpy()
        frames = [host[i] for i in range(host.shape[0])]
    return frames, hc, wc, acc['smooth_mesh1'], acc['smooth_mesh2']


# ------------------------------------------------------------------ three-view (threeview:345-505)
@torch.no_grad()
def three_view_compose(w12_m1, w12_m2, w23_m1, w23_m2, img_h, img_w, first_canvas=None):
    """Mesh alignment, middle plane and TPS re-projection of the outer views (test_online_tra_threeview.py:345-420).
    Inputs [1,N,7,9,2] (LR scale) -> (mesh1, middle, mesh3) in first-canvas HR pixels.  All of it on the HIP kernels:
    `ss_three_view_align` (scale, per-frame mean offset, middle mesh), `ss_mesh_bbox` / `ss_mesh_normalize` (first canvas),
    `ss_tps_solve` / `ss_tps_points` (re-projection), `ss_three_view_finish` (back to canvas pixels); no host sync.
    first_canvas: device box [4] to use as the first canvas instead of the box of these frames (streaming mode: fixed once)."""
    a1, a2, b1, b2, mid = ops.three_view_align(w12_m1, w12_m2, w23_m1, w23_m2, img_h, img_w)
    bbox = first_canvas if first_canvas is not None else ops.mesh_bbox([a1, a2, b1, b2], 0.0, 0.0)   # (meshes are HR pixels already)
    # both re-projections (view 1 through pair (1,2)'s spline, view 3 through pair (2,3)'s) as ONE normalisation launch, ONE batched
    # TPS solve of 2 N systems and ONE point evaluation (round 6: they were 5 + 2 + 2 launches; a solve is 47 us of one workgroup's
    # latency whether the launch holds one system or many) -- the same arithmetic per system
    nrm = ops.three_view_normalize(a1, a2, b1, b2, mid, bbox)       # [6,N,63,2] = {a1, b2 | a2, b1 | mid, mid}
    n = nrm.shape[1]
    src = nrm[2:4].reshape(2 * n, 63, 2)
    pts = ops.tps_points(nrm[0:2].reshape(2 * n, 63, 2), src, ops.tps_solve(src, nrm[4:6].reshape(2 * n, 63, 2)))
    return tuple(ops.three_view_finish(pts[:n], pts[n:], mid, bbox))


@torch.no_grad()
def three_view_render(img1, img2, img3, mesh1, middle, mesh3, warp_mode='NORMAL', fusion_mode='AVERAGE', out=None):
    """Meshes are HR-scale canvas pixels here (output of three_view_compose)."""
    return render_frames([img1, img2, img3], [mesh1, middle, mesh3], warp_mode, fusion_mode, out=out, prescaled=True)


@torch.no_grad()
def run_three_view(hr1, hr2, hr3, lr1, lr2, lr3, nets, warp_mode='NORMAL', fusion_mode='AVERAGE', out=None, deterministic=False):
    # the middle view's TemporalNet motions and SpatialNet trunk features are computed once and reused by pair (2,3)
    with ops.deterministic(deterministic):
        a12 = estimate_meshes(nets, lr1, lr2, keep_spatial_cache2=True)
        a23 = estimate_meshes(nets, lr2, lr3, tmotion1=a12['tmotion2'], spatial_cache1=a12.get('spatial_cache2'))
    img_h, img_w = hr1[0].shape[-2:]
    m1, mid, m3 = three_view_compose(a12['smooth_mesh1'], a12['smooth_mesh2'], a23['smooth_mesh1'],
                                     a23['smooth_mesh2'], img_h, img_w)
    frames, hc, wc = three_view_render(hr1, hr2, hr3, m1, mid, m3, warp_mode, fusion_mode, out=out)
    return frames, hc, wc, m1, mid, m3


# ------------------------------------------------------------------ frame I/O front-end / sink
@torch.no_grad()
def load_frames_u8(frames, lr_h=360, lr_w=480, device=None):
    """test_online_tra.py:250-278 for one view on the device: decoded uint8 frames [N,H,W,3] (ndarray or tensor,
    cv2 channel order) -> (hr [N,3,H,W] fp32 0..255, lr [N,3,lr_h,lr_w] fp32 in [-1,1] via the cv2-exact resize).
    Only the uint8 bytes cross PCIe (2.8 MB per 720p frame instead of 11 MB of fp32)."""
    if not torch.is_tensor(frames):
        frames = torch.from_numpy(frames)
    if device is not None:
        frames = frames.to(device, non_blocking=True)
    return ops.ingest_u8(frames.contiguous(), lr_h, lr_w)


@torch.no_grad()
def to_video_frames(frames, to_host=False):
    """`stable_list[k].astype(np.uint8)` (test_online_tra.py:413) for a whole clip on the device:
    [N,3,Hc,Wc] fp32 -> uint8 [N,Hc,Wc,3] (what cv2.VideoWriter.write consumes); to_host=True returns an ndarray."""
    u8 = ops.canvas_to_u8(frames.contiguous())
    return u8.cpu().numpy() if to_host else u8


# uint8 end to end: the AVERAGE render samples the decoded uint8 frames and writes the uint8 video frame itself
# (ss_render_average_u8) -- per 720p frame pair 22 MB of fp32 frame planes and 17 + 17 MB of fp32 canvas traffic less,
# bit-identical output.  SS_U8_FUSED=0 (or fusion LINEAR) goes through fp32 planes / canvas + ss_canvas_to_u8.
U8_FUSED = os.environ.get('SS_U8_FUSED', '1') == '1'


def _u8_fused(fusion_mode):
    return U8_FUSED and (fusion_mode == 'AVERAGE' or (fusion_mode == 'LINEAR' and LINEAR_CLIP))


@torch.no_grad()
def render_frames_u8(frame_lists, meshes, warp_mode='NORMAL', out=None, bbox=None, size=None, prescaled=False,
                     fusion_mode='AVERAGE'):
    """frame_lists: V device tensors [N,H,W,3] uint8; meshes: V tensors [1,N,7,9,2] -> (uint8 [N,Hc,Wc,3], Hc, Wc):
    `render_frames(..., fusion_mode)` followed by `to_video_frames`, fused: one launch for the clip (AVERAGE), three / four
    (LINEAR: the warped planes are fp32 either way, the blend writes the video frame)."""
    n = meshes[0].shape[1]
    img_h, img_w = frame_lists[0].shape[1], frame_lists[0].shape[2]
    hc, wc, src, T = render_plan(meshes, img_h, img_w, prescaled, bbox=bbox, size=size)
    if callable(out):                                   # a buffer provider (HostClipRunner's ring of result buffers)
        out = out((n, hc, wc, 3))
    if out is None or tuple(out.shape) != (n, hc, wc, 3) or not out.is_contiguous():
        out = torch.empty((n, hc, wc, 3), device=meshes[0].device, dtype=torch.uint8)
    if fusion_mode == 'LINEAR':
        _linear_clip([f if f.is_contiguous() else f.contiguous() for f in frame_lists], src, T, hc, wc, warp_mode, out)
        return out, hc, wc
    fp = ops.render_footprints(src, T, img_h, img_w, hc, wc) if SKIP_OUTSIDE else None
    ops.render_average_clip_u8([f if f.is_contiguous() else f.contiguous() for f in frame_lists], src, T, hc, wc, warp_mode,
                               out=out, footprint=fp)
    return out, hc, wc


def _as_device_u8(frames, device):
    if not torch.is_tensor(frames):
        frames = torch.from_numpy(frames)
    return frames.to(device, non_blocking=True).contiguous()


@torch.no_grad()
def run_two_view_u8(frames1, frames2, nets, warp_mode='NORMAL', fusion_mode='AVERAGE', device='cuda', to_host=False):
    """uint8 in, uint8 out: ingest -> estimate -> render -> video frames.  -> (uint8 [N,Hc,Wc,3], Hc, Wc, m1, m2)."""
    if _u8_fused(fusion_mode):
        f1, f2 = _as_device_u8(frames1, device), _as_device_u8(frames2, device)
        _, lr1 = ops.ingest_u8(f1, want_hr=False)
        _, lr2 = ops.ingest_u8(f2, want_hr=False)
        acc = estimate_meshes(nets, lr1, lr2)
        m1, m2 = acc['smooth_mesh1'], acc['smooth_mesh2']
        u8, hc, wc = render_frames_u8([f1, f2], [m1, m2], warp_mode, fusion_mode=fusion_mode)
        return (u8.cpu().numpy() if to_host else u8), hc, wc, m1, m2
    hr1, lr1 = load_frames_u8(frames1, device=device)
    hr2, lr2 = load_frames_u8(frames2, device=device)
    frames, hc, wc, m1, m2 = run_two_view(hr1, hr2, lr1, lr2, nets, warp_mode, fusion_mode)
    return to_video_frames(frames, to_host), hc, wc, m1, m2


@torch.no_grad()
def run_three_view_u8(frames1, frames2, frames3, nets, warp_mode='NORMAL', fusion_mode='AVERAGE', device='cuda', to_host=False):
    """`run_three_view` from decoded uint8 frames [N,H,W,3] to uint8 video frames (device-resident clip).
    -> (uint8 [N,Hc,Wc,3], Hc, Wc, mesh1, middle, mesh3)."""
    f = [_as_device_u8(x, device) for x in (frames1, frames2, frames3)]
    img_h, img_w = f[0].shape[1], f[0].shape[2]
    if _u8_fused(fusion_mode):
        lr = [ops.ingest_u8(x, want_hr=False)[1] for x in f]
        a12 = estimate_meshes(nets, lr[0], lr[1], keep_spatial_cache2=True)
        a23 = estimate_meshes(nets, lr[1], lr[2], tmotion1=a12['tmotion2'], spatial_cache1=a12.get('spatial_cache2'))
        ms = three_view_compose(a12['smooth_mesh1'], a12['smooth_mesh2'], a23['smooth_mesh1'], a23['smooth_mesh2'], img_h, img_w)
        u8, hc, wc = render_frames_u8(f, list(ms), warp_mode, prescaled=True, fusion_mode=fusion_mode)
        return (u8.cpu().numpy() if to_host else u8), hc, wc, ms[0], ms[1], ms[2]
    io = [ops.ingest_u8(x) for x in f]
    frames, hc, wc, m1, mid, m3 = run_three_view(io[0][0], io[1][0], io[2][0], io[0][1], io[1][1], io[2][1], nets, warp_mode,
                                                 fusion_mode)
    return to_video_frames(frames, to_host), hc, wc, m1, mid, m3


def io_streams(dev):
    """(upload, compute, download) HIP streams of a host-fed runner.  The three must sit on DIFFERENT hardware queues: the
    HIP runtime multiplexes all streams of a process onto GPU_MAX_HW_QUEUES (default 4) AQL queues, and two streams that
    share one execute in order -- an upload enqueued ahead of the current clip's kernels then holds them back for its whole
    PCIe time (measured, tools/diag_overlap.py: 9.4 ms per clip with a queue each, 12.1 with one copy stream on the compute
    queue, 15.3 with both; which streams collide depends on how many streams the process used before).  The package
    therefore raises GPU_MAX_HW_QUEUES to 16 at import (stabstitch2_amd/__init__.py; only effective before the HIP
    runtime initialises) -- `HostClipRunner.copy_stats()` reports what the copies achieved beside the compute."""
    return tuple(torch.cuda.Stream(dev) for _ in range(3))


class HostClipRunner:
    """uint8 clips in pinned host memory -> stitched uint8 clips in pinned host memory, with the PCIe copies of
    neighbouring clips hidden behind the compute of the current one (three HIP streams: upload, compute, download).

        for video, hc, wc in HostClipRunner(nets).run(clips):     # clips yields (frames1, frames2) uint8 [N,H,W,3]
            writer.write(video[k]) ...

    The upload of clip k+1 is enqueued before the host starts issuing clip k's kernels (whose canvas-size read-back is
    the path's one host sync), the download of clip k runs while clip k+1 computes.  Yields (ndarray-backed pinned
    uint8 tensor [N,Hc,Wc,3], Hc, Wc) one clip late at most; a yielded tensor stays valid until `depth` more clips
    have been yielded."""

    def __init__(self, nets, device='cuda', warp_mode='NORMAL', fusion_mode='AVERAGE', depth=2, streams=None, prefetch=2,
                 numa_bind=False):
        """numa_bind: True = pin the CALLING thread (and the threads it starts later) to the CPUs of this GPU's NUMA node and prefer
        that node for memory before the pinned buffers are allocated (hostbind.bind_to_gpu: process-affecting, so it is the
        application's decision -- bench.py binds every rank itself; measured irrelevant on the MI355X hosts of this pool)."""
        self.nets, self.dev = nets, torch.device(device)
        self.warp_mode, self.fusion_mode, self.depth = warp_mode, fusion_mode, depth
        self.up, self.comp, self.down = streams if streams is not None else io_streams(self.dev)
        # depth + 2 pinned result slots: the download of clip k + depth + 1 is enqueued before clip k + depth is yielded, so a
        # tensor yielded for clip k is only overwritten after `depth` more clips have been handed out
        self._host = [None] * (depth + 2)
        # clips uploaded ahead of the one being computed.  One is enough when H2D runs at its 55 GB/s (3.2 of a clip's 8.5 ms),
        # but beside the compute the copy rate of single clips drops to 15-30 GB/s now and then (bench: run totals 3070-3570
        # frames/s at an unchanged 3745 steady state); with two the next clip's frames have two clip periods to arrive
        self.prefetch = prefetch
        self.timed = False                   # True: HIP events around every upload / download (copy_stats)
        self._copies = {'h2d': [], 'd2h': []}
        # Device staging owned by the runner (round 4): rings of `prefetch + 2` uploaded clips and 3 result buffers.  Taking
        # them from the caching allocator per clip meant blocks of three streams' pools, released only when the recorded
        # events had passed: now and then a clip needed a fresh 88-130 MB hipMalloc inside a run (10-20 ms each -- the
        # 3070-3570 frames/s spread of the bench's run totals at an unchanged steady state).
        self._in, self._in_free, self._in_k, self._in_gen = [], [], 0, 0
        self._out, self._out_done, self._out_k = [], [], 0
        # host placement (hostbind): opt-in, before anything is pinned; a process that bound itself to THIS device already (bench.py
        # does, per rank) is left alone
        if numa_bind and self.dev.type == 'cuda':
            from . import hostbind
            if hostbind.report(self.dev) is None:
                hostbind.bind_to_gpu(self.dev)

    class _Staged(list):
        slot = None
        gen = 0

    def _in_slot(self, shapes):
        nbuf = self.prefetch + 2
        ok = bool(self._in) and len(self._in[0]) == len(shapes) and all(
            tuple(b.shape[1:]) == tuple(sh[1:]) and b.shape[0] >= sh[0] for b, sh in zip(self._in[0], shapes))
        if not ok:
            torch.cuda.synchronize(self.dev)             # new frame geometry (or a longer clip): nothing of the old ring is in flight
            same = bool(self._in) and len(self._in[0]) == len(shapes) and all(
                tuple(b.shape[1:]) == tuple(sh[1:]) for b, sh in zip(self._in[0], shapes))
            cap = max([sh[0] for sh in shapes] + ([self._in[0][0].shape[0]] if same else []))
            self._in = [[torch.empty((cap,) + tuple(sh[1:]), dtype=torch.uint8, device=self.dev) for sh in shapes]
                        for _ in range(nbuf)]
            self._in_free = [None] * nbuf
            self._in_gen += 1                            # clips already staged in the OLD ring keep it alive (see `consumed`)
        j = self._in_k % len(self._in)
        self._in_k += 1
        return j

    def consumed(self, d, ev):
        """The compute that reads the uploaded clip `d` has been enqueued; `ev` (recorded behind it) frees d's staging slot.
        A clip staged in a ring that has been replaced since (a longer clip or another geometry arrived while it waited in the
        prefetch queue) has no slot in the new ring: its buffers are handed back to the allocator, which must not reuse them
        before the compute stream's readers are done -- record_stream, not the new ring's events."""
        if d.gen != self._in_gen:
            for t in d:
                t.record_stream(self.comp)
            return
        self._in_free[d.slot] = ev

    def out_buffer(self, shape):
        """The next result buffer of the ring (uint8 [n,Hc,Wc,3]); the compute stream waits for the download that last read it."""
        ok = bool(self._out) and tuple(self._out[0].shape[1:]) == tuple(shape[1:]) and self._out[0].shape[0] >= shape[0]
        if not ok:
            torch.cuda.synchronize(self.dev)
            same = bool(self._out) and tuple(self._out[0].shape[1:]) == tuple(shape[1:])
            cap = max(shape[0], self._out[0].shape[0] if same else 0)
            self._out = [torch.empty((cap,) + tuple(shape[1:]), dtype=torch.uint8, device=self.dev) for _ in range(3)]
            self._out_done = [None] * 3
        j = self._out_k % 3
        self._out_k += 1
        if self._out_done[j] is not None:
            torch.cuda.current_stream(self.dev).wait_event(self._out_done[j])
        buf = self._out[j][:shape[0]]
        buf._ss_slot = j
        return buf

    def _timed_copy(self, kind, stream, nbytes, fn):
        if not self.timed:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        r = fn()
        e1.record(stream)
        self._copies[kind].append((e0, e1, nbytes))
        return r

    def copy_stats(self, reset=True):
        """With `timed` set: {'h2d_GBps', 'd2h_GBps', 'h2d_ms', 'd2h_ms'} of the copies enqueued since the last call, each
        measured on its own stream while the compute stream runs the neighbouring clip (call after a synchronize)."""
        out = {}
        for kind, evs in self._copies.items():
            ms = sum(a.elapsed_time(b) for a, b, _ in evs)
            nb = sum(n for _, _, n in evs)
            if evs and ms > 0:
                out[kind + '_GBps'] = round(nb / (ms * 1e-3) / 1e9, 1)
                out[kind + '_ms_per_clip'] = round(ms / len(evs), 3)
        if reset:
            self._copies = {'h2d': [], 'd2h': []}
        return out

    def _upload(self, clip):
        """uint8 frames of every view (host tensors / arrays [m,H,W,3]) -> (staged device tensors, ready event); the staging
        slot is reused once `consumed` has been told which event ends its last reader."""
        src = [(torch.from_numpy(f) if not torch.is_tensor(f) else f) for f in clip]
        j = self._in_slot([tuple(t.shape) for t in src])
        if self._in_free[j] is not None:
            # the slot's last reader ended `prefetch + 1` clips ago: settled on the HOST (returns at once) -- a GPU-side wait of the copy
            # stream on a compute event costs 0.13-0.17 ms of serialisation per edge (online.HostFrameStream, tools/diag_host_stream2.py)
            self._in_free[j].synchronize()
        with torch.cuda.stream(self.up):
            d = self._Staged(buf[:t.shape[0]] for buf, t in zip(self._in[j], src))
            d.slot = j
            d.gen = self._in_gen
            self._timed_copy('h2d', self.up, sum(t.numel() * t.element_size() for t in src),
                             lambda: [dst.copy_(t, non_blocking=True) for dst, t in zip(d, src)])
            ev = torch.cuda.Event()
            ev.record(self.up)
        return d, ev

    def _compute(self, d, ev):
        self.comp.wait_event(ev)
        with torch.cuda.stream(self.comp):
            if _u8_fused(self.fusion_mode):
                _, lr1 = ops.ingest_u8(d[0], want_hr=False)
                _, lr2 = ops.ingest_u8(d[1], want_hr=False)
                acc = estimate_meshes(self.nets, lr1, lr2)
                u8, hc, wc = render_frames_u8(list(d), [acc['smooth_mesh1'], acc['smooth_mesh2']], self.warp_mode,
                                              out=self.out_buffer, fusion_mode=self.fusion_mode)
            else:
                hr1, lr1 = ops.ingest_u8(d[0])
                hr2, lr2 = ops.ingest_u8(d[1])
                frames, hc, wc, _, _ = run_two_view(hr1, hr2, lr1, lr2, self.nets, self.warp_mode, self.fusion_mode)
                u8 = ops.canvas_to_u8(frames, out=self.out_buffer(tuple(frames.shape[0:1]) + (hc, wc, 3)))
            ev2 = torch.cuda.Event()
            ev2.record(self.comp)
        self.consumed(d, ev2)
        return u8, hc, wc, ev2

    def _download(self, k, u8, ev):
        m, frame = u8.shape[0], tuple(u8.shape[1:])
        h0 = self._host[0]
        if h0 is None or tuple(h0.shape[1:]) != frame or h0.shape[0] < m:
            # new canvas size (or a longer clip than any before): pin every slot now, not one clip at a time.  Slots are
            # sized for the longest clip seen; a shorter last chunk takes a slice (no re-pinning at the end of a video)
            cap = m if h0 is None or tuple(h0.shape[1:]) != frame else max(m, h0.shape[0])
            self._host = [torch.empty((cap,) + frame, dtype=torch.uint8).pin_memory() for _ in self._host]
        dst = self._host[k % len(self._host)][:m]
        self.down.wait_event(ev)
        with torch.cuda.stream(self.down):
            slot = getattr(u8, '_ss_slot', None)
            if slot is None:
                u8.record_stream(self.down)                # a tensor of the caller's, not of the ring
            self._timed_copy('d2h', self.down, u8.numel(), lambda: dst.copy_(u8, non_blocking=True))
            done = torch.cuda.Event()
            done.record(self.down)
            if slot is not None:
                self._out_done[slot] = done
        return dst, done

    @torch.no_grad()
    def run(self, clips):
        from collections import deque
        it = iter(clips)
        ahead = deque()                      # uploads in flight: `prefetch` clips beyond the one being computed

        def top_up():
            while len(ahead) < self.prefetch + 1:
                try:
                    ahead.append(self._upload(next(it)))
                except StopIteration:
                    return
        top_up()
        if not ahead:
            return
        pending = None
        k = 0
        while ahead:
            cur = ahead.popleft()
            top_up()
            u8, hc, wc, ev = self._compute(*cur)
            host, done = self._download(k, u8, ev)
            if pending is not None:
                pending[3].synchronize()
                yield pending[0], pending[1], pending[2]
            pending = (host, hc, wc, done)
            k += 1
        pending[3].synchronize()
        yield pending[0], pending[1], pending[2]


# ------------------------------------------------------------------ long videos, one global canvas, bounded device memory
class LongVideoStitcher:
    """The reference's whole-video behaviour for videos of any length, two or three views: it keeps every frame in host lists
    (test_online_tra.py:250-278), smooths over the whole sequence (:359-392) and renders every frame onto ONE canvas, the
    bounding box of all frames' meshes (:106-120; three views: test_online_tra_threeview.py:154-505) -- cutting a video into
    independent clips (HostClipRunner) does not reproduce that output.  Here the frames stay on the host (uint8 [N,H,W,3]
    arrays, memory-mapped files, ...) and go through the device twice, `chunk` frames at a time:
      pass 1  `estimate`: uint8 upload -> cv2-exact LR resize -> SpatialNet / TemporalNet (JointEstimator; a chunk boundary
              carries one feature map per view; three views: pair (2,3) takes the middle view's trunk features of the same
              chunk and its temporal motions from pair (1,2)) -> the stream's motions [N,7,9,2]; then tsmotion, all sliding
              SmoothNet windows, the three-view alignment and the global canvas box on those mesh-sized tensors;
      pass 2  `render`: uint8 upload -> fused TPS warp + fusion onto the shared canvas -> uint8 video frames -> host.
    Device memory is that of one chunk plus O(N) mesh-sized tensors (504 bytes per frame, view and motion kind); uploads,
    compute and downloads of neighbouring chunks overlap on three HIP streams.  The meshes equal those of the resident
    path bit for bit -- PROVIDED both run the same chunking (`chunk` = the resident path's 32-pair chunks): the conv engine picks its
    kernel per launch size (ss_conv_uses_wino43 / ss_conv_uses_winograd / split-K), and another chunk length sums in another order
    (~1e-5 px; set SS_WINO43_MIN_WGS=1 to pin the kernel choice per layer) -- hence so do the canvas and the frames."""

    def __init__(self, nets, device='cuda', warp_mode='NORMAL', fusion_mode='AVERAGE', chunk=None, numa_bind=False):
        self.nets, self.dev = nets, torch.device(device)
        self.warp_mode, self.fusion_mode = warp_mode, fusion_mode
        self.chunk = chunk or SPATIAL_CHUNK
        self.io = HostClipRunner(nets, device, warp_mode, fusion_mode, numa_bind=numa_bind)
        self.acc = self.meshes = self.bbox = self.hc = self.wc = None
        self.prescaled = False

    def _chunks(self, views):
        n = len(views[0])
        if len(views) not in (2, 3):
            raise ValueError('two or three views, got %d' % len(views))
        for v in views[1:]:
            if len(v) != n:
                raise ValueError('views differ in length: %d vs %d frames' % (n, len(v)))
        for s in range(0, n, self.chunk):
            e = min(s + self.chunk, n)
            yield s, e, tuple(v[s:e] for v in views)

    def _uploads(self, views):
        """(s, e, device uint8 tensors, ready event), the next chunk's upload always enqueued before this one is consumed."""
        it = self._chunks(views)
        nxt = next(it, None)
        up = None if nxt is None else self.io._upload(nxt[2])
        while nxt is not None:
            cur, cur_up = nxt, up
            nxt = next(it, None)
            up = None if nxt is None else self.io._upload(nxt[2])
            yield cur[0], cur[1], cur_up[0], cur_up[1]

    @torch.no_grad()
    def estimate(self, *views):
        """Pass 1 over V = 2 or 3 views -> the dict of `estimate_meshes` for the WHOLE video (three views: of pair (1,2), plus
        'pair23'); sets self.meshes (the V render meshes [1,N,7,9,2]) and the global canvas self.bbox, self.hc, self.wc."""
        n = len(views[0])
        if n < WINDOW:
            raise ValueError('need at least %d frames for the sliding smooth window, got %d' % (WINDOW, n))
        spatial_net, temporal_net, smooth_net = self.nets
        comp = self.io.comp
        est = est23 = None
        mid_cache = [] if len(views) == 3 else None
        img_h = img_w = None
        for s, e, d, ev in self._uploads(views):
            comp.wait_event(ev)
            with torch.cuda.stream(comp):
                img_h, img_w = d[0].shape[1], d[0].shape[2]
                if est is None:
                    est = JointEstimator(spatial_net, temporal_net, n, self.dev, cache2=mid_cache)
                    if mid_cache is not None:       # the middle view's temporal motions: pair (1,2)'s buffer, filled chunk by chunk
                        est23 = JointEstimator(spatial_net, temporal_net, n, self.dev, tmotion1=est.tm[1])
                lrs = [ops.ingest_u8(t, want_hr=False)[1] for t in d]
                est.push(lrs[0], lrs[1])
                if est23 is not None:
                    est23.push(None, lrs[2], mid_cache.pop())
                used = torch.cuda.Event()
                used.record(comp)
            self.io.consumed(d, used)
        with torch.cuda.stream(comp):
            self.acc = smooth_stage(smooth_net, *est.result())
            if est23 is None:
                self.meshes, self.prescaled = [self.acc['smooth_mesh1'], self.acc['smooth_mesh2']], False
            else:
                a23 = smooth_stage(smooth_net, *est23.result())
                self.acc['pair23'] = a23
                self.meshes = list(three_view_compose(self.acc['smooth_mesh1'], self.acc['smooth_mesh2'], a23['smooth_mesh1'],
                                                      a23['smooth_mesh2'], img_h, img_w))
                self.prescaled = True
            self.bbox = canvas_bbox(self.meshes, img_h, img_w, self.prescaled)
            self.hc, self.wc = canvas_size(self.bbox)       # (host read-back: the compute stream is drained here)
        return self.acc

    @torch.no_grad()
    def render(self, *views):
        """Pass 2: yields (uint8 [m,Hc,Wc,3] pinned host tensor, s, e) per chunk of frames [s, e), one chunk late at most;
        a yielded tensor stays valid until two more chunks have been yielded (HostClipRunner keeps depth + 2 slots)."""
        if self.acc is None:
            raise RuntimeError('estimate() first')
        if len(views) != len(self.meshes):
            raise ValueError('estimate() saw %d views, render() got %d' % (len(self.meshes), len(views)))
        io = self.io
        pending = None
        k = 0
        for s, e, d, ev in self._uploads(views):
            io.comp.wait_event(ev)
            with torch.cuda.stream(io.comp):
                ms = [m[:, s:e].contiguous() for m in self.meshes]
                if _u8_fused(self.fusion_mode):
                    u8, _, _ = render_frames_u8(list(d), ms, self.warp_mode, out=io.out_buffer, bbox=self.bbox,
                                                size=(self.hc, self.wc), prescaled=self.prescaled, fusion_mode=self.fusion_mode)
                else:
                    hrs = [ops.ingest_u8(t)[0] for t in d]
                    fr, _, _ = render_frames(hrs, ms, self.warp_mode, self.fusion_mode, prescaled=self.prescaled,
                                             bbox=self.bbox, size=(self.hc, self.wc))
                    u8 = ops.canvas_to_u8(fr, out=io.out_buffer((e - s, self.hc, self.wc, 3)))
                ev2 = torch.cuda.Event()
                ev2.record(io.comp)
            io.consumed(d, ev2)
            host, done = io._download(k, u8, ev2)
            if pending is not None:
                pending[0].synchronize()
                yield pending[1], pending[2], pending[3]
            pending = (done, host, s, e)
            k += 1
        if pending is not None:
            pending[0].synchronize()
            yield pending[1], pending[2], pending[3]


def _run_long(views, nets, warp_mode, fusion_mode, device, chunk, sink):
    st = LongVideoStitcher(nets, device, warp_mode, fusion_mode, chunk)
    st.estimate(*views)
    video = None
    if sink is None:
        import numpy as np
        video = np.empty((len(views[0]), st.hc, st.wc, 3), dtype=np.uint8)
    for host, s, e in st.render(*views):
        if sink is None:
            video[s:e] = host.numpy()
        else:
            sink(host, s, e)
    return (video, st.hc, st.wc) + tuple(st.meshes)


@torch.no_grad()
def run_two_view_long(frames1, frames2, nets, warp_mode='NORMAL', fusion_mode='AVERAGE', device='cuda', chunk=None,
                      sink=None):
    """`run_two_view_u8` for videos that do not fit the device: frames1, frames2 = uint8 [N,H,W,3] host arrays (anything
    sliceable along frames: ndarray, pinned tensor, np.memmap), rendered onto the reference's single global canvas
    (LongVideoStitcher).  sink(video_chunk [m,Hc,Wc,3] uint8 host tensor, s, e) receives the stitched frames in order
    (e.g. cv2.VideoWriter.write per frame, test_online_tra.py:409-417); without a sink they are collected into one
    ndarray.  -> (video | None, Hc, Wc, smooth_mesh1, smooth_mesh2)."""
    return _run_long((frames1, frames2), nets, warp_mode, fusion_mode, device, chunk, sink)


@torch.no_grad()
def run_three_view_long(frames1, frames2, frames3, nets, warp_mode='NORMAL', fusion_mode='AVERAGE', device='cuda', chunk=None,
                        sink=None):
    """`run_three_view` (test_online_tra_threeview.py:154-505) for host-resident uint8 videos of any length, on the
    reference's single global canvas.  -> (video | None, Hc, Wc, mesh1, middle, mesh3), the meshes in canvas pixels."""
    return _run_long((frames1, frames2, frames3), nets, warp_mode, fusion_mode, device, chunk, sink)
