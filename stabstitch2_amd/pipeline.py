"""Online 2-view / 3-view stitching pipeline on the MI355X HIP engine.

Tensor-level counterpart of `test()` in Full_model_inference/Codes/test_online_tra.py:158-426 and
test_online_tra_threeview.py:95-519 (file / video I/O excluded), with the reference's helper names
(`get_stable_sqe`, `linear_blender`, `recover_mesh`, `get_rigid_mesh`, `get_norm_mesh`).

The reference walks the clip frame by frame with batch 1.  Here every stage is one batched pass
over the whole clip (the clip is resident in HBM): SpatialNet over all frame pairs, TemporalNet
over all frames of a view, one batched tsmotion composition per view, all sliding SmoothNet windows
as one batch, one batched TPS solve for every (frame, view), then one fused warp+blend launch per
stitched frame.  The only host round trip is the data-dependent canvas size (test_online_tra.py:122-123).

Everything tensor-sized (frames, feature maps, cost volumes, canvases) is computed by the HIP kernels.  What stays as
torch expressions here is mesh-sized bookkeeping on [N,7,9,2] tensors (a few KB): assembling the clip's meshes from the
sliding windows and the metric harness's stitched paths (`_stitch_windows`), and the three-view mesh alignment
(`three_view_compose`: scale, mean offset, middle mesh, bbox, normalise) -- torch device ops, no host sync.
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


@torch.no_grad()
def spatial_stage(spatial_net, lr1, lr2, chunk=None, cache1=None):
    """lr1, lr2 [N,3,360,480] device -> smotion1, smotion2 [N,7,9,2].
    cache1: per chunk the (f64, f32) trunk features of view 1 kept by an earlier pass (then only view 2 goes through
    the trunk)."""
    chunk = chunk or SPATIAL_CHUNK
    m1, m2 = [], []
    for i, s in enumerate(range(0, lr1.shape[0], chunk)):
        if cache1 is None:
            o = build_SpatialNet(spatial_net, lr1[s:s + chunk], lr2[s:s + chunk])
            a1, a2 = o['motion1'], o['motion2']
        else:
            f64_2, f32_2 = spatial_net.trunk_features([lr2[s:s + chunk]])
            off = spatial_net.forward_pair(cache1[i][0], f64_2, cache1[i][1], f32_2, LR_H, LR_W)
            a1, a2 = ops.spatial_meshes(off[0], off[1], off[2], LR_H, LR_W)
        m1.append(a1)
        m2.append(a2)
    return torch.cat(m1, 0), torch.cat(m2, 0)


@torch.no_grad()
def temporal_stage(temporal_net, lr):
    """lr [N,3,360,480] device -> tmotion [N,7,9,2] (frame 0 = 0)."""
    m = temporal_net.motions(lr.unsqueeze(1))[:, 0]
    return torch.cat((torch.zeros_like(m[:1]), m), 0)


@torch.no_grad()
def temporal_stage_views(temporal_net, lrs):
    """Both (all) views of a clip in ONE batched pass: lrs = list of V tensors [N,3,360,480]
    -> list of V tmotion tensors [N,7,9,2] (frame 0 = 0)."""
    ms = temporal_net.motions_views(lrs)                          # V x [N-1,7,9,2]
    z = torch.zeros_like(ms[0][:1])
    return [torch.cat((z, m), 0) for m in ms]


SHARED_STEM = os.environ.get('SS_SHARED_STEM', '1') == '1'


@torch.no_grad()
def joint_stage(spatial_net, temporal_net, lr1, lr2, chunk=None, tmotion1=None, cache2=None):
    """SpatialNet and TemporalNet of a 2-view clip in one sweep: both nets start with the same 7x7/2 conv + pool on the
    same LR frames (spatial_network.py:127-130 and temporal_network.py:47-50 build identical stems), so the stem runs
    ONCE with 2 x 64 filters and each net continues from its half of the channels.
    -> (smotion1, smotion2, tmotion1, tmotion2), each [N,7,9,2] (tmotion frame 0 = 0).
    tmotion1: view 1's temporal motions when a previous pass already produced them (three-view: the middle view is
    view 2 of pair (1,2) and view 1 of pair (2,3)); its TemporalNet trunk is then skipped."""
    from . import layers as L
    chunk = chunk or SPATIAL_CHUNK
    sp, tp = spatial_net._prepared(), temporal_net._prepared()
    # shared-stem filters derive from BOTH nets' weights: keyed on both versions, one live entry (replaced on a miss)
    key = 'stem_pair'
    ver = (spatial_net.weights_version, temporal_net.weights_version)
    if sp.get('stem_pair_version') != ver:
        sp[key] = L.pair_stems(sp['s1'], tp['s1'])
        sp['stem_pair_version'] = ver
    n = lr1.shape[0]
    m1, m2 = [], []
    ft = None
    for s in range(0, n, chunk):
        e = min(s + chunk, n)
        b = e - s
        xa, xb = L.run_stem_shared([lr1[s:e], lr2[s:e]], sp[key])
        f64 = L.run_trunk_body(xa, sp['s1'])
        f32 = L.run_stage2(f64, sp['s2'])
        off1, off_ref, off_tgt = spatial_net.forward_pair(f64[:b], f64[b:], f32[:b], f32[b:], LR_H, LR_W)
        if cache2 is not None:                    # view 2's trunk features, for a later pair that starts with this view
            cache2.append((f64[b:], f32[b:]))
        a1, a2 = ops.spatial_meshes(off1, off_ref, off_tgt, LR_H, LR_W)
        m1.append(a1)
        m2.append(a2)
        views = (1,) if tmotion1 is not None else (0, 1)
        f = L.run_trunk_body(xb if tmotion1 is None else xb[b:], tp['s1'])  # [len(views)*b,45,60,128], view-major
        if n <= chunk:
            ft = [f[i * b:(i + 1) * b] for i in range(len(views))]
        else:
            if ft is None:
                ft = [torch.empty((n,) + tuple(f.shape[1:]), device=f.device, dtype=torch.float32) for _ in views]
            for i in range(len(views)):
                ft[i][s:e].copy_(f[i * b:(i + 1) * b])
    ms = temporal_net.motions_from_view_features(ft)
    z = torch.zeros_like(ms[0][:1])
    tm = [torch.cat((z, m), 0) for m in ms]
    if tmotion1 is not None:
        tm = [tmotion1, tm[0]]
    return torch.cat(m1, 0), torch.cat(m2, 0), tm[0], tm[1]


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
    elif SHARED_STEM and tmotion1 is None and spatial_cache1 is None:
        # (with view 1's motions / features known, half of a shared stem would be wasted)
        s1, s2, t1, t2 = joint_stage(spatial_net, temporal_net, lr1, lr2, cache2=cache2)
    else:
        s1, s2 = spatial_stage(spatial_net, lr1, lr2, cache1=spatial_cache1)
        if tmotion1 is None:
            t1, t2 = temporal_stage_views(temporal_net, [lr1, lr2])
        else:
            t1, t2 = tmotion1, temporal_stage_views(temporal_net, [lr2])[0]
    smesh1, tsm1 = ops.tsmotion(s1, t1, LR_H, LR_W)
    smesh2, tsm2 = ops.tsmotion(s2, t2, LR_H, LR_W)
    nw = n - (WINDOW - 1)
    o, _ = smooth_net.run_windows(smesh1, smesh2, tsm1, tsm2, nw, WINDOW, 1, 1)
    out = _stitch_windows(o)
    out['smotion1'], out['smotion2'], out['tmotion1'], out['tmotion2'] = s1, s2, t1, t2
    out['tsmotion1'], out['tsmotion2'] = tsm1, tsm2
    if cache2:
        out['spatial_cache2'] = cache2
    return out


def _stitch_windows(o):
    """Per-window SmoothNet outputs [nw,7,7,9,2] -> the clip's tensors [1,N,7,9,2] (mesh-sized torch glue).
    Window 0 contributes its 7 frames, every later window its last frame (test_online_tra.py:377-392); the metric
    harness's paths are chained across windows as test_metric_ssd.py:427-436 does."""
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
    key = (int(img_h), int(img_w), str(device))
    t = _nrigid_cache.get(key)
    if t is None:
        t = get_norm_mesh(get_rigid_mesh(1, img_h, img_w, device=device), img_h, img_w).contiguous()
        _nrigid_cache[key] = t
    return t


# AVERAGE render: skip a view's 63-term spline on canvas tiles it provably cannot reach (ops.render_footprints) and take
# its contribution there as exactly 0 instead of the rounding residue of the reference's clamped sampler.
# SS_SKIP_OUTSIDE=0: evaluate every view at every pixel.
SKIP_OUTSIDE = os.environ.get('SS_SKIP_OUTSIDE', '1') == '1'


# ------------------------------------------------------------------ render
@torch.no_grad()
def render_plan(meshes, img_h, img_w, prescaled=False):
    """Canvas + TPS coefficients for every (frame, view).
    meshes: list of V tensors [1,N,7,9,2] (LR scale, or HR canvas pixels with prescaled=True).
    -> (Hc, Wc, source [N,V,63,2], T [N,V,2,66])."""
    dev = meshes[0].device
    v = len(meshes)
    n = meshes[0].shape[1]
    sh, sw = (0.0, 0.0) if prescaled else (img_h, img_w)
    bbox = ops.mesh_bbox([m for m in meshes], sh, sw)
    bb = bbox.cpu()                                       # the one host sync: data-dependent canvas size
    wc_f, hc_f = bb[1] - bb[0], bb[3] - bb[2]
    hc, wc = int(hc_f.int()), int(wc_f.int())
    src = torch.stack([ops.mesh_normalize(m[0], bbox, sh, sw) for m in meshes], 1).contiguous()   # [N,V,63,2]
    tgt = norm_rigid_mesh(img_h, img_w, dev).expand(n * v, -1, -1).contiguous()
    T = ops.tps_solve(src.view(n * v, 63, 2), tgt).view(n, v, 2, 66)
    return hc, wc, src, T


@torch.no_grad()
def render_frames(img_lists, meshes, warp_mode='NORMAL', fusion_mode='AVERAGE', out=None, prescaled=False):
    """img_lists: V lists (or tensors [N,3,H,W]) of HR frames (0..255); meshes: V tensors [1,N,7,9,2].
    -> (frames [N,3,Hc,Wc] device tensor, Hc, Wc)."""
    v = len(img_lists)
    dev = meshes[0].device
    n = meshes[0].shape[1]
    first = img_lists[0][0]
    img_h, img_w = first.shape[-2:]
    hc, wc, src, T = render_plan(meshes, img_h, img_w, prescaled)
    if out is None:
        out = torch.empty((n, 3, hc, wc), device=dev, dtype=torch.float32)
    fp = ops.render_footprints(src, T, img_h, img_w, hc, wc) if (SKIP_OUTSIDE and fusion_mode == 'AVERAGE') else None
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
def run_two_view(hr1, hr2, lr1, lr2, nets, warp_mode='NORMAL', fusion_mode='AVERAGE', to_host=False):
    """-> (frames, Hc, Wc, smooth_mesh1, smooth_mesh2); frames = device tensor [N,3,Hc,Wc]
    (or list of HWC ndarrays with to_host=True)."""
    acc = estimate_meshes(nets, lr1, lr2)
    frames, hc, wc = render_frames([hr1, hr2], [acc['smooth_mesh1'], acc['smooth_mesh2']], warp_mode, fusion_mode)
    if to_host:
        host = frames.permute(0, 2, 3, 1).cpu().numpy()
        frames = [host[i] for i in range(host.shape[0])]
    return frames, hc, wc, acc['smooth_mesh1'], acc['smooth_mesh2']


# ------------------------------------------------------------------ three-view (threeview:345-505)
def _scale(m, img_h, img_w):
    return torch.stack([m[..., 0] * img_w / 480, m[..., 1] * img_h / 360], 4)


@torch.no_grad()
def three_view_compose(w12_m1, w12_m2, w23_m1, w23_m2, img_h, img_w):
    """Mesh alignment, middle plane and TPS re-projection of the outer views.  Inputs [1,N,7,9,2] (LR scale)
    -> (mesh1, middle, mesh3) in first-canvas HR pixels.  Mesh-sized glue (scale, mean offset, middle mesh, bbox,
    normalise: [1,N,7,9,2] tensors) stays in torch on the device; the TPS solves and point evaluations run on the HIP
    kernels."""
    a1, a2 = _scale(w12_m1, img_h, img_w), _scale(w12_m2, img_h, img_w)
    b1, b2 = _scale(w23_m1, img_h, img_w), _scale(w23_m2, img_h, img_w)
    off = (a2 - b1).reshape(a2.shape[0], a2.shape[1], -1, 2).mean(2).unsqueeze(2).unsqueeze(2)
    b1, b2 = b1 + off, b2 + off
    mid = (a2 + b1) / 2.
    wmin = torch.stack([m[..., 0].min() for m in (a1, a2, b1, b2)]).min()
    wmax = torch.stack([m[..., 0].max() for m in (a1, a2, b1, b2)]).max()
    hmin = torch.stack([m[..., 1].min() for m in (a1, a2, b1, b2)]).min()
    hmax = torch.stack([m[..., 1].max() for m in (a1, a2, b1, b2)]).max()
    ow, oh = wmax - wmin, hmax - hmin

    def shift(m):
        return torch.stack([m[..., 0] - wmin, m[..., 1] - hmin], 4)

    def nrm(m):      # [1,N,7,9,2] -> [N,63,2]
        return torch.stack([m[0, ..., 0] * 2. / ow - 1., m[0, ..., 1] * 2. / oh - 1.], 3).reshape(m.shape[1], -1, 2)

    def rec(nm):
        return torch.stack([(nm[..., 0] + 1) * ow / 2., (nm[..., 1] + 1) * oh / 2.], 2).reshape(1, -1, 7, 9, 2)
    a1, a2, b1, b2, mid = map(shift, (a1, a2, b1, b2, mid))
    nmid = nrm(mid).contiguous()
    n1 = ops.tps_points(nrm(a1).contiguous(), nrm(a2).contiguous(), ops.tps_solve(nrm(a2).contiguous(), nmid))
    n3 = ops.tps_points(nrm(b2).contiguous(), nrm(b1).contiguous(), ops.tps_solve(nrm(b1).contiguous(), nmid))
    return rec(n1), mid, rec(n3)


@torch.no_grad()
def three_view_render(img1, img2, img3, mesh1, middle, mesh3, warp_mode='NORMAL', fusion_mode='AVERAGE'):
    """Meshes are HR-scale canvas pixels here (output of three_view_compose)."""
    return render_frames([img1, img2, img3], [mesh1, middle, mesh3], warp_mode, fusion_mode, prescaled=True)


@torch.no_grad()
def run_three_view(hr1, hr2, hr3, lr1, lr2, lr3, nets, warp_mode='NORMAL', fusion_mode='AVERAGE'):
    # the middle view's TemporalNet motions and SpatialNet trunk features are computed once and reused by pair (2,3)
    a12 = estimate_meshes(nets, lr1, lr2, keep_spatial_cache2=True)
    a23 = estimate_meshes(nets, lr2, lr3, tmotion1=a12['tmotion2'], spatial_cache1=a12.get('spatial_cache2'))
    img_h, img_w = hr1[0].shape[-2:]
    m1, mid, m3 = three_view_compose(a12['smooth_mesh1'], a12['smooth_mesh2'], a23['smooth_mesh1'],
                                     a23['smooth_mesh2'], img_h, img_w)
    frames, hc, wc = three_view_render(hr1, hr2, hr3, m1, mid, m3, warp_mode, fusion_mode)
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


@torch.no_grad()
def render_frames_u8(frame_lists, meshes, warp_mode='NORMAL', out=None):
    """frame_lists: V device tensors [N,H,W,3] uint8; meshes: V tensors [1,N,7,9,2] -> (uint8 [N,Hc,Wc,3], Hc, Wc):
    `render_frames(..., 'AVERAGE')` followed by `to_video_frames`, fused."""
    v = len(frame_lists)
    n = meshes[0].shape[1]
    img_h, img_w = frame_lists[0].shape[1], frame_lists[0].shape[2]
    hc, wc, src, T = render_plan(meshes, img_h, img_w)
    if out is None:
        out = torch.empty((n, hc, wc, 3), device=meshes[0].device, dtype=torch.uint8)
    fp = ops.render_footprints(src, T, img_h, img_w, hc, wc) if SKIP_OUTSIDE else None
    for i in range(n):
        ops.render_average_u8([frame_lists[k][i] for k in range(v)], src[i], T[i], hc, wc, warp_mode, out=out[i],
                              footprint=None if fp is None else fp[i])
    return out, hc, wc


def _as_device_u8(frames, device):
    if not torch.is_tensor(frames):
        frames = torch.from_numpy(frames)
    return frames.to(device, non_blocking=True).contiguous()


@torch.no_grad()
def run_two_view_u8(frames1, frames2, nets, warp_mode='NORMAL', fusion_mode='AVERAGE', device='cuda', to_host=False):
    """uint8 in, uint8 out: ingest -> estimate -> render -> video frames.  -> (uint8 [N,Hc,Wc,3], Hc, Wc, m1, m2)."""
    if U8_FUSED and fusion_mode == 'AVERAGE':
        f1, f2 = _as_device_u8(frames1, device), _as_device_u8(frames2, device)
        _, lr1 = ops.ingest_u8(f1, want_hr=False)
        _, lr2 = ops.ingest_u8(f2, want_hr=False)
        acc = estimate_meshes(nets, lr1, lr2)
        m1, m2 = acc['smooth_mesh1'], acc['smooth_mesh2']
        u8, hc, wc = render_frames_u8([f1, f2], [m1, m2], warp_mode)
        return (u8.cpu().numpy() if to_host else u8), hc, wc, m1, m2
    hr1, lr1 = load_frames_u8(frames1, device=device)
    hr2, lr2 = load_frames_u8(frames2, device=device)
    frames, hc, wc, m1, m2 = run_two_view(hr1, hr2, lr1, lr2, nets, warp_mode, fusion_mode)
    return to_video_frames(frames, to_host), hc, wc, m1, m2


class HostClipRunner:
    """uint8 clips in pinned host memory -> stitched uint8 clips in pinned host memory, with the PCIe copies of
    neighbouring clips hidden behind the compute of the current one (three HIP streams: upload, compute, download).

        for video, hc, wc in HostClipRunner(nets).run(clips):     # clips yields (frames1, frames2) uint8 [N,H,W,3]
            writer.write(video[k]) ...

    The upload of clip k+1 is enqueued before the host starts issuing clip k's kernels (whose canvas-size read-back is
    the path's one host sync), the download of clip k runs while clip k+1 computes.  Yields (ndarray-backed pinned
    uint8 tensor [N,Hc,Wc,3], Hc, Wc) one clip late at most; a yielded tensor stays valid until `depth` more clips
    have been yielded."""

    def __init__(self, nets, device='cuda', warp_mode='NORMAL', fusion_mode='AVERAGE', depth=2):
        self.nets, self.dev = nets, torch.device(device)
        self.warp_mode, self.fusion_mode, self.depth = warp_mode, fusion_mode, depth
        self.up, self.comp, self.down = (torch.cuda.Stream(self.dev) for _ in range(3))
        self._host = [dict() for _ in range(depth + 1)]

    def _upload(self, clip):
        with torch.cuda.stream(self.up):
            d = [(torch.from_numpy(f) if not torch.is_tensor(f) else f).to(self.dev, non_blocking=True) for f in clip]
            ev = torch.cuda.Event()
            ev.record(self.up)
        return d, ev

    def _compute(self, d, ev):
        self.comp.wait_event(ev)
        with torch.cuda.stream(self.comp):
            for t in d:
                t.record_stream(self.comp)
            d = [t if t.is_contiguous() else t.contiguous() for t in d]
            if U8_FUSED and self.fusion_mode == 'AVERAGE':
                _, lr1 = ops.ingest_u8(d[0], want_hr=False)
                _, lr2 = ops.ingest_u8(d[1], want_hr=False)
                acc = estimate_meshes(self.nets, lr1, lr2)
                u8, hc, wc = render_frames_u8(d, [acc['smooth_mesh1'], acc['smooth_mesh2']], self.warp_mode)
            else:
                hr1, lr1 = ops.ingest_u8(d[0])
                hr2, lr2 = ops.ingest_u8(d[1])
                frames, hc, wc, _, _ = run_two_view(hr1, hr2, lr1, lr2, self.nets, self.warp_mode, self.fusion_mode)
                u8 = ops.canvas_to_u8(frames)
            ev2 = torch.cuda.Event()
            ev2.record(self.comp)
        return u8, hc, wc, ev2

    def _download(self, k, u8, ev):
        key = tuple(u8.shape)
        if key not in self._host[0]:         # new canvas size: pin every slot now, not one clip at a time
            for s_ in self._host:
                s_.clear()
                s_[key] = torch.empty(key, dtype=torch.uint8).pin_memory()
        slot = self._host[k % (self.depth + 1)]
        self.down.wait_event(ev)
        with torch.cuda.stream(self.down):
            u8.record_stream(self.down)
            slot[key].copy_(u8, non_blocking=True)
            done = torch.cuda.Event()
            done.record(self.down)
        return slot[key], done

    @torch.no_grad()
    def run(self, clips):
        it = iter(clips)
        try:
            nxt = self._upload(next(it))
        except StopIteration:
            return
        pending = None
        k = 0
        while nxt is not None:
            cur = nxt
            try:
                nxt = self._upload(next(it))
            except StopIteration:
                nxt = None
            u8, hc, wc, ev = self._compute(*cur)
            host, done = self._download(k, u8, ev)
            if pending is not None:
                pending[3].synchronize()
                yield pending[0], pending[1], pending[2]
            pending = (host, hc, wc, done)
            k += 1
        pending[3].synchronize()
        yield pending[0], pending[1], pending[2]
