"""Online pipeline, canvas, blending and three-view composition (oracle, CPU).

Reference sites (relative to /root/reference/Full_model_inference/Codes):
  linear_blender            test_online_tra.py:34-58   (GaussianBlur = torchvision 0.14.1, restated)
  get_stable_sqe            test_online_tra.py:96-154
  spatial / temporal loops  test_online_tra.py:284-299
  tsmotion preparation      test_online_tra.py:309-347
  sliding smooth window     test_online_tra.py:359-392
  three-view composition    test_online_tra_threeview.py:345-505
"""
import torch
import torch.nn.functional as F

from . import geometry as G
from . import samplers as S
from . import nets as N

LR_H, LR_W = 360, 480
WINDOW = 7


# --------------------------------------------------------------------------- blending
def gaussian_blur_21_20(x):
    """torchvision.transforms.GaussianBlur((21,21), sigma=20) restated:
    1-D kernel exp(-0.5 (t/sigma)^2) on linspace(-10, 10, 21), normalised; reflect pad 10;
    depthwise separable (outer product) convolution."""
    k, sigma = 21, 20.0
    t = torch.linspace(-(k - 1) * 0.5, (k - 1) * 0.5, k)
    pdf = torch.exp(-0.5 * (t / sigma) ** 2)
    k1 = pdf / pdf.sum()
    k2 = torch.mm(k1[:, None], k1[None, :])
    c = x.shape[1]
    wgt = k2.expand(c, 1, k, k)
    xp = F.pad(x, [k // 2] * 4, mode='reflect')
    return F.conv2d(xp, wgt, groups=c)


def average_fusion(a, b):
    """a, b [C,H,W] on the 0..255 scale."""
    return a * (a / (a + b + 1e-6)) + b * (b / (a + b + 1e-6))


def linear_blender(ref, tgt, ref_m, tgt_m, mask=False):
    """ref,tgt [1,3,H,W]; ref_m,tgt_m [1,1,H,W] warped ones-masks."""
    r1, c1 = torch.nonzero(ref_m[0, 0], as_tuple=True)
    r2, c2 = torch.nonzero(tgt_m[0, 0], as_tuple=True)
    cen1 = (r1.float().mean(), c1.float().mean())
    cen2 = (r2.float().mean(), c2.float().mean())
    vec = (cen2[0] - cen1[0], cen2[1] - cen1[1])

    ovl = (ref_m * tgt_m).round()[:, 0].unsqueeze(1)
    ref_only = ref_m[:, 0].unsqueeze(1) - ovl
    r, c = torch.nonzero(ovl[0, 0], as_tuple=True)
    ovl_mask = torch.zeros_like(ref_only)
    proj = (r - cen1[0]) * vec[0] + (c - cen1[1]) * vec[1]
    ovl_mask[ovl.bool()] = (proj - proj.min()) / (proj.max() - proj.min() + 1e-3)

    mask1 = (gaussian_blur_21_20(ref_only + (1 - ovl_mask) * ref_m[:, 0].unsqueeze(1)) * ref_m
             + ref_only).clamp(0, 1)
    if mask:
        return mask1
    mask2 = (1 - mask1) * tgt_m
    return ref * mask1 + tgt * mask2


# --------------------------------------------------------------------------- canvas + render
def _bbox(meshes):
    wmax = torch.stack([m[..., 0].max() for m in meshes]).max()
    wmin = torch.stack([m[..., 0].min() for m in meshes]).min()
    hmax = torch.stack([m[..., 1].max() for m in meshes]).max()
    hmin = torch.stack([m[..., 1].min() for m in meshes]).min()
    return wmin, wmax, hmin, hmax


def _scale_to_hr(mesh, img_h, img_w):
    return torch.stack((mesh[..., 0] * img_w / LR_W, mesh[..., 1] * img_h / LR_H), dim=4)


def get_stable_sqe(img1_list, img2_list, smooth_mesh1, smooth_mesh2, warp_mode, fusion_mode):
    """HR frames lists of [1,3,H,W] (0..255), meshes [1,N,7,9,2] at LR scale.
    -> (list of ndarray [Hc,Wc,3] fp32, Wc int tensor, Hc int tensor)."""
    b, _, img_h, img_w = img2_list[0].shape
    rigid = G.rigid_mesh(b, img_h, img_w)
    nrigid = G.norm_mesh(rigid, img_h, img_w)
    m1 = _scale_to_hr(smooth_mesh1, img_h, img_w)
    m2 = _scale_to_hr(smooth_mesh2, img_h, img_w)
    wmin, wmax, hmin, hmax = _bbox([m1, m2])
    out_w = wmax - wmin
    out_h = hmax - hmin
    size = (int(out_h.int()), int(out_w.int()))

    frames = []
    for i in range(len(img2_list)):
        a = m1[:, i]
        nm1 = G.norm_mesh(torch.stack((a[..., 0] - wmin, a[..., 1] - hmin), dim=3), out_h, out_w)
        c = m2[:, i]
        nm2 = G.norm_mesh(torch.stack((c[..., 0] - wmin, c[..., 1] - hmin), dim=3), out_h, out_w)
        img1, img2 = img1_list[i], img2_list[i]
        src = torch.cat((nm1, nm2), dim=0)
        tgt = torch.cat((nrigid, nrigid), dim=0)
        if fusion_mode == 'AVERAGE':
            w = S.tps_warp(torch.cat((img1, img2), dim=0), src, tgt, size, warp_mode)
            fused = average_fusion(w[0], w[1])
        else:
            one = torch.ones_like(img1[:, 0:1])
            w = S.tps_warp(torch.cat((torch.cat((img1, one), 1), torch.cat((img2, one), 1)), 0),
                           src, tgt, size, warp_mode)
            fused = linear_blender(w[0:1, 0:3], w[1:2, 0:3], w[0:1, 3:4], w[1:2, 3:4])[0]
        frames.append(fused.numpy().transpose(1, 2, 0))
    return frames, out_w.int(), out_h.int()


# --------------------------------------------------------------------------- motion stages
def spatial_stage(spatial_net, lr1_list, lr2_list):
    s1, s2 = [], []
    with torch.no_grad():
        for a, c in zip(lr1_list, lr2_list):
            out = N.build_SpatialNet(spatial_net, a, c)
            s1.append(out['motion1'])
            s2.append(out['motion2'])
    return s1, s2


def temporal_stage(temporal_net, lr_list):
    with torch.no_grad():
        return N.build_TemporalNet(temporal_net, lr_list)['motion_list']


def tsmotion_prepare(smotion_list, tmotion_list):
    """-> (smesh_list, tsmotion_list), each N x [1,7,9,2] at LR scale."""
    rigid = G.rigid_mesh(1, LR_H, LR_W)
    nrigid = G.norm_mesh(rigid, LR_H, LR_W)
    smesh, tsm = [], []
    for k in range(len(tmotion_list)):
        sm = rigid + smotion_list[k]
        if k == 0:
            ts = smotion_list[k] * 0
        else:
            prev = G.norm_mesh(rigid + smotion_list[k - 1], LR_H, LR_W)
            tm = G.norm_mesh(rigid + tmotion_list[k], LR_H, LR_W)
            ts = G.recover_mesh(S.tps_points(tm, nrigid, prev), LR_H, LR_W) - sm
        smesh.append(sm)
        tsm.append(ts)
    return smesh, tsm


def smooth_stage(smooth_net, tsm1, tsm2, smesh1, smesh2):
    """Sliding 7-frame window; window 0 contributes 7 meshes, later windows their last one."""
    keys = ('ori_mesh1', 'smooth_mesh1', 'ori_mesh2', 'smooth_mesh2', 'ori_path2', 'smooth_path2')
    acc = {}
    n = len(tsm1)
    for k in range(n - (WINDOW - 1)):
        t1 = list(tsm1[k:k + WINDOW])
        t2 = list(tsm2[k:k + WINDOW])
        t1[0] = t1[0] * 0
        t2[0] = t2[0] * 0
        with torch.no_grad():
            o = N.build_SmoothNet(smooth_net, t1, t2, smesh1[k:k + WINDOW], smesh2[k:k + WINDOW])
        if k == 0:
            for key in keys:
                acc[key] = o[key]
        else:
            for key in ('ori_mesh1', 'smooth_mesh1', 'ori_mesh2', 'smooth_mesh2'):
                acc[key] = torch.cat((acc[key], o[key][:, -1:]), dim=1)
            # path stitching of test_metric_ssd.py:433-436
            new_ori = acc['ori_path2'][:, -1] + (o['ori_path2'][:, -1] - o['ori_path2'][:, -2])
            acc['ori_path2'] = torch.cat((acc['ori_path2'], new_ori.unsqueeze(1)), dim=1)
            new_sm = acc['ori_path2'][:, -1] + (o['smooth_path2'][:, -1] - o['ori_path2'][:, -1])
            acc['smooth_path2'] = torch.cat((acc['smooth_path2'], new_sm.unsqueeze(1)), dim=1)
    return acc


def estimate_meshes(nets, lr1_list, lr2_list):
    """Stages 1-3 for one 2-view clip -> dict from smooth_stage."""
    spatial_net, temporal_net, smooth_net = nets
    s1, s2 = spatial_stage(spatial_net, lr1_list, lr2_list)
    t1 = temporal_stage(temporal_net, lr1_list)
    t2 = temporal_stage(temporal_net, lr2_list)
    smesh1, tsm1 = tsmotion_prepare(s1, t1)
    smesh2, tsm2 = tsmotion_prepare(s2, t2)
    return smooth_stage(smooth_net, tsm1, tsm2, smesh1, smesh2)


def run_two_view(hr1_list, hr2_list, lr1_list, lr2_list, nets, warp_mode='NORMAL',
                 fusion_mode='AVERAGE'):
    """Counterpart of test() in test_online_tra.py:158-426 at the tensor level.
    -> (frames, Hc, Wc, smooth_mesh1, smooth_mesh2)."""
    acc = estimate_meshes(nets, lr1_list, lr2_list)
    frames, wc, hc = get_stable_sqe(hr1_list, hr2_list, acc['smooth_mesh1'], acc['smooth_mesh2'],
                                    warp_mode, fusion_mode)
    return frames, int(hc), int(wc), acc['smooth_mesh1'], acc['smooth_mesh2']


# --------------------------------------------------------------------------- three-view
def three_view_compose(w12_m1, w12_m2, w23_m1, w23_m2, img_h, img_w):
    """Mesh alignment + middle plane + re-projection (threeview:345-427).
    Inputs [1,N,7,9,2] at LR scale. -> (mesh1, middle, mesh3) in first-canvas pixels."""
    w12_m1 = _scale_to_hr(w12_m1, img_h, img_w)
    w12_m2 = _scale_to_hr(w12_m2, img_h, img_w)
    w23_m1 = _scale_to_hr(w23_m1, img_h, img_w)
    w23_m2 = _scale_to_hr(w23_m2, img_h, img_w)
    off = (w12_m2 - w23_m1).reshape(w12_m2.shape[0], w12_m2.shape[1], -1, 2).mean(dim=2)
    off = off.unsqueeze(2).unsqueeze(2)
    w23_m1 = w23_m1 + off
    w23_m2 = w23_m2 + off
    middle = (w12_m2 + w23_m1) / 2.0

    wmin, wmax, hmin, hmax = _bbox([w12_m1, w12_m2, w23_m1, w23_m2])
    out_w = wmax - wmin
    out_h = hmax - hmin

    def shift(m):
        return torch.stack((m[..., 0] - wmin, m[..., 1] - hmin), dim=4)
    w12_m1, w12_m2, w23_m1, w23_m2, middle = map(shift, (w12_m1, w12_m2, w23_m1, w23_m2, middle))

    new1, new3 = [], []
    for i in range(middle.shape[1]):
        n12_1 = G.norm_mesh(w12_m1[:, i], out_h, out_w)
        n12_2 = G.norm_mesh(w12_m2[:, i], out_h, out_w)
        n23_1 = G.norm_mesh(w23_m1[:, i], out_h, out_w)
        n23_2 = G.norm_mesh(w23_m2[:, i], out_h, out_w)
        nmid = G.norm_mesh(middle[:, i], out_h, out_w)
        new1.append(G.recover_mesh(S.tps_points(n12_1, n12_2, nmid), out_h, out_w))
        new3.append(G.recover_mesh(S.tps_points(n23_2, n23_1, nmid), out_h, out_w))
    return torch.stack(new1, dim=1), middle, torch.stack(new3, dim=1)


def three_view_render(img1_list, img2_list, img3_list, mesh1, middle, mesh3, warp_mode, fusion_mode):
    """threeview:430-505 -> (list of [3,Hc,Wc] tensors, Wc, Hc)."""
    wmin, wmax, hmin, hmax = _bbox([mesh1, middle, mesh3])
    out_w = wmax - wmin
    out_h = hmax - hmin
    size = (int(out_h.int()), int(out_w.int()))
    b, _, img_h, img_w = img1_list[0].shape
    nrigid = G.norm_mesh(G.rigid_mesh(b, img_h, img_w), img_h, img_w)
    frames = []
    for i in range(mesh1.shape[1]):
        nm = []
        for m in (mesh1, middle, mesh3):
            a = m[:, i]
            nm.append(G.norm_mesh(torch.stack((a[..., 0] - wmin, a[..., 1] - hmin), dim=3), out_h, out_w))
        imgs = [img1_list[i], img2_list[i], img3_list[i]]
        src = torch.cat(nm, dim=0)
        tgt = torch.cat((nrigid, nrigid, nrigid), dim=0)
        if fusion_mode == 'AVERAGE':
            w = S.tps_warp(torch.cat(imgs, dim=0), src, tgt, size, warp_mode)
            f12 = average_fusion(w[0], w[1])
            fused = average_fusion(f12, w[2])
        else:
            one = torch.ones_like(imgs[0][:, 0:1])
            w = S.tps_warp(torch.cat([torch.cat((im, one), 1) for im in imgs], 0), src, tgt, size, warp_mode)
            k1, k2, k3 = w[0:1, 3:4], w[1:2, 3:4], w[2:3, 3:4]
            f12 = linear_blender(w[0:1, 0:3], w[1:2, 0:3], k1, k2)
            k12 = k1 + k2 - k1 * k2
            fused = linear_blender(f12, w[2:3, 0:3], k12, k3)[0]
        frames.append(fused)
    return frames, out_w.int(), out_h.int()


def run_three_view(hr1, hr2, hr3, lr1, lr2, lr3, nets, warp_mode='NORMAL', fusion_mode='AVERAGE'):
    a12 = estimate_meshes(nets, lr1, lr2)
    a23 = estimate_meshes(nets, lr2, lr3)
    _, _, img_h, img_w = hr1[0].shape
    m1, mid, m3 = three_view_compose(a12['smooth_mesh1'], a12['smooth_mesh2'],
                                     a23['smooth_mesh1'], a23['smooth_mesh2'], img_h, img_w)
    frames, wc, hc = three_view_render(hr1, hr2, hr3, m1, mid, m3, warp_mode, fusion_mode)
    return frames, int(hc), int(wc), m1, mid, m3
