"""Metric harness of the reference (test_metric_ssd.py) on the HIP engine.

  * warp_lr_with_mask      LR frames + ones-mask channels warped at LR size with the NORMAL sampler (:151-181)
  * alignment_psnr_ssim    PSNR / SSIM of the two masked warps, fp64 on device with scikit-image 0.15 semantics (:513-527)
  * stability_score        7-tap path differences of the stitched smooth path (:444-469)
  * distortion_score       inter + intra grid loss, max over frames (:38-87, 473-482)
"""
import ctypes

import torch

from . import _hip as H, ops
from .spatial_network import get_rigid_mesh, get_norm_mesh


@torch.no_grad()
def warp_lr_planes(lr, smooth_mesh):
    """lr [N,3,360,480] device in [-1,1]; smooth_mesh [1,N,7,9,2] -> [N,4,360,480] (3 colour 0..255 + mask)."""
    n, _, h, w = lr.shape
    dev = lr.device
    img = ops.add_mul(lr.contiguous().float(), 1.0, 127.5)
    nm = get_norm_mesh(smooth_mesh[0], h, w).contiguous()
    nrigid = get_norm_mesh(get_rigid_mesh(1, h, w, device=dev), h, w).expand(n, -1, -1).contiguous()
    T = ops.tps_solve(nm, nrigid)
    return ops.tps_warp(img, nm, T, h, w, 'NORMAL', with_mask=True)


@torch.no_grad()
def warp_lr_with_mask(lr, smooth_mesh):
    """Reference layout of get_stable_sqe in test_metric_ssd.py: [N,360,480,6] (3 colour + 3 copies of the mask)."""
    wp = warp_lr_planes(lr, smooth_mesh)
    out = torch.cat((wp[:, 0:3], wp[:, 3:4].expand(-1, 3, -1, -1)), 1)
    return out.permute(0, 2, 3, 1).contiguous()


@torch.no_grad()
def alignment_psnr_ssim(w1, w2):
    """w1, w2 [N,4,h,w] from warp_lr_planes -> (psnr [N], ssim [N]) fp64 device tensors."""
    n, c, h, w = w1.shape
    assert c == 4 and w2.shape == w1.shape
    out = torch.empty((n, 2), device=w1.device, dtype=torch.float64)
    ws = torch.empty((n, 2), device=w1.device, dtype=torch.float64)
    H.call('ss_alignment_psnr_ssim', H.dptr(w1), H.dptr(w2), H.dptr(out, dtype=out.dtype),
           H.dptr(ws, dtype=ws.dtype), n, h, w, H.stream())
    return out[:, 0], out[:, 1]


@torch.no_grad()
def stability_score(path):
    """path [1,T,7,9,2] (stitched smooth path of view 2) -> python float."""
    p = path[0].contiguous().float()
    out = torch.empty(1, device=p.device, dtype=torch.float32)
    H.call('ss_stability_score', H.dptr(p), H.dptr(out), p.shape[0], H.stream())
    return float(out)


@torch.no_grad()
def distortion_score(mesh):
    """mesh [1,T,7,9,2] (smooth mesh of view 2, LR px) -> python float."""
    m = mesh[0].contiguous().float()
    out = torch.empty(1, device=m.device, dtype=torch.float32)
    ws = torch.empty(m.shape[0], device=m.device, dtype=torch.float32)
    H.call('ss_distortion_score', H.dptr(m), H.dptr(out), H.dptr(ws), m.shape[0], H.stream())
    return float(out)


@torch.no_grad()
def evaluate_clip(nets, lr1, lr2):
    """test_metric_ssd.test() for one clip at the tensor level -> dict(psnr [N], ssim [N], stability, distortion)."""
    from . import pipeline
    acc = pipeline.estimate_meshes(nets, lr1, lr2)
    dev = acc['smooth_mesh1'].device
    if isinstance(lr1, (list, tuple)):
        lr1 = torch.cat([t.to(dev) for t in lr1], 0)
        lr2 = torch.cat([t.to(dev) for t in lr2], 0)
    p, s = alignment_psnr_ssim(warp_lr_planes(lr1, acc['smooth_mesh1']), warp_lr_planes(lr2, acc['smooth_mesh2']))
    return dict(psnr=p, ssim=s, stability=stability_score(acc['smooth_path2']),
                distortion=distortion_score(acc['smooth_mesh2']))
