"""Metric-harness warps (test_metric_ssd.py:151-181) on the HIP engine: LR frames + ones-mask channels
warped at LR size with the NORMAL sampler.  PSNR/SSIM themselves are fp64 host arithmetic in the
reference (scikit-image) and stay on the host side of the harness."""
import torch

from . import ops
from .spatial_network import get_rigid_mesh, get_norm_mesh


@torch.no_grad()
def warp_lr_with_mask(lr, smooth_mesh):
    """lr [N,3,360,480] device in [-1,1]; smooth_mesh [1,N,7,9,2] -> [N,360,480,6] (3 colour + 3 mask)."""
    n, _, h, w = lr.shape
    dev = lr.device
    img = ((lr + 1) * 127.5).contiguous()
    nm = get_norm_mesh(smooth_mesh[0], h, w).contiguous()
    nrigid = get_norm_mesh(get_rigid_mesh(1, h, w, device=dev), h, w).expand(n, -1, -1).contiguous()
    T = ops.tps_solve(nm, nrigid)
    wp = ops.tps_warp(img, nm, T, h, w, 'NORMAL', with_mask=True)          # [N,4,h,w]
    out = torch.cat((wp[:, 0:3], wp[:, 3:4].expand(-1, 3, -1, -1)), 1)
    return out.permute(0, 2, 3, 1).contiguous()
