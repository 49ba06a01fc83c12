"""Alignment / stability / distortion metrics (oracle, CPU, fp64 where skimage is).

Reference sites (relative to /root/reference/Full_model_inference/Codes):
  LR warps with 3 ones-mask channels   test_metric_ssd.py:151-181
  PSNR / SSIM on the overlap           test_metric_ssd.py:513-527 (scikit-image 0.15 compare_psnr/compare_ssim,
                                       restated: 7x7 uniform window, reflect border, K1=.01 K2=.03,
                                       sample covariance, crop 3, channel mean, fp64)
  stability                            test_metric_ssd.py:444-469
  distortion                           test_metric_ssd.py:38-87, 473-482
"""
import numpy as np
import torch
import torch.nn.functional as F
from scipy.ndimage import uniform_filter

from . import GRID_H, GRID_W
from . import geometry as G
from . import samplers as S


def psnr(a, b, data_range=255.0):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    mse = np.mean((a - b) ** 2)
    return 10.0 * np.log10(data_range ** 2 / mse)


def ssim(a, b, data_range=255.0, win=7):
    """multichannel SSIM, a/b [H,W,C]."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    c1 = (0.01 * data_range) ** 2
    c2 = (0.03 * data_range) ** 2
    npx = win * win
    cov_norm = npx / (npx - 1.0)
    pad = (win - 1) // 2
    vals = []
    for ch in range(a.shape[2]):
        x, y = a[..., ch], b[..., ch]
        ux = uniform_filter(x, size=win)
        uy = uniform_filter(y, size=win)
        uxx = uniform_filter(x * x, size=win)
        uyy = uniform_filter(y * y, size=win)
        uxy = uniform_filter(x * y, size=win)
        vx = cov_norm * (uxx - ux * ux)
        vy = cov_norm * (uyy - uy * uy)
        vxy = cov_norm * (uxy - ux * uy)
        s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux ** 2 + uy ** 2 + c1) * (vx + vy + c2))
        vals.append(s[pad:-pad, pad:-pad].mean())
    return float(np.mean(vals))


def warp_lr_with_mask(lr_list, smooth_mesh):
    """LR frames in [-1,1] -> list of [H,W,6] (3 colour + 3 mask channels), NORMAL warp at LR size."""
    b, _, h, w = lr_list[0].shape
    nrigid = G.norm_mesh(G.rigid_mesh(b, h, w), h, w)
    out = []
    for i, fr in enumerate(lr_list):
        img = (fr + 1) * 127.5
        nm = G.norm_mesh(smooth_mesh[:, i], h, w)
        wp = S.tps_warp(torch.cat((img, torch.ones_like(img)), 1), nm, nrigid, (h, w), 'NORMAL')
        out.append(wp[0].numpy().transpose(1, 2, 0))
    return out


def alignment_psnr_ssim(w1, w2):
    ov = w1[..., 3:6] * w2[..., 3:6]
    return psnr(w1[..., 0:3] * ov, w2[..., 0:3] * ov), ssim(w1[..., 0:3] * ov, w2[..., 0:3] * ov)


def stability_score(path):
    """path [B,T,h,w,2] (stitched smooth path of view 2)."""
    def l2(a, b):
        return torch.mean(torch.abs((a - b) ** 2))
    mid = path[:, 3:-3]
    s = (l2(path[:, :-6], mid) + l2(path[:, 6:], mid)) * 0.1
    s = s + (l2(path[:, 1:-5], mid) + l2(path[:, 5:-1], mid)) * 0.3
    s = s + (l2(path[:, 2:-4], mid) + l2(path[:, 4:-2], mid)) * 0.9
    return float(s)


def inter_grid(mesh):
    """NB: on the 5-D [B,T,h,w,2] meshes the harness passes, the reference reduces over
    dim 3 (the vertex-column axis), not the (x,y) axis (test_metric_ssd.py:45,56), and the
    following slices act on what is left; reproduced as executed."""
    we = mesh[:, :, :, 0:GRID_W] - mesh[:, :, :, 1:GRID_W + 1]
    a, b = we[:, :, :, 0:GRID_W - 1], we[:, :, :, 1:GRID_W]
    cw = (a * b).sum(3) / (torch.sqrt((a * a).sum(3)) * torch.sqrt((b * b).sum(3)))
    dw = 1 - cw
    dw = dw[:, :, 0:GRID_H] + dw[:, :, 1:GRID_H + 1]
    he = mesh[:, :, 0:GRID_H] - mesh[:, :, 1:GRID_H + 1]
    a, b = he[:, :, 0:GRID_H - 1], he[:, :, 1:GRID_H]
    ch = (a * b).sum(3) / (torch.sqrt((a * a).sum(3)) * torch.sqrt((b * b).sum(3)))
    dh = 1 - ch
    dh = dh[:, :, :, 0:GRID_W] + dh[:, :, :, 1:GRID_W + 1]
    return dw.mean() + dh.mean()


def intra_grid(mesh):
    max_w = 480 / GRID_W * 2
    max_h = 360 / GRID_H * 2
    dx = mesh[:, :, :, 1:GRID_W + 1, 0] - mesh[:, :, :, 0:GRID_W, 0]
    dy = mesh[:, :, 1:GRID_H + 1, :, 1] - mesh[:, :, 0:GRID_H, :, 1]
    return F.relu(dx - max_w).mean() + F.relu(dy - max_h).mean()


def distortion_score(mesh):
    """mesh [B,T,h,w,2]: max over frames of inter + intra grid terms."""
    return max(float(inter_grid(mesh[:, k:k + 1]) + intra_grid(mesh[:, k:k + 1]))
               for k in range(mesh.shape[1]))
