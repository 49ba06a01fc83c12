"""Homography / thin-plate-spline samplers (oracle, CPU).

Reference sites (relative to /root/reference/Full_model_inference/Codes):
  bilinear core (clamped-index weights)   utils/torch_homo_transform.py:50-125
                                          utils/torch_tps_transform.py:30-106
  projective grid                         utils/torch_homo_transform.py:127-180
  TPS system solve (fp64 inverse)         utils/torch_tps_transform.py:168-226
  TPS dense evaluation                    utils/torch_tps_transform.py:108-165
  TPS at points                           utils/torch_tps_transform_point.py:21-80
"""
import torch
import torch.nn.functional as F


def bilinear_clamped(img, xn, yn):
    """The reference's hand-rolled bilinear gather.

    img [B,C,H,W]; xn, yn [B,P] in the [-1,1] convention where x=(xn+1)*W/2.
    Indices are clamped to the image and the CLAMPED values are used as floats in
    the weights, so out-of-range taps get weight exactly 0.  Returns [B,C,P].
    """
    b, c, h, w = img.shape
    x = (xn + 1.0) * float(w) / 2.0
    y = (yn + 1.0) * float(h) / 2.0
    x0 = torch.floor(x).to(torch.int32)
    y0 = torch.floor(y).to(torch.int32)
    x1 = x0 + 1
    y1 = y0 + 1
    x0 = x0.clamp(0, w - 1)
    x1 = x1.clamp(0, w - 1)
    y0 = y0.clamp(0, h - 1)
    y1 = y1.clamp(0, h - 1)
    flat = img.reshape(b, c, h * w)

    def tap(yy, xx):
        idx = (yy.long() * w + xx.long()).unsqueeze(1).expand(-1, c, -1)
        return torch.gather(flat, 2, idx)

    x0f, x1f, y0f, y1f = x0.float(), x1.float(), y0.float(), y1.float()
    wa = ((x1f - x) * (y1f - y)).unsqueeze(1)
    wb = ((x1f - x) * (y - y0f)).unsqueeze(1)
    wc = ((x - x0f) * (y1f - y)).unsqueeze(1)
    wd = ((x - x0f) * (y - y0f)).unsqueeze(1)
    return wa * tap(y0, x0) + wb * tap(y1, x0) + wc * tap(y0, x1) + wd * tap(y1, x1)


def homography_coords(theta, out_h, out_w):
    """Normalised sampling coordinates of the projective warp, [B, out_h*out_w] each."""
    b = theta.shape[0]
    theta = theta.reshape(b, 3, 3).float()
    gx = torch.linspace(-1.0, 1.0, out_w).view(1, -1).expand(out_h, -1).reshape(-1)
    gy = torch.linspace(-1.0, 1.0, out_h).view(-1, 1).expand(-1, out_w).reshape(-1)
    grid = torch.stack((gx, gy, torch.ones_like(gx)), dim=0)  # [3,P]
    t = torch.matmul(theta, grid.unsqueeze(0).expand(b, -1, -1))
    ts = t[:, 2, :]
    ts = ts + 1e-6 * (1.0 - (ts.abs() >= 1e-7).float())
    return t[:, 0, :] / ts, t[:, 1, :] / ts


def homography_warp(img, theta, out_size):
    """torch_homo_transform.transformer: img [B,C,H,W], theta [B,3,3]|[B,9]."""
    out_h, out_w = int(out_size[0]), int(out_size[1])
    xn, yn = homography_coords(theta, out_h, out_w)
    out = bilinear_clamped(img, xn, yn)
    return out.reshape(img.shape[0], img.shape[1], out_h, out_w)


def _rbf(d2):
    return d2 * torch.log(d2 + 1e-6)


def tps_solve(source, target):
    """TPS coefficients T [B,2,P+3] mapping `source` control points onto `target`.

    System [[P, R],[0, P^T]] (P = [1, Sx, Sy], R_ij = d2 log(d2+1e-6), fp32),
    inverted in fp64, applied to [target; 0], cast to fp32.
    """
    b, n, _ = source.shape
    p = torch.cat((torch.ones(b, n, 1), source.float()), dim=2)  # [B,n,3]
    diff = p.unsqueeze(2) - p.unsqueeze(1)
    d2 = (diff * diff).sum(dim=3)
    r = _rbf(d2)
    top = torch.cat((p, r), dim=2)
    bot = torch.cat((torch.zeros(b, 3, 3), p.transpose(1, 2)), dim=2)
    W = torch.cat((top, bot), dim=1).double()
    rhs = torch.cat((target.float(), torch.zeros(b, 3, 2)), dim=1).double()
    T = torch.matmul(torch.inverse(W), rhs)
    return T.transpose(1, 2).float()


def tps_eval(T, source, xq, yq):
    """Evaluate the spline at query coords xq,yq [B,Q] (or [1,Q]) -> (xs, ys) [B,Q]."""
    b = source.shape[0]
    xq = xq.expand(b, -1) if xq.shape[0] != b else xq
    yq = yq.expand(b, -1) if yq.shape[0] != b else yq
    px = source[:, :, 0:1]
    py = source[:, :, 1:2]
    dx = xq.unsqueeze(1) - px
    dy = yq.unsqueeze(1) - py
    d2 = dx * dx + dy * dy
    r = _rbf(d2)
    phi = torch.cat((torch.ones_like(xq).unsqueeze(1), xq.unsqueeze(1), yq.unsqueeze(1), r), dim=1)
    out = torch.matmul(T, phi)
    return out[:, 0, :], out[:, 1, :]


def tps_dense_coords(source, target, out_h, out_w, chunk=1 << 16):
    """Normalised source-image sampling coords for every canvas pixel, [B, out_h*out_w]."""
    T = tps_solve(source, target)
    gx = torch.linspace(-1.0, 1.0, out_w).view(1, -1).expand(out_h, -1).reshape(1, -1)
    gy = torch.linspace(-1.0, 1.0, out_h).view(-1, 1).expand(-1, out_w).reshape(1, -1)
    xs, ys = [], []
    for s in range(0, gx.shape[1], chunk):
        a, c = tps_eval(T, source, gx[:, s:s + chunk], gy[:, s:s + chunk])
        xs.append(a)
        ys.append(c)
    return torch.cat(xs, dim=1), torch.cat(ys, dim=1)


def tps_warp(img, source, target, out_size, mode='NORMAL'):
    """torch_tps_transform.transformer.

    img [B,C,H,W]; source = warped mesh (canvas-normalised), target = rigid mesh
    (image-normalised), i.e. a backward map canvas -> input image.
    NORMAL = clamped bilinear core, FAST = F.grid_sample(align_corners=True).
    """
    out_h, out_w = int(out_size[0]), int(out_size[1])
    b, c = img.shape[0], img.shape[1]
    xn, yn = tps_dense_coords(source, target, out_h, out_w)
    if mode == 'NORMAL':
        return bilinear_clamped(img, xn, yn).reshape(b, c, out_h, out_w)
    grid = torch.stack((xn.reshape(b, out_h, out_w), yn.reshape(b, out_h, out_w)), dim=3)
    return F.grid_sample(img, grid, mode='bilinear', padding_mode='zeros', align_corners=True)


def tps_points(point, source, target):
    """torch_tps_transform_point.transformer: point [B,Q,2] -> [B,Q,2]."""
    T = tps_solve(source, target)
    xs, ys = tps_eval(T, source, point[:, :, 0], point[:, :, 1])
    return torch.stack((xs, ys), dim=2)
