"""Mesh / homography helpers (oracle, CPU).

Reference sites (relative to /root/reference/Full_model_inference/Codes):
  rigid mesh           spatial_network.py:39-50,  test_online_tra.py:71-83
  normalise / recover  spatial_network.py:53-59,  test_online_tra.py:61-69, 85-91
  4-point DLT          utils/torch_DLT.py:17-45
  H -> mesh            spatial_network.py:20-36
  decomposition        spatial_network.py:72-104, 291-300
"""
import torch

from . import GRID_H, GRID_W


def rigid_mesh(batch, height, width):
    """[B, GRID_H+1, GRID_W+1, 2] regular vertex grid, (x, y) in pixels."""
    xs = torch.linspace(0.0, float(width), GRID_W + 1)
    ys = torch.linspace(0.0, float(height), GRID_H + 1)
    gx = xs.view(1, -1).expand(GRID_H + 1, -1)
    gy = ys.view(-1, 1).expand(-1, GRID_W + 1)
    m = torch.stack((gx, gy), dim=2)
    return m.unsqueeze(0).expand(batch, -1, -1, -1)


def norm_mesh(mesh, height, width):
    """pixels -> [-1, 1]; output flattened to [B, P, 2]."""
    b = mesh.shape[0]
    x = mesh[..., 0] * 2.0 / float(width) - 1.0
    y = mesh[..., 1] * 2.0 / float(height) - 1.0
    return torch.stack((x, y), dim=-1).reshape(b, -1, 2)


def recover_mesh(nmesh, height, width):
    """[-1, 1] flattened [B, P, 2] -> pixels [B, GRID_H+1, GRID_W+1, 2]."""
    b = nmesh.shape[0]
    x = (nmesh[..., 0] + 1.0) * float(width) / 2.0
    y = (nmesh[..., 1] + 1.0) * float(height) / 2.0
    return torch.stack((x, y), dim=2).reshape(b, GRID_H + 1, GRID_W + 1, 2)


def dlt4(src, dst):
    """Homography from 4 correspondences, src/dst [B,4,2] -> H [B,3,3] (src -> dst).

    Rows per corner (x,y)->(u,v):  [x y 1 0 0 0 -ux -uy] = u
                                   [0 0 0 x y 1 -vx -vy] = v
    solved with an explicit fp32 inverse like utils/torch_DLT.py:41-42.
    """
    b = src.shape[0]
    x, y = src[..., 0], src[..., 1]
    u, v = dst[..., 0], dst[..., 1]
    one = torch.ones_like(x)
    zero = torch.zeros_like(x)
    row_u = torch.stack((x, y, one, zero, zero, zero, -(u * x), -(u * y)), dim=-1)
    row_v = torch.stack((zero, zero, zero, x, y, one, -(v * x), -(v * y)), dim=-1)
    A = torch.stack((row_u, row_v), dim=2).reshape(b, 8, 8)
    rhs = dst.reshape(b, 8, 1)
    h = torch.matmul(torch.inverse(A), rhs).reshape(b, 8)
    return torch.cat((h, torch.ones(b, 1, dtype=h.dtype)), dim=1).reshape(b, 3, 3)


def corners(batch, height, width):
    c = torch.tensor([[0.0, 0.0], [width, 0.0], [0.0, height], [width, height]])
    return c.unsqueeze(0).expand(batch, -1, -1)


def decompose(offset8, height, width, scale=1.0):
    """Bidirectional decomposition onto the virtual middle plane.

    offset8 [B,8] -> (H, H_tgt, H_ref) with H = DLT(c, c+m), H_tgt = DLT(c, c+m/2),
    H_ref = H^-1 H_tgt; all corner points divided by `scale`
    (8 inside SpatialNet.forward, 1 in build_SpatialNet).
    """
    b = offset8.shape[0]
    m = offset8.reshape(b, 4, 2)
    c = corners(b, float(height), float(width))
    H = dlt4(c / scale, (c + m) / scale)
    H_tgt = dlt4(c / scale, (c + m / 2.0) / scale)
    H_ref = torch.matmul(torch.inverse(H), H_tgt)
    return H, H_tgt, H_ref


def homography_to_mesh(H, rmesh):
    """mesh vertices = persp_divide(H^-1 [x y 1]^T) over the rigid vertices."""
    b = rmesh.shape[0]
    pts = rmesh.reshape(b, -1, 2)
    hom = torch.cat((pts, torch.ones(b, pts.shape[1], 1)), dim=2)
    t = torch.matmul(torch.inverse(H), hom.transpose(1, 2))
    mx = t[:, 0, :] / t[:, 2, :]
    my = t[:, 1, :] / t[:, 2, :]
    return torch.stack((mx, my), dim=2).reshape(b, GRID_H + 1, GRID_W + 1, 2)
