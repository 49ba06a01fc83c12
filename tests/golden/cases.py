"""Seeded inputs shared by the golden generator (build container only) and the parity tests
(here and on the GPU box).  Pure numpy/torch; never touches /root/reference."""
import numpy as np
import torch

GRID_H, GRID_W = 6, 8
LR_H, LR_W = 360, 480


def rigid(h, w):
    xs = np.linspace(0.0, float(w), GRID_W + 1, dtype=np.float32)
    ys = np.linspace(0.0, float(h), GRID_H + 1, dtype=np.float32)
    return np.stack(np.broadcast_arrays(xs[None, :], ys[:, None]), axis=2)[None].astype(np.float32)


def norm(mesh, h, w):
    out = mesh.copy()
    out[..., 0] = mesh[..., 0] * 2.0 / w - 1.0
    out[..., 1] = mesh[..., 1] * 2.0 / h - 1.0
    return out.reshape(mesh.shape[0], -1, 2).astype(np.float32)


def g1_offsets():
    rs = np.random.RandomState(101)
    return torch.from_numpy(rs.uniform(-60, 60, size=(16, 8)).astype(np.float32))


def g2_inputs():
    rs = np.random.RandomState(102)
    U = torch.from_numpy(rs.normal(0, 1, size=(3, 8, 45, 60)).astype(np.float32))
    thetas = np.stack([
        np.eye(3),
        np.array([[1.02, 0.03, 0.10], [-0.02, 0.97, -0.05], [0.01, -0.02, 1.0]]),
        np.array([[0.8, 0.1, 0.9], [0.05, 1.3, -0.7], [0.15, 0.1, 1.0]]),   # far out of bounds
    ]).astype(np.float32)
    return U, torch.from_numpy(thetas)


def g3_inputs(full=False):
    rs = np.random.RandomState(103 + int(full))
    shape = (1, 128, 45, 60) if full else (2, 16, 9, 12)
    a = torch.from_numpy(rs.normal(0, 1, size=shape).astype(np.float32))
    b = torch.from_numpy(rs.normal(0, 1, size=shape).astype(np.float32))
    return a, b


def g4_inputs(full=False):
    rs = np.random.RandomState(105 + int(full))
    shape = (1, 256, 23, 30) if full else (2, 32, 5, 6)
    a = torch.from_numpy(np.abs(rs.normal(0, 1, size=shape)).astype(np.float32))
    # second map = noisy shifted copy so that the soft-argmax is not flat
    b = torch.roll(a, shifts=(1, -1), dims=(2, 3)) + torch.from_numpy(
        (0.3 * rs.normal(0, 1, size=shape)).astype(np.float32))
    return a, b.abs()


def g5_meshes(n=8, h=LR_H, w=LR_W, seed=107, sigma=6.0):
    """-> (rigid_norm [n,63,2], warped_norm [n,63,2], query_norm [n,63,2])"""
    rs = np.random.RandomState(seed)
    r = np.repeat(rigid(h, w), n, axis=0)
    warped = r + rs.normal(0, sigma, size=r.shape).astype(np.float32) + \
        rs.uniform(-40, 40, size=(n, 1, 1, 2)).astype(np.float32)
    query = r + rs.normal(0, sigma, size=r.shape).astype(np.float32)
    return (torch.from_numpy(norm(r, h, w)), torch.from_numpy(norm(warped, h, w)),
            torch.from_numpy(norm(query, h, w)))


def g6_inputs():
    """Smooth texture + coordinate ramps: U [2,5,72,96] = 3 texture channels, x-ramp, y-ramp."""
    from stabstitch2_amd import synth
    h, w = 72, 96
    imgs = []
    for v in range(2):
        tex = synth.texture_window(20.0 + 40 * v, 10.0, h, w)
        xr = np.broadcast_to(np.arange(w, dtype=np.float32)[None, :], (h, w))
        yr = np.broadcast_to(np.arange(h, dtype=np.float32)[:, None], (h, w))
        imgs.append(np.concatenate([tex, xr[None], yr[None]], axis=0))
    U = torch.from_numpy(np.stack(imgs).astype(np.float32))
    out_h, out_w = 80, 120
    rs = np.random.RandomState(108)
    r = np.repeat(rigid(h, w), 2, axis=0)
    m = r + rs.normal(0, 1.5, size=r.shape).astype(np.float32)
    m[0, ..., 0] += 2.0
    m[1, ..., 0] += 22.0
    m[..., 1] += 4.0
    src = torch.from_numpy(norm(m, out_h, out_w))
    tgt = torch.from_numpy(norm(r, h, w))
    ident_src = torch.from_numpy(norm(np.repeat(rigid(h, w), 2, axis=0), h, w))
    return U, src, tgt, (out_h, out_w), ident_src


def g10_meshes(n=4, seed=110):
    """four [1,n,7,9,2] LR-scale mesh tensors resembling two 2-view passes (v1,v2),(v2,v3)."""
    rs = np.random.RandomState(seed)
    r = rigid(LR_H, LR_W)[:, None]
    def jit():
        return rs.normal(0, 1.5, size=(1, n, GRID_H + 1, GRID_W + 1, 2)).astype(np.float32)
    shift = np.zeros((1, 1, 1, 1, 2), np.float32)
    shift[..., 0] = 90.0
    m12_1 = r - shift + jit()
    m12_2 = r + shift + jit()
    m23_1 = r - shift + jit()
    m23_2 = r + shift + jit()
    return tuple(torch.from_numpy(x.astype(np.float32)) for x in (m12_1, m12_2, m23_1, m23_2))


def g11_images():
    rs = np.random.RandomState(111)
    a = rs.uniform(0, 255, size=(36, 48, 3)).astype(np.float32)
    b = np.clip(a + rs.normal(0, 12, size=a.shape), 0, 255).astype(np.float32)
    return a, b


def box_down(frame, k=8):
    """[H,W,C] -> per-box MEDIAN [H//k, W//k, C] (crop remainder).  Median, not mean: the reference's AVERAGE
    fusion a*a/(a+b+1e-6) is singular where a+b ~ -1e-6 (outside the valid region the clamped-weight sampler leaves
    +-1e-5 residues instead of exact zeros), so isolated pixels are arbitrarily large in any implementation."""
    h, w, c = frame.shape
    hh, ww = (h // k) * k, (w // k) * k
    f = frame[:hh, :ww].reshape(hh // k, k, ww // k, k, c).transpose(0, 2, 4, 1, 3).reshape(hh // k, ww // k, c, k * k)
    return np.median(f, axis=3).astype(np.float32)


def box_iqr(frame, k=8):
    """Value range (max - min) per box.  Boxes straddling a view boundary (part ~0, part image) have a median that
    flips with a one-pixel shift of the boundary; the comparisons skip them (smooth_boxes).  (Kept under the name
    `iqr` in the fixtures.)"""
    h, w, c = frame.shape
    hh, ww = (h // k) * k, (w // k) * k
    f = frame[:hh, :ww].reshape(hh // k, k, ww // k, k, c).transpose(0, 2, 4, 1, 3).reshape(hh // k, ww // k, c, k * k)
    return (f.max(axis=3) - f.min(axis=3)).astype(np.float32)


def smooth_boxes(rng, k=16):
    """Boolean mask of boxes whose golden value range is explained by the smooth synthetic texture (< ~3.5 / px)."""
    return rng < 3.5 * k + 8.0
