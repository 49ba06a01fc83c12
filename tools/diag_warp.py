import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, 'tests/golden')
import cases
from stabstitch2_amd import ops
from oracle import samplers as S
dev = torch.device('cuda:0')
g = np.load('tests/golden/g6_tps_warp.npz')
U, src, tgt, size, ident = cases.g6_inputs()
Ud, sd, td = U.to(dev), src.to(dev), tgt.to(dev)
T = ops.tps_solve(sd, td)
To = S.tps_solve(src, tgt)
print('T max diff', float((T.cpu() - To).abs().max()), 'T max', float(To.abs().max()))
for mode in ('NORMAL', 'FAST'):
    w = ops.tps_warp(Ud, sd, T, size[0], size[1], mode).cpu().numpy()
    r = g[mode.lower()]
    d = np.abs(w - r)
    for ch in range(5):
        print(mode, 'ch', ch, 'max', d[:, ch].max(), 'mean', d[:, ch].mean(), 'frac>2e-3', (d[:, ch] > 2e-3).mean())
# coordinates directly: oracle fp32 vs fp64 vs device (via ramp in interior)
xn, yn = S.tps_dense_coords(src, tgt, size[0], size[1])
xd, yd = S.tps_dense_coords(src.double(), tgt.double(), size[0], size[1]) if False else (None, None)
