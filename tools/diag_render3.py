import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, 'tests/golden')
import cases
from stabstitch2_amd import ops, pipeline, synth
from oracle import pipeline as OP
dev = torch.device('cuda:0')
torch.set_grad_enabled(False)
g = np.load('tests/golden/g9_pipeline.npz')
hr, lr = synth.make_clip(16, 360, 480, seed=0)
m1 = torch.from_numpy(g['smooth_mesh1']); m2 = torch.from_numpy(g['smooth_mesh2'])
of, ow, oh = OP.get_stable_sqe(hr[0], hr[1], m1, m2, 'NORMAL', 'AVERAGE')
frames, ow2, oh2 = pipeline.get_stable_sqe(hr[0], hr[1], m1.to(dev), m2.to(dev), 'NORMAL', 'AVERAGE')
ok = cases.smooth_boxes(g['iqr_normal_average'])
gd = np.stack([cases.box_down(f, 16) for f in frames]); go = np.stack([cases.box_down(f, 16) for f in of])
ref = g['frames_normal_average']
dd = np.abs(np.where(ok, gd, 0) - np.where(ok, ref, 0)); do = np.abs(np.where(ok, go, 0) - np.where(ok, ref, 0))
print('dev vs golden per frame', dd.reshape(16, -1).max(axis=1))
print('box-oracle vs golden per frame', do.reshape(16, -1).max(axis=1))
i = np.unravel_index(np.argmax(dd), dd.shape); print(i, gd[i], go[i], ref[i], g['iqr_normal_average'][i])
f, by, bx, c = i
blk = frames[f][by*16:(by+1)*16, bx*16:(bx+1)*16, c]; blo = of[f][by*16:(by+1)*16, bx*16:(bx+1)*16, c]
print(np.round(blk[:4, :], 2)); print(np.round(blo[:4, :], 2))
