import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, 'tests/golden')
import cases
from stabstitch2_amd import ops, pipeline, synth
from oracle import samplers as S, pipeline as OP, geometry as G
dev = torch.device('cuda:0')
torch.set_grad_enabled(False)
def report(a, b, k, name):
    d = np.abs(a - b)
    ma = cases.box_down(a, k); mb = cases.box_down(b, k)
    dm = np.abs(ma - mb).max(axis=2)
    iy, ix = np.unravel_index(np.argmax(dm), dm.shape)
    print(name, 'pix max', d.max(), 'mean', d.mean(), 'median-box max', dm.max(), 'at box', iy, ix, 'of', dm.shape)
    blk_a = a[iy*k:(iy+1)*k, ix*k:(ix+1)*k, 0]; blk_b = b[iy*k:(iy+1)*k, ix*k:(ix+1)*k, 0]
    print('  dev block ch0 row0', blk_a[0, :8]); print('  ora block ch0 row0', blk_b[0, :8])
    bad = (d.max(axis=2) > 1.0)
    ys, xs = np.where(bad)
    print('  n>1.0:', bad.sum(), 'x range', xs.min() if len(xs) else None, xs.max() if len(xs) else None, 'y range', ys.min() if len(ys) else None, ys.max() if len(ys) else None)
g = np.load('tests/golden/g9_pipeline.npz')
hr, lr = synth.make_clip(16, 360, 480, seed=0)
m1 = torch.from_numpy(g['smooth_mesh1']); m2 = torch.from_numpy(g['smooth_mesh2'])
of, ow, oh = OP.get_stable_sqe(hr[0], hr[1], m1, m2, 'NORMAL', 'AVERAGE')
fr, hc, wc = pipeline.render_frames([hr[0], hr[1]], [m1.to(dev), m2.to(dev)], 'NORMAL', 'AVERAGE')
for i in range(16):
    report(fr[i].permute(1, 2, 0).cpu().numpy(), of[i], 16, '2view f%d' % i)
g10 = np.load('tests/golden/g10_threeview.npz')
hr3, _ = synth.make_clip(4, 180, 320, seed=3, views=3)
gm = [torch.from_numpy(g10[k]) for k in ('mesh1', 'middle', 'mesh3')]
of3, ow3, oh3 = OP.three_view_render(hr3[0], hr3[1], hr3[2], *gm, 'NORMAL', 'AVERAGE')
fr3, hc3, wc3 = pipeline.three_view_render(hr3[0], hr3[1], hr3[2], *[m.to(dev) for m in gm], 'NORMAL', 'AVERAGE')
for i in range(4):
    report(fr3[i].permute(1, 2, 0).cpu().numpy(), of3[i].permute(1, 2, 0).numpy(), 4, '3view f%d' % i)
