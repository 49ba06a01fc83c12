import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, 'tests/golden')
import cases
from stabstitch2_amd import ops, pipeline, synth
from oracle import samplers as S, pipeline as OP, geometry as G
dev = torch.device('cuda:0')
torch.set_grad_enabled(False)
g = np.load('tests/golden/g9_pipeline.npz')
hr, lr = synth.make_clip(16, 360, 480, seed=0)
m1 = torch.from_numpy(g['smooth_mesh1']); m2 = torch.from_numpy(g['smooth_mesh2'])
of, ow, oh = OP.get_stable_sqe(hr[0][:2], hr[1][:2], m1, m2, 'NORMAL', 'AVERAGE')
fr, hc, wc = pipeline.render_frames([hr[0], hr[1]], [m1.to(dev), m2.to(dev)], 'NORMAL', 'AVERAGE')
a = fr[0].permute(1, 2, 0).cpu().numpy(); b = of[0]
d = np.abs(a - b)
print('canvas', hc, wc, int(oh), int(ow), 'max', d.max(), 'mean', d.mean(), 'frac>0.05', (d > 0.05).mean())
ys, xs = np.where(d.max(axis=2) > 0.5)
print('n bad', len(ys), 'rows', ys[:20], 'cols', xs[:20])
if len(ys):
    print('vals', a[ys[0], xs[0]], b[ys[0], xs[0]])
# plan pieces
hc2, wc2, src, T = pipeline.render_plan([m1.to(dev), m2.to(dev)], 360, 480)
rigid = G.rigid_mesh(1, 360, 480); nrigid = G.norm_mesh(rigid, 360, 480)
mm1 = OP._scale_to_hr(m1, 360, 480); mm2 = OP._scale_to_hr(m2, 360, 480)
wmin, wmax, hmin, hmax = OP._bbox([mm1, mm2])
a0 = mm1[:, 0]
nm1 = G.norm_mesh(torch.stack((a0[..., 0] - wmin, a0[..., 1] - hmin), 3), hmax - hmin, wmax - wmin)
print('src diff', float((src[0, 0].cpu() - nm1[0]).abs().max()))
To = S.tps_solve(nm1, nrigid)
print('T diff', float((T[0, 0].cpu() - To[0]).abs().max()), float(To.abs().max()))
# three view
g10 = np.load('tests/golden/g10_threeview.npz')
hr3, _ = synth.make_clip(4, 180, 320, seed=3, views=3)
gm = [torch.from_numpy(g10[k]) for k in ('mesh1', 'middle', 'mesh3')]
of3, ow3, oh3 = OP.three_view_render(hr3[0], hr3[1], hr3[2], *gm, 'NORMAL', 'AVERAGE')
fr3, hc3, wc3 = pipeline.three_view_render(hr3[0], hr3[1], hr3[2], *[m.to(dev) for m in gm], 'NORMAL', 'AVERAGE')
d3 = (fr3[0].cpu() - of3[0]).abs()
print('3view canvas', hc3, wc3, int(oh3), int(ow3), 'max', float(d3.max()), 'mean', float(d3.mean()))
print('per-view check')
hcp, wcp, src3, T3 = pipeline.render_plan([m.to(dev) for m in gm], 180, 320, prescaled=True)
wminb = OP._bbox(gm)
print('bbox oracle', [float(x) for x in wminb])
print('bbox dev', ops.mesh_bbox([m.to(dev) for m in gm], 0.0, 0.0).cpu())
