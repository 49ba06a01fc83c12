import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from stabstitch2_amd import synth
from stabstitch2_amd.online import MultiOnlineStitcher, OnlineStitcher
dev = torch.device('cuda:0'); torch.set_grad_enabled(False)
nets, _ = bench.build_nets(dev)
S, n, h, w = 2, 9, 360, 480
hrs, lrs = [], []
for sd in range(S):
    a, b = synth.make_clip_device(n, h, w, seed=sd, device=dev); hrs.append(a); lrs.append(b)
hr = [torch.stack([hrs[s][v] for s in range(S)], 0) for v in range(2)]
lr = [torch.stack([lrs[s][v] for s in range(S)], 0) for v in range(2)]
multi = MultiOnlineStitcher(nets, h, w, streams=S, use_graph=False)
single = [OnlineStitcher(nets, h, w, use_graph=False) for _ in range(S)]
for t in range(n):
    args = (hr[0][:, t].contiguous(), hr[1][:, t].contiguous(), lr[0][:, t].contiguous(), lr[1][:, t].contiguous())
    got = multi.push(*args)
    want = [single[s].push(args[0][s:s + 1], args[1][s:s + 1], args[2][s:s + 1], args[3][s:s + 1]) for s in range(S)]
    if t >= 6:
        st = multi.static
        for s in range(S):
            o = single[s].static
            print('t', t, 's', s, 'pair_s', float((st['pair_s'][:, :, s] - o['pair_s']).abs().max()),
                  'pair_t', float((st['pair_t'][:, :, s] - o['pair_t']).abs().max()),
                  'ring', [float((st['ring'][k, s] - o['ring'][k]).abs().max()) for k in range(4)],
                  'feat', float((st['prev_feat'][[s, S + s]] - o['prev_feat']).abs().max()),
                  'ts_out', float((st['ts_out'][:, [S + s, 3 * S + s]] - o['ts_out'][:, [1, 3]]).abs().max()) if t > 6 else None,
                  'frame', [float((a - b).abs().max()) for a, b in zip(got[s], want[s])][-1:])
print('---- render args')
rec = {}
orig = OnlineStitcher._render
def spy(self, hr1, hr2, mesh1, mesh2, out=None):
    r = orig(self, hr1, hr2, mesh1, mesh2, out)
    rec.setdefault(id(self), []).append((hr1.clone(), hr2.clone(), mesh1.clone(), mesh2.clone(), r.clone()))
    return r
OnlineStitcher._render = spy
multi = MultiOnlineStitcher(nets, h, w, streams=S, use_graph=False)
single = [OnlineStitcher(nets, h, w, use_graph=False) for _ in range(S)]
for t in range(8):
    args = (hr[0][:, t].contiguous(), hr[1][:, t].contiguous(), lr[0][:, t].contiguous(), lr[1][:, t].contiguous())
    multi.push(*args)
    [single[s].push(args[0][s:s + 1], args[1][s:s + 1], args[2][s:s + 1], args[3][s:s + 1]) for s in range(S)]
for s in range(S):
    a, b = rec[id(multi.single[s])][-1], rec[id(single[s])][-1]
    print(s, len(rec[id(multi.single[s])]), len(rec[id(single[s])]), [float((x.float() - y.float()).abs().max()) for x, y in zip(a, b)])
print('---- inside render')
from stabstitch2_amd import ops, pipeline
o = single[0]
a, b = rec[id(multi.single[0])][-1], rec[id(single[0])][-1]
res = []
for (h1, h2, m1, m2, r) in (a, b):
    src = ops.mesh_normalize_views([m1, m2], o.bbox, o.h, o.w)[0]
    T = ops.tps_solve_shared(src, o.nrigid)
    fp = ops.render_footprints(src[None], T[None], o.h, o.w, o.hc, o.wc)[0]
    out_fp = ops.render_average([h1, h2], src, T, o.hc, o.wc, 'NORMAL', footprint=fp)
    out_nofp = ops.render_average([h1, h2], src, T, o.hc, o.wc, 'NORMAL', footprint=None)
    res.append((src, T, fp, out_fp, out_nofp, r))
names = ('src', 'T', 'fp', 'out_fp', 'out_nofp', 'recorded')
print({n: float((x - y).abs().max()) for n, x, y in zip(names, res[0], res[1])})
print('recorded vs recomputed (multi):', float((res[0][3] - res[0][5]).abs().max()), ' (single):', float((res[1][3] - res[1][5]).abs().max()))
print('T absmax', float(res[0][1].abs().max()), float(res[1][1].abs().max()))
