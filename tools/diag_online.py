"""Batch-1 (online) per-frame cost of the stages with the current eager launch path."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from stabstitch2_amd import synth, pipeline, ops
from stabstitch2_amd.spatial_network import build_SpatialNet
dev = torch.device('cuda:0'); torch.set_grad_enabled(False)
nets, _ = bench.build_nets(dev)
hr, lr = synth.make_clip_device(8, 720, 1280, 0, device=dev)
def T(name, fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): r = fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print('%-28s enqueue %.3f ms   total %.3f ms' % (name, (t1 - t) / n * 1e3, (t2 - t) / n * 1e3), flush=True); return r
s = T('spatial 1 pair', lambda: build_SpatialNet(nets[0], lr[0][:1], lr[1][:1]))
t = T('temporal 2 views x 2 frames', lambda: nets[1].motions_views([lr[0][:2], lr[1][:2]]))
acc = pipeline.estimate_meshes(nets, lr[0], lr[1])
sm = T('smooth 1 window', lambda: nets[2].run_windows(acc['ori_mesh1'][0, :7].contiguous(), acc['ori_mesh2'][0, :7].contiguous(), acc['tsmotion1'][:7].contiguous(), acc['tsmotion2'][:7].contiguous(), 1, 7, 1, 1))
m1 = acc['smooth_mesh1'][:, :1].contiguous(); m2 = acc['smooth_mesh2'][:, :1].contiguous()
T('render 1 frame (plan+warp)', lambda: pipeline.render_frames([hr[0][:1], hr[1][:1]], [m1, m2]))
