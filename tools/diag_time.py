import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from stabstitch2_amd import synth, pipeline, ops
dev = torch.device('cuda:0'); torch.set_grad_enabled(False)
H, W, F = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
t = time.time(); nets, sds = bench.build_nets(dev); torch.cuda.synchronize(); print('nets', time.time() - t, flush=True)
t = time.time(); hr, lr = synth.make_clip_device(F, H, W, 0, device=dev); torch.cuda.synchronize(); print('clip', time.time() - t, flush=True)
def T(name, fn):
    torch.cuda.synchronize(); t = time.time(); r = fn(); torch.cuda.synchronize(); print('%-10s %.4f s' % (name, time.time() - t), flush=True); return r
for rep in range(2):
    s = T('spatial', lambda: pipeline.spatial_stage(nets[0], lr[0], lr[1]))
    t1 = T('temporal1', lambda: pipeline.temporal_stage(nets[1], lr[0]))
    t2 = T('temporal2', lambda: pipeline.temporal_stage(nets[1], lr[1]))
    a = T('tsm', lambda: (ops.tsmotion(s[0], t1), ops.tsmotion(s[1], t2)))
    o = T('smooth', lambda: nets[2].run_windows(a[0][0], a[1][0], a[0][1], a[1][1], F - 6, 7, 1, 1))
    acc = T('estimate', lambda: pipeline.estimate_meshes(nets, lr[0], lr[1]))
    plan = T('plan', lambda: pipeline.render_plan([acc['smooth_mesh1'], acc['smooth_mesh2']], H, W))
    fr = T('render', lambda: pipeline.render_frames([hr[0], hr[1]], [acc['smooth_mesh1'], acc['smooth_mesh2']]))
    print('canvas', fr[1], fr[2], flush=True)
