import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from stabstitch2_amd import synth, pipeline
dev = torch.device('cuda:0'); torch.set_grad_enabled(False)
nets, _ = bench.build_nets(dev)
hr, lr = synth.make_clip_device(32, 720, 1280, 0, device=dev)
for i in range(3):
    pipeline.run_two_view(hr[0], hr[1], lr[0], lr[1], nets)
torch.cuda.synchronize()
for i in range(4):
    t0 = time.perf_counter()
    acc = pipeline.estimate_meshes(nets, lr[0], lr[1])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    out = pipeline.render_frames([hr[0], hr[1]], [acc['smooth_mesh1'], acc['smooth_mesh2']])
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print('estimate: enqueue %.2f ms, +sync %.2f ms | render: enqueue(incl bbox sync) %.2f ms, +sync %.2f ms' % ((t1 - t0) * 1e3, (t2 - t0) * 1e3, (t3 - t2) * 1e3, (t4 - t2) * 1e3))
