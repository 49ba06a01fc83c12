import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stabstitch2_amd  # noqa
import torch
import bench
from stabstitch2_amd import synth, pipeline
dev = torch.device('cuda:0'); torch.set_grad_enabled(False)
nets, _ = bench.build_nets(dev)
hr, lr = synth.make_clip_device(32, 720, 1280, seed=0, device=dev)
run = lambda: pipeline.run_two_view(hr[0], hr[1], lr[0], lr[1], nets)
ref = run()
for rounds in range(3):
    for ov in (False, True):
        pipeline.QUAD_OVERLAP = ov
        for _ in range(3): o = run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): o = run()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        print('overlap %s: %.3f ms/clip %.0f fps  equal %s' % (ov, dt * 1e3, 32 / dt, torch.equal(o[0], ref[0])))
