"""Whole 720p 2-view clip with / without the first-round stagger of the fused stem kernel (tuning build, ss_debug_set(16, k):
the second resident workgroup of a CU sleeps k x 1024 clocks in the first round).      python tools/ab_stem_stagger.py"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from stabstitch2_amd import synth, pipeline
import _tuning
lib = _tuning.lib()
dev = torch.device('cuda:0'); torch.set_grad_enabled(False)
nets, _ = bench.build_nets(dev)
hr, lr = synth.make_clip_device(32, 720, 1280, seed=0, device=dev)
step = lambda: pipeline.run_two_view(hr[0], hr[1], lr[0], lr[1], nets)
for _ in range(5): step()
torch.cuda.synchronize()
for rnd in range(3):
    for k in (0, 16, 32, 48):
        lib.ss_debug_set(16, k)
        step(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        print('stagger %2dk clocks: %.3f ms per clip = %.1f frames/s' % (k, dt * 1e3, 32 / dt))
lib.ss_debug_set(16, 0)
