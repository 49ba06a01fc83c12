"""Whole 720p clip (32 pairs, the headline workload) under different split-K targets of the implicit GEMM (tuning build, key 2).
    python tools/ab_splitk_clip.py 256,384,512,640,768,1024"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from stabstitch2_amd import synth, pipeline
import _tuning
lib = _tuning.lib()
dev = torch.device('cuda:0'); torch.set_grad_enabled(False)
nets, _ = bench.build_nets(dev)
hr, lr = synth.make_clip_device(32, 720, 1280, seed=0, device=dev)
targets = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '256,384,512,640,768,1024').split(',')]
def clip():
    return pipeline.run_two_view(hr[0], hr[1], lr[0], lr[1], nets)
res = {t: [] for t in targets}
for rep in range(4):
    for t in targets:
        lib.ss_debug_set(2, t)
        for _ in range(2): clip()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8): clip()
        torch.cuda.synchronize()
        res[t].append((time.perf_counter() - t0) / 8 * 1e3)
lib.ss_debug_set(2, 512)
for t in targets:
    v = sorted(res[t])
    print('split target %4d: %.3f ms per clip (min %.3f)  %.0f frames/s' % (t, v[len(v) // 2], v[0], 32e3 / v[len(v) // 2]))
