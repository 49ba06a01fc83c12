"""In-process A/B of host-side switches in the single-stream streaming mode (graph captured per stitcher)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stabstitch2_amd  # noqa
import torch
import bench
from stabstitch2_amd import synth, ops, layers
from stabstitch2_amd.online import OnlineStitcher
dev = torch.device('cuda:0'); torch.set_grad_enabled(False)
nets, _ = bench.build_nets(dev)
hr, lr = synth.make_clip_device(2, 720, 1280, seed=0, device=dev)
a = [hr[0][:1].contiguous(), hr[1][:1].contiguous(), lr[0][:1].contiguous(), lr[1][:1].contiguous()]
def run():
    one = OnlineStitcher(nets, 720, 1280)
    for _ in range(12): one.push(*a)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): one.push(*a)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 200 * 1e3
for rounds in range(3):
    for pool, quad in ((True, True), (False, True), (True, False), (False, False)):
        ops.POOL_SPLITK = pool; layers.QUAD = quad
        print('pool-in-reduce %-5s quad %-5s: %.3f ms per push' % (pool, quad, run()))
