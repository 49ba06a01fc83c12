"""Latency-bound mesh-sized kernels of the path, per launch (HIP events): the homography decomposition (spatial_decompose,
spatial_meshes: two 8x8 fp64 DLT solves per frame pair), the TPS solve, the render's tile order.   python tools/bench_small_kernels.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabstitch2_amd import ops, pipeline
dev = torch.device('cuda:0')
torch.manual_seed(0)


def timed(fn, reps=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for n in (1, 32):
    off = torch.randn(n, 8, device=dev) * 20
    o2 = torch.randn(n, 63, 2, device=dev)
    print('n=%2d  spatial_decompose %.1f us   spatial_meshes %.1f us' % (
        n, timed(lambda: ops.spatial_decompose(off, 360, 480)), timed(lambda: ops.spatial_meshes(off, o2, o2, 360, 480))))
nr = pipeline.norm_rigid_mesh(720, 1280, dev)
for n in (1, 32):
    src = (nr.view(1, 1, 63, 2) + 0.03 * torch.randn(n, 2, 63, 2, device=dev)).contiguous()
    T = ops.tps_solve_shared(src.view(n * 2, 63, 2), nr).view(n, 2, 2, 66)
    print('n=%2d  tps_solve %.1f us   render_footprints (lattice + tile order) %.1f us' % (
        n, timed(lambda: ops.tps_solve_shared(src.view(n * 2, 63, 2), nr)), timed(lambda: ops.render_footprints(src, T, 720, 1280, 784, 1995))))
