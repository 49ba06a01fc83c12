"""Do PCIe copies overlap with compute on this box?  H2D/D2H alone, together, and beside the 2-view pipeline."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench
from stabstitch2_amd import synth, pipeline

dev = torch.device('cuda:0')
torch.set_grad_enabled(False)
nets, _ = bench.build_nets(dev)
hr, lr = synth.make_clip_device(32, 720, 1280, seed=0, device=dev)
src = torch.empty((2, 32, 720, 1280, 3), dtype=torch.uint8).pin_memory()
dst_d = torch.empty((32, 730, 1414, 3), dtype=torch.uint8, device=dev)
dst_h = torch.empty((32, 730, 1414, 3), dtype=torch.uint8).pin_memory()
buf_d = torch.empty_like(src, device=dev)
up, down = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def h2d():
    with torch.cuda.stream(up):
        buf_d.copy_(src, non_blocking=True)


def d2h():
    with torch.cuda.stream(down):
        dst_h.copy_(dst_d, non_blocking=True)


def comp():
    pipeline.run_two_view(hr[0], hr[1], lr[0], lr[1], nets)


print('h2d %.1f MB: %.2f ms' % (src.numel() / 1e6, t(h2d)))
print('d2h %.1f MB: %.2f ms' % (dst_h.numel() / 1e6, t(d2h)))
print('h2d+d2h together: %.2f ms' % t(lambda: (h2d(), d2h())))
print('compute alone: %.2f ms' % t(comp))
print('compute + h2d: %.2f ms' % t(lambda: (h2d(), comp())))
print('compute + d2h: %.2f ms' % t(lambda: (d2h(), comp())))
print('compute + both: %.2f ms' % t(lambda: (h2d(), d2h(), comp())))
