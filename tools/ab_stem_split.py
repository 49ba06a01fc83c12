"""A/B: stem_pool_kernel (both 32-channel halves per workgroup, 2 workgroups per CU) vs stem_pool_kernel_half (one half per workgroup,
4 per CU).  64 images of 360x480, two filter banks; bit-identity + time.   python tools/ab_stem_split.py   (tuning build, ss_debug_set(18, x))"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _tuning
lib = _tuning.lib()
from stabstitch2_amd import ops
dev = torch.device('cuda:0')
torch.manual_seed(0)
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for n, h, wd in ((64, 360, 480), (3, 90, 130), (2, 47, 61)):
    x = torch.randn(n, 3, h, wd, device=dev)
    w = torch.zeros(128, 7, 24, device=dev); w[:, :, :21] = torch.randn(128, 7, 21, device=dev) * 0.1
    b = torch.randn(128, device=dev)
    buf = ops.stem_input(x)
    lib.ss_debug_set(18, 2); y0 = ops.stem_pool(buf, w, b).clone()
    lib.ss_debug_set(18, 0); y1 = ops.stem_pool(buf, w, b).clone()
    torch.cuda.synchronize()
    print('%dx%dx%d: split == joint bit for bit: %s (max |diff| %.3e)' % (n, h, wd, bool(torch.equal(y0, y1)), float((y0 - y1).abs().max())), flush=True)
    if n == 64:
        for rnd in range(3):
            lib.ss_debug_set(18, 2); a = t(lambda: ops.stem_pool(buf, w, b))
            lib.ss_debug_set(18, 0); c = t(lambda: ops.stem_pool(buf, w, b))
            print('  joint %.1f us   split %.1f us' % (a, c), flush=True)
lib.ss_debug_set(18, 0)
