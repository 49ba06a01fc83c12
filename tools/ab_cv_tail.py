"""A/B of the cost volume's 16-byte tail reads: tools/lib_old_cv.so (before) vs the product library, alternating.  python tools/ab_cv_tail.py"""
import sys, os, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabstitch2_amd import ops, _hip
dev = torch.device('cuda:0')
new = _hip.lib()
old = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'lib_old_cv.so'))
for name, (res, args) in _hip.SIGNATURES.items():
    if hasattr(old, name):
        fn = getattr(old, name); fn.restype = res; fn.argtypes = args
def t(n, r, bidir):
    a = torch.randn(n, 45, 60, 128, device=dev); b = torch.randn(n, 45, 60, 128, device=dev)
    f = (lambda: ops.cost_volume_bidir(a, b, r)) if bidir else (lambda: ops.cost_volume(a, b, r))
    y = f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 50 * 1e3, y
for n, r, bidir in ((32, 5, True), (64, 3, False)):
    for rnd in range(3):
        _hip._lib = old; a, ya = t(n, r, bidir)
        _hip._lib = new; b, yb = t(n, r, bidir)
        print('n=%d r=%d bidir=%d: old %.1f us  new %.1f us' % (n, r, bidir, a, b), flush=True)
