"""Streaming-sized (2-3 images) trunk conv launches: the dispatched F(2x2,3x3) kernel (64-channel blocks, 2 workgroups / CU) against
the tuning build's 32-channel blocks (3 workgroups / CU), the forced F(4x4,3x3) kernel and the implicit GEMM.
    python tools/stream_conv_variants.py"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabstitch2_amd import ops, _hip
import _tuning
lib = _tuning.lib()
dev = torch.device('cuda:0')
SHAPES = {'layer1': (90, 120, 64, 64), 'layer2': (45, 60, 128, 128), 'layer3': (23, 30, 256, 256)}
def timeit(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for n in (2, 3, 4):
    for name, (h, w, cin, cout) in SHAPES.items():
        x = torch.randn(n, h, w, cin, device=dev); wt = torch.randn(cout, 1, 3, 3, cin, device=dev) * 0.05; b = torch.randn(cout, device=dev)
        res = torch.randn(n, h, w, cout, device=dev)
        out = torch.empty_like(res)
        r = {}
        lib.ss_debug_set(5, 0)
        r['wino22 64ch'] = timeit(lambda: ops.conv_winograd(x, wt, b, res, relu=True, out=out))
        lib.ss_debug_set(5, 4096)
        r['wino22 32ch'] = timeit(lambda: ops.conv_winograd(x, wt, b, res, relu=True, out=out))
        lib.ss_debug_set(5, 0)
        try:
            r['wino43'] = timeit(lambda: ops.conv_winograd43(x, wt, b, res, True))
        except Exception as e:
            r['wino43'] = float('nan')
        print('n=%d %-7s' % (n, name), '  '.join('%s %.1f us' % kv for kv in r.items()), flush=True)
