"""Split-K target of the implicit GEMM on streaming-sized twin-trunk launches (tuning build, ss_debug_set(2, target)):
    python tools/ab_splitk_stream.py"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabstitch2_amd import ops
import _tuning
lib = _tuning.lib()
dev = torch.device('cuda:0')
def timeit(fn, reps=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
# (name, groups, images, h, w, cin, cout, stride)
CASES = [('layer2 first conv, 3 views (chain)', 2, 3, 90, 120, 64, 128, 2), ('layer2 first conv, 2 views', 2, 2, 90, 120, 64, 128, 2),
         ('layer2 first conv, 4 views', 2, 4, 90, 120, 64, 128, 2),
         ('layer3 first conv, 3 views', 2, 3, 45, 60, 128, 256, 2), ('layer3 first conv, 2 views', 2, 2, 45, 60, 128, 256, 2),
         ('layer3 3x3, 3 views', 2, 3, 23, 30, 256, 256, 1), ('layer3 3x3, 2 views', 2, 2, 23, 30, 256, 256, 1)]
for name, g, n, h, w, cin, cout, stride in CASES:
    x = torch.randn(g, n, h, w, cin, device=dev); wt = torch.randn(g, cout, 1, 3, 3, cin, device=dev) * 0.05; b = torch.randn(g, cout, device=dev)
    r = []
    for target in (512, 384, 256, 128, 1):
        lib.ss_debug_set(2, target)
        r.append('%d: %.1f us' % (target, timeit(lambda: ops.conv_grouped(x, wt, b, None, stride=stride, pad=(0, 1, 1), relu=True))))
    lib.ss_debug_set(2, 512)
    print('%-40s' % name, '   '.join(r), flush=True)
