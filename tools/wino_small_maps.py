"""Winograd against the split-K implicit GEMM on the regressors' SMALL maps (tile-slot utilisation below the dispatch rule's
0.70): 11x15 (64 %), 5x7 / 6x8 (27-38 %) at the batch sizes of a 32-frame clip.      python tools/wino_small_maps.py"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabstitch2_amd import ops
dev = torch.device('cuda:0')
def t(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (n, h, w, cin, cout) in ((62, 11, 15, 128, 128), (64, 11, 15, 128, 128), (62, 11, 15, 64, 128), (64, 22, 30, 64, 128), (62, 5, 7, 256, 256),
                             (64, 5, 7, 128, 256), (32, 11, 15, 128, 128), (32, 5, 7, 128, 128), (62, 22, 30, 128, 128)):
    x = torch.randn(n, h, w, cin, device=dev); wt = torch.randn(cout, 1, 3, 3, cin, device=dev) * 0.05
    out = torch.empty(n, h, w, cout, device=dev)
    tw = t(lambda: ops.conv_winograd(x, wt, None, None, relu=True, out=out))
    ops.WINOGRAD = False
    ti = t(lambda: ops.conv(x, wt, None, None, relu=True, out=out))
    ops.WINOGRAD = True
    print('n %2d %2dx%2d %3d->%3d: winograd %6.1f us, igemm (+ split-K reduce) %6.1f us   dispatch rule says winograd: %d'
          % (n, h, w, cin, cout, tw, ti, ops._uses_winograd(1, 3, 3, 1, (0, 1, 1), cin, cout, h, w, n)))
