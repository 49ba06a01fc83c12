"""Which part of the Winograd stream keeps a LONE workgroup per CU from the matrix pipe's pace?  Compile-time ablations of the
stream kernel (tuning build, ss_debug_set(7, 100 + mask): 1 no filter loads, 2 no raw-patch loads, 4 no LDS staging writes,
8 no LDS reads / transform, 16 no barrier; results are wrong, timing only), one and two workgroups per CU.
    python tools/exp_wino_ablate.py"""
import sys, os, torch, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabstitch2_amd import ops
import _tuning
lib = _tuning.lib()
dev = torch.device('cuda:0')
SHAPES = {'layer1': (64, 90, 120, 64, 64), 'layer2': (64, 45, 60, 128, 128)}
NAMES = {0: 'full stream', 1: 'no filter loads', 2: 'no raw loads', 4: 'no LDS writes', 8: 'no LDS reads/transform', 16: 'no barrier',
         3: 'no global loads at all', 12: 'no LDS traffic', 15: 'MFMAs + barrier only', 31: 'MFMAs only'}
for name, (n, h, w, cin, cout) in SHAPES.items():
    x = torch.randn(n, h, w, cin, device=dev); wt = torch.randn(cout, 1, 3, 3, cin, device=dev) * 0.05; b = torch.randn(cout, device=dev)
    out = ops.conv_winograd(x, wt, b, None, relu=True)
    res = torch.randn_like(out)
    nch = (cin + 15) // 16
    for pad, what in ((40960, 'ONE per CU'), (0, 'two per CU')):
        for mask in (0, 1, 2, 4, 8, 16, 3, 12, 15, 31):
            lib.ss_debug_set(20, pad); lib.ss_debug_set(7, 100 + mask if mask else 0)
            for _ in range(3): ops.conv_winograd(x, wt, b, res, relu=True, out=out)
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): ops.conv_winograd(x, wt, b, res, relu=True, out=out)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 10 * 1e3
            dbg = torch.zeros((1 << 16, 10), dtype=torch.int64, device=dev)
            lib.ss_debug_ptr(ctypes.c_void_p(dbg.data_ptr()))
            ops.conv_winograd(x, wt, b, res, relu=True, out=out)
            torch.cuda.synchronize()
            lib.ss_debug_ptr(None)
            d = dbg.cpu().numpy().astype(np.int64); d = d[d[:, 0] > 0]
            k = int(np.median(d[:, 2] - d[:, 1]))
            print('%s [%s] %-26s %7.1f us  K loop %6d ticks = %5d per chunk (%3.0f %% of the pipe pace for ONE wave per SIMD)  total %6d'
                  % (name, what, NAMES[mask], us, k, k // nch, 100.0 * 4096 * nch / k, int(np.median(d[:, 8] - d[:, 0]))))
lib.ss_debug_set(20, 0); lib.ss_debug_set(7, 0)
