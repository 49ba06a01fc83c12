"""How well does ONE Winograd workgroup per CU feed the matrix pipe?  (tuning build: ss_debug_set(20, bytes) adds dynamic LDS so
that only one 64 KB workgroup fits a CU; per-workgroup s_memtime stamps give the K loop's ticks per 64-MFMA chunk: 4096 = the
pipe's own pace.)      python tools/exp_wino_alone.py"""
import sys, os, torch, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabstitch2_amd import ops
import _tuning
lib = _tuning.lib()
dev = torch.device('cuda:0')
SHAPES = {'layer1': (64, 90, 120, 64, 64), 'layer2': (64, 45, 60, 128, 128), 'layer3': (64, 23, 30, 256, 256)}
for name, (n, h, w, cin, cout) in SHAPES.items():
    x = torch.randn(n, h, w, cin, device=dev); wt = torch.randn(cout, 1, 3, 3, cin, device=dev) * 0.05; b = torch.randn(cout, device=dev)
    out = ops.conv_winograd(x, wt, b, None, relu=True)
    res = torch.randn_like(out)
    for var, pad, what in ((0, 0, 'two workgroups per CU'), (0, 40960, 'ONE workgroup per CU'), (6, 0, 'VAR4 two per CU'), (6, 40960, 'VAR4 ONE per CU')):
        lib.ss_debug_set(20, pad); lib.ss_debug_set(7, var)
        ref = None if var == 0 else keep
        for _ in range(3): ops.conv_winograd(x, wt, b, res, relu=True, out=out)
        torch.cuda.synchronize()
        if var == 0: keep = out.clone()
        elif var != 6: assert torch.equal(out, keep), 'variant differs'
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.conv_winograd(x, wt, b, res, relu=True, out=out)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        dbg = torch.zeros((1 << 16, 10), dtype=torch.int64, device=dev)
        lib.ss_debug_ptr(ctypes.c_void_p(dbg.data_ptr()))
        ops.conv_winograd(x, wt, b, res, relu=True, out=out)
        torch.cuda.synchronize()
        lib.ss_debug_ptr(None)
        d = dbg.cpu().numpy().astype(np.int64); d = d[d[:, 0] > 0]
        med = lambda a: int(np.median(a))
        nch = (cin + 15) // 16
        print('%s [%s]: %.1f us; per workgroup median ticks: prologue %d | K loop %d = %d per chunk (pipe pace 4096: %.0f %%) | '
              'epilogue %d | total %d' % (name, what, us, med(d[:, 1] - d[:, 0]), med(d[:, 2] - d[:, 1]), med(d[:, 2] - d[:, 1]) // nch,
                                         100.0 * 4096 * nch / med(d[:, 2] - d[:, 1]), med(d[:, 8] - d[:, 2]), med(d[:, 8] - d[:, 0])))
    lib.ss_debug_set(20, 0); lib.ss_debug_set(7, 0)
