"""Per-workgroup phase clocks of the fused stem kernel (tuning build, s_memtime stamps).   python tools/diag_stem.py"""
import sys, os, torch, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabstitch2_amd import ops
import _tuning
lib = _tuning.lib()
dev = torch.device('cuda:0')
n = 64
x = torch.randn(n, 3, 360, 480, device=dev)
w = torch.zeros(128, 7, 24, device=dev); w[:, :, :21] = torch.randn(128, 7, 21, device=dev) * 0.1
b = torch.randn(128, device=dev)
buf = ops.stem_input(x)
for pad, stag in ((0, 0), (0, 16), (0, 32), (0, 48), (0, 64), (40960, 0)):
  lib.ss_debug_set(20, pad); lib.ss_debug_set(16, stag)
  print('--- %s, first-round stagger %dk clocks' % ('two workgroups per CU' if pad == 0 else 'ONE workgroup per CU', stag))
  for _ in range(3): ops.stem_pool(buf, w, b)
  torch.cuda.synchronize()
  e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(20): ops.stem_pool(buf, w, b)
  e1.record(); torch.cuda.synchronize()
  print('   %.1f us per launch' % (e0.elapsed_time(e1) / 20 * 1e3))
  dbg = torch.zeros((1 << 15, 10), dtype=torch.int64, device=dev)
  lib.ss_debug_ptr(ctypes.c_void_p(dbg.data_ptr()))
  ops.stem_pool(buf, w, b)
  torch.cuda.synchronize()
  lib.ss_debug_ptr(None)
  d = dbg.cpu().numpy().astype(np.int64); d = d[d[:, 0] > 0]
  med = lambda a: int(np.median(a))
  k = med(d[:, 2] - d[:, 1])
  print('   epilogue: wait for the slowest wave %d | stage half 0 + barrier %d | pool + store half 0 %d | barrier + stage half 1 + barrier %d | pool + store half 1 %d'
      % (med(d[:, 3] - d[:, 2]), med(d[:, 4] - d[:, 3]), med(d[:, 5] - d[:, 4]), med(d[:, 6] - d[:, 5]), med(d[:, 8] - d[:, 6])))
  print('%d workgroups; median clocks: patch staging %d | K loop %d (672 MFMAs per wave = 43008 of the pipe: %.0f %% of its pace per workgroup) | '
      'epilogue (2 x stage + pool + store) %d | total %d; launch span %d clocks'
      % (len(d), med(d[:, 1] - d[:, 0]), k, 100.0 * 43008 / k, med(d[:, 8] - d[:, 2]), med(d[:, 8] - d[:, 0]), d[:, 8].max() - d[:, 0].min()))

lib.ss_debug_set(20, 0); lib.ss_debug_set(16, 0)
