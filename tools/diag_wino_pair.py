"""Where a persistent pair-kernel workgroup spends its time (s_memtime sums per phase, tuning build)."""
import sys, os, torch, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabstitch2_amd import ops
import _tuning
lib = _tuning.lib()
dev = torch.device('cuda:0')
SHAPES = {'layer1': (64, 90, 120, 64, 64), 'layer2': (64, 45, 60, 128, 128), 'layer3': (64, 23, 30, 256, 256)}
lib.ss_debug_set(7, 2); lib.ss_debug_set(6, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
for name in (sys.argv[1].split(',') if len(sys.argv) > 1 else SHAPES):
    n, h, w, cin, cout = SHAPES[name]
    x = torch.randn(n, h, w, cin, device=dev); wt = torch.randn(cout, 1, 3, 3, cin, device=dev) * 0.05; b = torch.randn(cout, device=dev)
    out = ops.conv_winograd(x, wt, b, None, relu=True); res = torch.randn_like(out)
    for _ in range(3): ops.conv_winograd(x, wt, b, res, relu=True, out=out)
    torch.cuda.synchronize()
    dbg = torch.zeros((4096, 10), dtype=torch.int64, device=dev)
    lib.ss_debug_ptr(ctypes.c_void_p(dbg.data_ptr()))
    ops.conv_winograd(x, wt, b, res, relu=True, out=out)
    torch.cuda.synchronize()
    lib.ss_debug_ptr(None)
    d = dbg.cpu().numpy(); d = d[d[:, 7] > 0]
    tot = d[:, 7] - d[:, 6]
    nchunk = (cin + 15) // 16
    tiles = ((h + 1) // 2 + 7) // 8 * (((w + 1) // 2 + 3) // 4)  # not exact for (4,8); only for the job count estimate
    print('%s: %d workgroups, span %d ticks; per workgroup median ticks: total %d = setup %d + chunk0 %d + other chunks %d (%d each) + T stage %d + combine/store %d'
          % (name, len(d), d[:, 7].max() - d[:, 6].min(), np.median(tot), *[np.median(d[:, i]) for i in range(3)],
             np.median(d[:, 2]) / max(1, (nchunk - 1)), np.median(d[:, 3]), np.median(d[:, 4])))
lib.ss_debug_set(7, 0)
