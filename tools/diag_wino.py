"""Per-workgroup phase timeline of one Winograd conv launch (s_memtime stamps, tuning build).
    python tools/diag_wino.py layer1,layer2,layer3"""
import sys, os, torch, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabstitch2_amd import ops, _hip
import _tuning
lib = _tuning.lib()
dev = torch.device('cuda:0')
SHAPES = {'layer1': (64, 90, 120, 64, 64), 'layer2': (64, 45, 60, 128, 128), 'layer3': (64, 23, 30, 256, 256)}
VARIANTS = {'64-channel blocks (2 WG/CU)': 0, '32-channel blocks (3 WG/CU)': 4096}
if len(sys.argv) > 2:
    VARIANTS = {'64-channel blocks (2 WG/CU)': 0}
for name, vname in ((a, b) for a in sys.argv[1].split(',') for b in VARIANTS):
    lib.ss_debug_set(5, VARIANTS[vname])
    lib.ss_debug_set(6, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    n, h, w, cin, cout = SHAPES[name]
    x = torch.randn(n, h, w, cin, device=dev); wt = torch.randn(cout, 1, 3, 3, cin, device=dev) * 0.05; b = torch.randn(cout, device=dev)
    out = ops.conv_winograd(x, wt, b, None, relu=True)
    res = torch.randn_like(out)
    nblk = 1 << 16
    dbg = torch.zeros((nblk, 10), dtype=torch.int64, device=dev)
    for _ in range(3): ops.conv_winograd(x, wt, b, res, relu=True, out=out)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.conv_winograd(x, wt, b, res, relu=True, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print('%s [%s]: %.1f us per launch, %.1f TF/s direct-equivalent' % (name, vname, ms * 1e3, 2.0 * n * h * w * cout * 9 * cin / ms / 1e9))
    lib.ss_debug_ptr(ctypes.c_void_p(dbg.data_ptr()))
    ops.conv_winograd(x, wt, b, res, relu=True, out=out)
    torch.cuda.synchronize()
    lib.ss_debug_ptr(None)
    d = dbg.cpu().numpy().astype(np.int64)
    d = d[d[:, 0] > 0]
    t0 = d[:, 0].min()
    span = d[:, 8].max() - t0
    med = lambda a: int(np.median(a))
    print('   %d blocks; per block median ticks (~shader clocks): setup %d | first raw wait + barrier %d | transform (chunk 0) %d | '
          'whole K loop %d | epilogue: residual issue + wait for the slowest wave %d, T stage + barrier %d, combine + store %d | total %d'
          % (len(d), med(d[:, 1] - d[:, 0]), med(d[:, 5] - d[:, 1]), med(d[:, 6] - d[:, 5]),
             med(d[:, 2] - d[:, 1]), med(d[:, 4] - d[:, 2]), med(d[:, 7] - d[:, 4]), med(d[:, 8] - d[:, 7]), med(d[:, 8] - d[:, 0])))
    hw = d[:, 9] & 0xFFFFFFFF; xcc = (d[:, 9] >> 32) & 0xF
    cu = ((hw >> 8) & 0xF) | (((hw >> 13) & 0x7) << 4) | (((hw >> 12) & 1) << 7) | (xcc << 8)
    ucu = np.unique(cu)
    c0 = ucu[len(ucu) // 2]
    sel = d[cu == c0]
    ts = np.linspace(sel[:, 0].min(), sel[:, 8].max(), 400)
    res_ = [np.sum((sel[:, 0] <= t) & (t < sel[:, 8])) for t in ts]
    inloop = [np.sum((sel[:, 1] <= t) & (t < sel[:, 2])) for t in ts]
    print('   distinct CUs %d; CU %d ran %d blocks; time-avg resident blocks %.2f, in K loop %.2f' % (len(ucu), c0, len(sel), np.mean(res_), np.mean(inloop)))
