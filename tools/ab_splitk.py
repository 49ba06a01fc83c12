import sys, os, torch, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabstitch2_amd import ops, _hip
import _tuning
lib = _tuning.lib()
dev = torch.device('cuda:0')
cases = {'smooth3d': ((26, 7, 7, 9, 128), (128, 5, 3, 3, 128), (2, 1, 1)), 'reg128@11x15x32': ((32, 11, 15, 128), (128, 1, 3, 3, 128), (0, 1, 1)),
         'reg256@5x7x32': ((32, 5, 7, 256), (256, 1, 3, 3, 256), (0, 1, 1)), 'reg128@22x30x32': ((32, 22, 30, 128), (128, 1, 3, 3, 128), (0, 1, 1)),
         'layer3': ((64, 23, 30, 256), (256, 1, 3, 3, 256), (0, 1, 1))}
targets = [int(v) for v in sys.argv[1].split(',')]
for name, (xs, ws, pad) in cases.items():
    x = torch.randn(*xs, device=dev); w = torch.randn(*ws, device=dev) * 0.02
    out = ops.conv(x, w, None, pad=pad, relu=True); res = {t: [] for t in targets}
    for r in range(6):
        for t in targets:
            lib.ss_debug_set(2, t)
            for _ in range(3): ops.conv(x, w, None, pad=pad, relu=True, out=out)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
            for _ in range(20): ops.conv(x, w, None, pad=pad, relu=True, out=out)
            e1.record(); torch.cuda.synchronize(); res[t].append(e0.elapsed_time(e1) / 20)
    m = out.numel() // ws[0]; fl = 2.0 * m * ws[0] * ws[1] * ws[2] * ws[3] * ws[4]
    print('%-18s M=%6d ' % (name, m) + '  '.join('s%d: %.1f us %5.1f TF' % (t, sorted(res[t])[3] * 1e3, fl / sorted(res[t])[3] / 1e9) for t in targets), flush=True)
lib.ss_debug_set(2, 512)
