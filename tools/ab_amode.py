"""Interleaved in-process A/B of the conv address modes (ss_debug_set key 3: 0 = auto (aligned fast path where it
applies), 2 = LDS tap table without the fast path, 1 = arithmetic)."""
import sys, os, torch, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabstitch2_amd import ops, _hip
import _tuning
lib = _tuning.lib()
dev = torch.device('cuda:0')
SHAPES = {'conv1': (64, 1, 360, 480, 4, 64, 1, 7, 2, 3), 'layer1': (64, 1, 90, 120, 64, 64, 1, 3, 1, 1),
          'layer2': (64, 1, 45, 60, 128, 128, 1, 3, 1, 1), 'layer3': (64, 1, 23, 30, 256, 256, 1, 3, 1, 1),
          'l2.0': (64, 1, 90, 120, 64, 128, 1, 3, 2, 1), 'reg124': (32, 1, 45, 60, 124, 64, 1, 3, 1, 1),
          'ds1x1': (64, 1, 90, 120, 64, 128, 1, 1, 2, 0), 'smooth3d': (26, 7, 7, 9, 128, 128, 5, 3, 1, 1),
          'reg5x7': (64, 1, 5, 7, 256, 256, 1, 3, 1, 1)}
names = sys.argv[1].split(',') if len(sys.argv) > 1 else list(SHAPES)
rounds = 6
for name in names:
    n, t, h, w, cin, cout, kt, k, s, p = SHAPES[name]
    x = torch.randn(n, t, h, w, cin, device=dev) if t > 1 else torch.randn(n, h, w, cin, device=dev)
    wt = torch.randn(cout, kt, k, k, cin, device=dev) * 0.05; b = torch.randn(cout, device=dev)
    pad = (kt // 2, p, p)
    outs = {}
    res = {0: [], 2: [], 1: []}
    for r in range(rounds):
        for v in (0, 2, 1):
            lib.ss_debug_set(3, v)
            for _ in range(3): out = ops.conv(x, wt, b, stride=s, pad=pad, relu=True)
            outs[v] = out
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
            for _ in range(20): ops.conv(x, wt, b, stride=s, pad=pad, relu=True, out=out)
            e1.record(); torch.cuda.synchronize(); res[v].append(e0.elapsed_time(e1) / 20)
    lib.ss_debug_set(3, 0)
    m = outs[0].numel() // cout; fl = 2.0 * m * cout * kt * k * k * cin
    med = {v: sorted(res[v])[len(res[v]) // 2] for v in res}
    print('%-8s M=%7d N=%3d K=%4d  auto: %.3f ms %5.1f TF   table: %.3f ms %5.1f TF   arith: %.3f ms %5.1f TF   equal=%s' % (
        name, m, cout, kt * k * k * cin, med[0], fl / med[0] / 1e9, med[2], fl / med[2] / 1e9, med[1], fl / med[1] / 1e9,
        bool(torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]))), flush=True)
