"""Interleaved A/B of conv tile variants inside ONE process (boxes differ by >10 % between gpurun calls)."""
import sys, os, torch, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabstitch2_amd import ops, _hip
import _tuning
lib = _tuning.lib()
dev = torch.device('cuda:0')
SHAPES = {'conv1': (64, 360, 480, 4, 64, 7, 2, 3), 'layer1': (64, 90, 120, 64, 64, 3, 1, 1), 'layer2': (64, 45, 60, 128, 128, 3, 1, 1),
          'layer3': (64, 23, 30, 256, 256, 3, 1, 1), 'l2.0': (64, 90, 120, 64, 128, 3, 2, 1), 'reg124': (32, 45, 60, 124, 64, 3, 1, 1)}
variants = [int(v) for v in sys.argv[1].split(',')]
names = sys.argv[2].split(',') if len(sys.argv) > 2 else list(SHAPES)
rounds = 6
for name in names:
    n, h, w, cin, cout, k, s, p = SHAPES[name]
    x = torch.randn(n, h, w, cin, device=dev); wt = torch.randn(cout, 1, k, k, cin, device=dev) * 0.05; b = torch.randn(cout, device=dev)
    out = ops.conv(x, wt, b, stride=s, pad=(0, p, p), relu=True)
    m = out.numel() // cout; fl = 2.0 * m * cout * k * k * cin
    res = {v: [] for v in variants}
    for r in range(rounds):
        for v in variants:
            lib.ss_debug_set(0, v)
            for _ in range(3): ops.conv(x, wt, b, stride=s, pad=(0, p, p), relu=True, out=out)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
            for _ in range(20): ops.conv(x, wt, b, stride=s, pad=(0, p, p), relu=True, out=out)
            e1.record(); torch.cuda.synchronize(); res[v].append(e0.elapsed_time(e1) / 20)
    lib.ss_debug_set(0, 0)
    print('%-7s M=%7d N=%3d K=%4d ' % (name, m, cout, k * k * cin) + '  '.join('t%d: %.3f ms %5.1f TF' % (v, sorted(res[v])[len(res[v]) // 2], fl / sorted(res[v])[len(res[v]) // 2] / 1e9) for v in variants), flush=True)
