"""Round-3 Winograd experiments (tuning build): first-round stagger of a CU's second workgroup, prologue / epilogue / K-loop
wave priorities.  Interleaved A/B in one process (boxes differ by several % between calls), 5 rounds x 10 launches each.
    python tools/exp_wino_r3.py [shapes]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabstitch2_amd import ops
import _tuning
lib = _tuning.lib()
dev = torch.device('cuda:0')
SHAPES = {'layer1': (64, 90, 120, 64, 64), 'layer2': (64, 45, 60, 128, 128), 'layer3': (64, 23, 30, 256, 256),
          'reg': (64, 45, 60, 124, 64)}
CONFIGS = [('base', {})] + [('stagger%d' % k, {0: k}) for k in (8, 16)] + \
          [('eprio1', {1: 2}), ('pprio1', {2: 2}), ('e1p1', {1: 2, 2: 2}), ('e1p1s8', {1: 2, 2: 2, 0: 8}), ('e1p1s16', {1: 2, 2: 2, 0: 16}),
           ('e0p1', {1: 1, 2: 2}), ('e1p0', {1: 2, 2: 1}), ('e2p2', {1: 3, 2: 3}), ('base2', {})]
if os.environ.get('WINO_CONFIGS') == 'first':
    CONFIGS = [('base', {})] + [('stagger%d' % k, {0: k}) for k in (8, 16, 32, 64)] + \
              [('eprio%d' % (v - 1), {1: v}) for v in (1, 2, 3)] + [('pprio%d' % (v - 1), {2: v}) for v in (1, 2, 3)] + \
              [('kprio1', {3: 2}), ('kprio1+eprio2', {3: 2, 1: 3}), ('stagger16+eprio1', {0: 16, 1: 2})]
names = sys.argv[1].split(',') if len(sys.argv) > 1 else list(SHAPES)
for name in names:
    n, h, w, cin, cout = SHAPES[name]
    torch.manual_seed(1)
    x = torch.randn(n, h, w, cin, device=dev); wt = torch.randn(cout, 1, 3, 3, cin, device=dev) * 0.05; b = torch.randn(cout, device=dev)
    res = torch.randn(n, h, w, cout, device=dev)
    out = ops.conv_winograd(x, wt, b, res, relu=True)
    ref = out.clone()
    tot = {c[0]: 0.0 for c in CONFIGS}
    for rnd in range(5):
        for cname, knobs in CONFIGS:
            for k in range(4): lib.ss_debug_set(16 + k, knobs.get(k, 0))
            ops.conv_winograd(x, wt, b, res, relu=True, out=out)
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): ops.conv_winograd(x, wt, b, res, relu=True, out=out)
            e1.record(); torch.cuda.synchronize()
            tot[cname] += e0.elapsed_time(e1) / 10 * 1e3
            assert torch.equal(out, ref), cname
    for k in range(4): lib.ss_debug_set(16 + k, 0)
    gf = 2.0 * n * h * w * cout * 9 * cin / 1e9
    print('%-7s ' % name + '  '.join('%s %.1f' % (c, tot[c] / 5) for c, _ in CONFIGS))
    print('        base %.1f us = %.0f TF/s direct-equivalent, %.1f executed' % (tot['base'] / 5, gf / (tot['base'] / 5) * 1e3, gf * 16 / 36 / (tot['base'] / 5) * 1e3))
