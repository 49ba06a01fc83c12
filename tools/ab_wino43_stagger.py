"""Experiment: does spreading the FIRST round of conv_wino43_kernel's workgroups over time (so that later rounds' prologues / epilogues
do not hit HBM as one burst of all 256 CUs) shorten the launch?   python tools/ab_wino43_stagger.py   (tuning build, ss_debug_set(17, clocks per CU slot))"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _tuning
lib = _tuning.lib()
from stabstitch2_amd import ops
dev = torch.device('cuda:0')
SHAPES = {'layer1': (64, 90, 120, 64, 64), 'layer2': (64, 45, 60, 128, 128), 'layer3': (64, 23, 30, 256, 256)}
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for name, (n, h, w, cin, cout) in SHAPES.items():
    x = torch.randn(n, h, w, cin, device=dev); wt = torch.randn(cout, 1, 3, 3, cin, device=dev) * 0.05; b = torch.randn(cout, device=dev)
    out = ops.conv_winograd43(x, wt, b, None, relu=True); res = torch.randn_like(out)
    row = []
    for st in ([0, -1, 0, -1] if '--prio' in sys.argv else [0, 256, 512, 1024, 2048, 0]):
        lib.ss_debug_set(17, st)
        row.append('%d: %.1f / %.1f' % (st, t(lambda: ops.conv_winograd43(x, wt, b, None, relu=True, out=out)), t(lambda: ops.conv_winograd43(x, wt, b, res, relu=True, out=out))))
    lib.ss_debug_set(17, 0)
    print(name, '(stagger clocks per slot: us without / with residual)  ', '   '.join(row), flush=True)
