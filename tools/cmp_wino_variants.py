"""Winograd kernel schedules against each other (tuning build, ss_debug_set key 7: 0 stream = dispatched, 1 alternating,
2 pair kernel): outputs must be bit-identical (same accumulation order), plus timing of both.
    python tools/cmp_wino_variants.py [shapes]          WINO_VARIANT=2 for the pair kernel; "single" = variant 1"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabstitch2_amd import ops
import _tuning
lib = _tuning.lib()
dev = torch.device('cuda:0')
SHAPES = {'layer1': (64, 90, 120, 64, 64), 'layer2': (64, 45, 60, 128, 128), 'layer3': (64, 23, 30, 256, 256),
          'odd': (3, 37, 53, 36, 64), 'small': (1, 8, 8, 32, 128), 'reg': (32, 45, 60, 160, 64)}
names = sys.argv[1].split(',') if len(sys.argv) > 1 else list(SHAPES)
for name in names:
    n, h, w, cin, cout = SHAPES[name]
    torch.manual_seed(1)
    x = torch.randn(n, h, w, cin, device=dev); wt = torch.randn(cout, 1, 3, 3, cin, device=dev) * 0.05; b = torch.randn(cout, device=dev)
    res = torch.randn(n, h, w, cout, device=dev)
    outs, times = [], []
    for variant in (1, int(os.environ.get("WINO_VARIANT", "0"))):
        lib.ss_debug_set(7, variant)
        for r in (None, res):
            out = ops.conv_winograd(x, wt, b, r, relu=r is not None)
            outs.append(out.clone())
        for _ in range(3): ops.conv_winograd(x, wt, b, res, relu=True, out=out)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.conv_winograd(x, wt, b, res, relu=True, out=out)
        e1.record(); torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) / 10 * 1e3)
    lib.ss_debug_set(7, 0)
    d0 = (outs[0] - outs[2]).abs().max().item(); d1 = (outs[1] - outs[3]).abs().max().item()
    gf = 2.0 * n * h * w * cout * 9 * cin / 1e9
    print('%-7s max|single - variant| = %.3g / %.3g (with residual+relu); single %.1f us (%.0f TF/s eq), variant %.1f us (%.0f TF/s eq)'
          % (name, d0, d1, times[0], gf / times[0] * 1e3, times[1], gf / times[1] * 1e3))
