"""fp32-MFMA Winograd kernel against the bf16x9 variant (three exact bf16 slices per operand, nine slice products on the
bf16 pipe): deviation of both from an fp64 convolution, and timing.   python tools/cmp_wino_math.py"""
import sys, os, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabstitch2_amd import ops
dev = torch.device('cuda:0')
SHAPES = {'layer1': (64, 90, 120, 64, 64), 'layer2': (64, 45, 60, 128, 128), 'layer3': (64, 23, 30, 256, 256), 'reg': (32, 45, 60, 160, 64),
          'odd': (3, 37, 53, 36, 64)}
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, (n, h, w, cin, cout) in SHAPES.items():
    torch.manual_seed(3)
    x = torch.randn(n, h, w, cin, device=dev).relu(); wt = torch.randn(cout, 1, 3, 3, cin, device=dev) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn(cout, device=dev); res = torch.randn(n, h, w, cout, device=dev)
    nn = min(n, 4)
    ref = F.conv2d(x[:nn].permute(0, 3, 1, 2).double(), wt[:, 0].permute(0, 3, 1, 2).double(), b.double(), padding=1).permute(0, 2, 3, 1) + res[:nn].double()
    out = torch.empty(n, h, w, cout, device=dev)
    r = {}
    for math in ('f32', 'bf16x9'):
        ops.WINO_MATH = math
        o = ops.conv_winograd(x, wt, b, res, relu=False)
        err = (o[:nn].double() - ref).abs()
        r[math] = (err.max().item(), err.pow(2).mean().sqrt().item(), t(lambda: ops.conv_winograd(x, wt, b, res, relu=True, out=out)), o)
    ops.WINO_MATH = 'f32'
    print('%-7s |out| max %.2f; vs fp64: f32 max %.3g rms %.3g | bf16x9 max %.3g rms %.3g | f32 vs bf16x9 max %.3g;  %.1f us -> %.1f us (x%.2f)'
          % (name, ref.abs().max().item(), r['f32'][0], r['f32'][1], r['bf16x9'][0], r['bf16x9'][1],
             (r['f32'][3] - r['bf16x9'][3]).abs().max().item(), r['f32'][2], r['bf16x9'][2], r['f32'][2] / r['bf16x9'][2]))
