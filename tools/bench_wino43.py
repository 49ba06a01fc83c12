"""F(4x4,3x3) (csrc/wino43.hip) against F(2x2,3x3) (csrc/wino.hip) and torch's direct convolution: error and time per launch
on the trunk's stride-1 3x3 layers.      python tools/bench_wino43.py [--images 64]"""
import argparse, os, sys, time
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stabstitch2_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument('--images', type=int, default=64)
ap.add_argument('--iters', type=int, default=20)
args = ap.parse_args()
dev = torch.device('cuda:0')
torch.manual_seed(0)


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for (c, h, w, n) in [(128, 45, 60, args.images), (64, 90, 120, args.images), (256, 23, 30, args.images), (128, 45, 60, 3), (64, 13, 17, 2)]:
    x = torch.randn(n, h, w, c, device=dev)
    wgt = torch.randn(c, 1, 3, 3, c, device=dev) * (1.0 / (9 * c)) ** 0.5
    bias = torch.randn(c, device=dev) * 0.1
    res = torch.randn(n, h, w, c, device=dev)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), wgt[:, 0].permute(0, 3, 1, 2).double(), bias.double(), padding=1)
    ref = torch.relu(ref + res.permute(0, 3, 1, 2).double()).permute(0, 2, 3, 1)
    y43 = ops.conv_winograd43(x, wgt, bias, res, True)
    y23 = ops.conv_winograd(x, wgt, bias, res, True)
    torch.cuda.synchronize()
    e43 = (y43.double() - ref).abs().max().item()
    e23 = (y23.double() - ref).abs().max().item()
    t43 = timeit(lambda: ops.conv_winograd43(x, wgt, bias, res, True), args.iters)
    t23 = timeit(lambda: ops.conv_winograd(x, wgt, bias, res, True), args.iters)
    t43n = timeit(lambda: ops.conv_winograd43(x, wgt, bias, None, True), args.iters)
    t23n = timeit(lambda: ops.conv_winograd(x, wgt, bias, None, True), args.iters)
    fl = 2.0 * n * h * w * c * c * 9
    print('%3d->%3d @%dx%d x%d | max err F43 %.2e F23 %.2e (|y| max %.2f) | res: F43 %.1f us (%.0f TF/s direct-eq) F23 %.1f us | no res: F43 %.1f F23 %.1f'
          % (c, c, h, w, n, e43, e23, ref.abs().max().item(), t43, fl / t43 / 1e6, t23, t43n, t23n), flush=True)
