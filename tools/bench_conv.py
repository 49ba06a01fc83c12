"""Micro-benchmark of ss_conv_nhwc on the layer shapes of the pipeline.  python tools/bench_conv.py [reps]"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabstitch2_amd import ops
dev = torch.device('cuda:0')
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
only = sys.argv[2] if len(sys.argv) > 2 else None
# name, n, h, w, cin, cout, k, stride, pad
SHAPES = [
    ('conv1      ', 32, 360, 480, 4, 64, 7, 2, 3),
    ('layer1     ', 32, 90, 120, 64, 64, 3, 1, 1),
    ('layer2.0c1 ', 32, 90, 120, 64, 128, 3, 2, 1),
    ('layer2     ', 32, 45, 60, 128, 128, 3, 1, 1),
    ('layer3.0c1 ', 32, 45, 60, 128, 256, 3, 2, 1),
    ('layer3     ', 32, 23, 30, 256, 256, 3, 1, 1),
    ('reg 124->64', 16, 45, 60, 124, 64, 3, 1, 1),
    ('reg 64@45  ', 16, 45, 60, 64, 64, 3, 1, 1),
    ('reg 128@22 ', 16, 22, 30, 128, 128, 3, 1, 1),
    ('reg 128@11 ', 16, 11, 15, 128, 128, 3, 1, 1),
    ('reg 256@5  ', 16, 5, 7, 256, 256, 3, 1, 1),
]
torch.manual_seed(0)
for name, n, h, w, cin, cout, k, s, p in SHAPES:
    if only and only not in name:
        continue
    x = torch.randn(n, h, w, cin, device=dev)
    wt = torch.randn(cout, 1, k, k, cin, device=dev) * 0.05
    b = torch.randn(cout, device=dev)
    out = ops.conv(x, wt, b, stride=s, pad=(0, p, p), relu=True)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.conv(x, wt, b, stride=s, pad=(0, p, p), relu=True, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    m = out.numel() // cout
    fl = 2.0 * m * cout * k * k * cin
    print('%s M=%8d N=%4d K=%5d  %8.3f ms  %7.1f TF/s (padded-K)' % (name, m, cout, k * k * cin, ms, fl / ms / 1e9), flush=True)
if only == 'smooth' or not only:
    x = torch.randn(26, 7, 7, 9, 128, device=dev); wt = torch.randn(128, 5, 3, 3, 128, device=dev) * 0.02; b = torch.randn(128, device=dev)
    out = ops.conv(x, wt, b, stride=1, pad=(2, 1, 1), relu=True); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(reps):
        ops.conv(x, wt, b, stride=1, pad=(2, 1, 1), relu=True, out=out)
    e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * 26 * 441 * 128 * 45 * 128
    print('smooth conv3d M=%8d N= 128 K= 5760  %8.3f ms  %7.1f TF/s' % (26 * 441, ms, fl / ms / 1e9))
