import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabstitch2_amd import ops
dev = torch.device('cuda:0')
for (n, h, w, cin, cout) in [(16, 64, 64, 128, 128), (32, 64, 64, 128, 128), (64, 64, 64, 128, 128), (16, 64, 64, 256, 256), (32, 64, 64, 256, 256), (32, 64, 64, 64, 64), (64, 64, 64, 64, 64)]:
    x = torch.randn(n, h, w, cin, device=dev); wt = torch.randn(cout, 1, 3, 3, cin, device=dev) * 0.05
    out = ops.conv(x, wt, None, stride=1, pad=(0, 1, 1)); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(20): ops.conv(x, wt, None, stride=1, pad=(0, 1, 1), out=out)
    e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / 20
    m = n * h * w
    print('M=%7d N=%3d K=%4d blocks128=%5d  %.3f ms %6.1f TF/s' % (m, cout, 9 * cin, (m // 128) * max(1, cout // 128), ms, 2.0 * m * cout * 9 * cin / ms / 1e9), flush=True)
