"""Does conv throughput depend on how the workgroup count divides by the resident capacity (256 CUs x 8)?"""
import sys, os, torch, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabstitch2_amd import ops, _hip
dev = torch.device('cuda:0')
for (n, h, w) in ((64, 48, 64), (64, 48, 65), (64, 48, 70), (64, 48, 75), (64, 48, 80), (64, 48, 85), (64, 32, 64), (64, 16, 64), (64, 8, 64), (64, 24, 64)):
    cin = cout = 128
    x = torch.randn(n, h, w, cin, device=dev); wt = torch.randn(cout, 1, 3, 3, cin, device=dev) * 0.05
    out = ops.conv(x, wt, None, stride=1, pad=(0, 1, 1), relu=True)
    m = n * h * w; fl = 2.0 * m * cout * 9 * cin
    ts = []
    for r in range(5):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
        for _ in range(20): ops.conv(x, wt, None, stride=1, pad=(0, 1, 1), relu=True, out=out)
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 20)
    t = sorted(ts)[2]
    blocks = (m // 64) * 2
    print('M=%7d blocks=%5d rounds(2048)=%.2f  %.3f ms %6.1f TF' % (m, blocks, blocks / 2048, t, fl / t / 1e9), flush=True)
