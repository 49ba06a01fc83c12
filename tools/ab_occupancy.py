"""Conv throughput vs resident workgroups per CU (extra dynamic LDS through ss_debug_set key 4)."""
import sys, os, torch, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabstitch2_amd import ops, _hip
import _tuning
lib = _tuning.lib()
dev = torch.device('cuda:0')
SHAPES = {'layer1': (64, 90, 120, 64, 64, 3, 1, 1), 'layer2': (64, 45, 60, 128, 128, 3, 1, 1), 'layer3': (64, 23, 30, 256, 256, 3, 1, 1)}
pads = [0, 4096, 12288, 20480, 32768, 57344]      # ~22 KB base: 7, 6, 4, 3, 2(3), 2 resident per CU
for name, (n, h, w, cin, cout, k, s, p) in SHAPES.items():
    x = torch.randn(n, h, w, cin, device=dev); wt = torch.randn(cout, 1, k, k, cin, device=dev) * 0.05; b = torch.randn(cout, device=dev)
    out = ops.conv(x, wt, b, stride=s, pad=(0, p, p), relu=True)
    m = out.numel() // cout; fl = 2.0 * m * cout * k * k * cin
    line = []
    for pad in pads:
        lib.ss_debug_set(4, pad)
        ts = []
        for r in range(4):
            for _ in range(3): ops.conv(x, wt, b, stride=s, pad=(0, p, p), relu=True, out=out)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
            for _ in range(20): ops.conv(x, wt, b, stride=s, pad=(0, p, p), relu=True, out=out)
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 20)
        t = sorted(ts)[1]
        line.append('+%dK: %.1f TF' % (pad // 1024, fl / t / 1e9))
    lib.ss_debug_set(4, 0)
    print(name, '  '.join(line), flush=True)
