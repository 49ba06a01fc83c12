import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabstitch2_amd import ops
dev = torch.device('cuda:0'); torch.manual_seed(0)
f1 = torch.randn(32, 23, 30, 256, device=dev); f2 = torch.randn(32, 23, 30, 256, device=dev)
for _ in range(3): ops.ccl(f1, f2, 10.0, want_nchw=False, want_nhwc4=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): ops.ccl(f1, f2, 10.0, want_nchw=False, want_nhwc4=True)
e1.record(); torch.cuda.synchronize()
print('ccl (l2norm x2 + Gram + softmax), 32 pairs: %.1f us' % (e0.elapsed_time(e1) * 1e3 / 50))
