import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabstitch2_amd import ops
dev = torch.device('cuda:0'); torch.manual_seed(0)
def timed(fn, reps=100):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for g, m, k, n in ((4, 32, 1536, 1024), (4, 32, 1024, 512), (4, 32, 512, 126), (4, 1, 1536, 1024)):
    x = torch.randn(g, m, k, device=dev); w = torch.randn(g, n, k, device=dev) * 0.02; b = torch.randn(g, n, device=dev)
    t = timed(lambda: ops.linear_grouped(x, w, b, relu=True))
    ref = torch.relu(torch.einsum('gmk,gnk->gmn', x.double(), w.double()) + b.double()[:, None]).float()
    print('linear_grouped g=%d m=%d k=%d n=%d: %.1f us  max err %.1e' % (g, m, k, n, t, float((ops.linear_grouped(x, w, b, relu=True) - ref).abs().max())))
x = torch.randn(32, 768, device=dev); w = torch.randn(512, 768, device=dev) * 0.02
print('linear m=32 k=768 n=512: %.1f us' % timed(lambda: ops.linear(x, w, None, relu=True)))
