"""The 4-head FC layers: ss_linear_grouped against the same products as a grouped 1x1 convolution on the conv engine."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabstitch2_amd import ops
dev = torch.device('cuda:0'); torch.manual_seed(0)
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps): fn()
    torch.cuda.current_stream().wait_stream(s)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * reps)
for G, m, k, n in ((4, 32, 1536, 1024), (4, 32, 1024, 512), (4, 32, 512, 126), (3, 32, 1536, 1024), (4, 1, 1536, 1024), (4, 8, 1536, 1024)):
    x = torch.randn(G, m, k, device=dev); w = torch.randn(G, n, k, device=dev) * 0.05; b = torch.randn(G, n, device=dev)
    ref = torch.relu(torch.einsum('gmk,gnk->gmn', x.double(), w.double()) + b.double()[:, None])
    a = ops.linear_grouped(x, w, b, relu=True)
    cw = w.view(G, n, 1, 1, 1, k)
    c = ops.conv_grouped(x.view(G, 1, 1, m, k), cw, b, None, stride=1, pad=(0, 0, 0), relu=True).view(G, m, n)
    print('G=%d m=%2d K=%4d N=%4d: linear_grouped %.1f us (err %.1e)   grouped conv-engine GEMM %.1f us (err %.1e)' % (
        G, m, k, n, timed(lambda: ops.linear_grouped(x, w, b, relu=True)), float((a - ref).abs().max()),
        timed(lambda: ops.conv_grouped(x.view(G, 1, 1, m, k), cw, b, None, stride=1, pad=(0, 0, 0), relu=True)), float((c - ref).abs().max())))
