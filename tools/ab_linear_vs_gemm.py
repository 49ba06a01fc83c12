"""The regressors' FC layers: ss_linear (one wave per output neuron, 8 batch rows per pass) against the same product on the conv
engine (a 1x1 convolution over a [1, 1, m, K] strip: fp32 MFMA tiles, split-K), per launch.     python tools/ab_linear_vs_gemm.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabstitch2_amd import ops
dev = torch.device('cuda:0')
torch.manual_seed(0)


def timed(fn, reps=20):
    """Per call, GPU time: `reps` calls captured into one HIP graph (the eager loop is bound by the host's ~10 us per call)."""
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
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


for m, k, n in ((32, 1536, 1024), (62, 1536, 1024), (32, 1024, 512), (62, 1024, 512), (32, 512, 126), (32, 768, 512), (32, 512, 128), (2, 1536, 1024), (1, 1536, 1024)):
    x = torch.randn(m, k, device=dev)
    w = torch.randn(n, k, device=dev) * 0.05
    b = torch.randn(n, device=dev)
    a = ops.linear(x, w, b, relu=True)
    c = ops.conv(x.view(1, 1, m, k), w.view(n, 1, 1, 1, k), b, None, stride=1, pad=(0, 0, 0), relu=True).view(m, n)
    ref = torch.relu(x.double() @ w.double().t() + b.double())
    print('m=%2d K=%4d N=%4d: linear %.1f us (err %.1e)   conv-engine GEMM %.1f us (err %.1e)' % (
        m, k, n, timed(lambda: ops.linear(x, w, b, relu=True)), float((a - ref).abs().max()),
        timed(lambda: ops.conv(x.view(1, 1, m, k), w.view(n, 1, 1, 1, k), b, None, stride=1, pad=(0, 0, 0), relu=True)),
        float((c - ref).abs().max())))
