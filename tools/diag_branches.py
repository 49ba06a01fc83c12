#!/usr/bin/env python
"""Do forked branches of ONE captured HIP graph run side by side on this runtime?  Two independent chains of N small launches,
captured (a) on one stream, (b) on two streams forked / joined inside the capture; us per replay."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
import torch
from stabstitch2_amd import ops

dev = torch.device('cuda:0')
torch.set_grad_enabled(False)
N = 20
xa = torch.randn((2, 23, 30, 256), device=dev)
xb = torch.randn((2, 23, 30, 256), device=dev)
wg = (torch.randn((256, 1, 3, 3, 256), device=dev) / 48)


def chain(x, kind):
    for _ in range(N):
        x = ops.l2norm(x) if kind == 'small' else ops.conv(x, wg, relu=True)
    return x


def timed(g, reps=20):
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for kind in ('small', 'conv'):
    chain(xa, kind); chain(xb, kind); torch.cuda.synchronize()
    g1 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1):
        ya = chain(xa, kind); yb = chain(xb, kind)
    side = torch.cuda.Stream(dev)
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        cur = torch.cuda.current_stream(dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            yb2 = chain(xb, kind)
        ya2 = chain(xa, kind)
        cur.wait_stream(side)
    g3 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g3):
        ya3 = chain(xa, kind)
    t1, t2, t3 = timed(g1), timed(g2), timed(g3)
    print('%-5s one chain of %d: %.1f us | two chains serial: %.1f us | two chains forked: %.1f us | equal %s'
          % (kind, N, t3, t1, t2, torch.equal(yb, yb2) and torch.equal(ya, ya2)))
