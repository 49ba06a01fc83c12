#!/usr/bin/env python
"""conv_wino43p_kernel (persistent workgroups, next block's rows requested in front of the epilogue) against conv_wino43_kernel
(one workgroup per tile block): bit-identical outputs, us per launch on the trunk's layers (64 images), with / without residual."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stabstitch2_amd import ops, _hip as H

dev = torch.device('cuda:0')
torch.manual_seed(0)


def timeit(fn, iters=30):
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


for (c, h, w, n) in [(64, 90, 120, 64), (128, 45, 60, 64), (256, 23, 30, 64), (64, 90, 120, 7), (128, 45, 60, 3), (16, 45, 60, 40), (64, 13, 17, 2)]:
    co = 64 if c == 16 else c
    x = torch.randn(n, h, w, c, device=dev)
    wgt = torch.randn(co, 1, 3, 3, c, device=dev) * (1.0 / (9 * c)) ** 0.5
    bias = torch.randn(co, device=dev) * 0.1
    res = torch.randn(n, h, w, co, device=dev)
    for use_res in (False, True):
        r = res if use_res else None
        out = {}
        t = {}
        for persist in (0, 1, 0, 1):
            H.lib().ss_wino43_set_persistent(persist)
            y = ops.conv_winograd43(x, wgt, bias, r, True)
            torch.cuda.synchronize()
            out[persist] = y
            t.setdefault(persist, []).append(timeit(lambda: ops.conv_winograd43(x, wgt, bias, r, True)))
        print('%3d->%3d %3dx%3d x%2d res=%d   block/wg %7.1f %7.1f us   persistent %7.1f %7.1f us   equal %s'
              % (c, co, h, w, n, use_res, t[0][0], t[0][1], t[1][0], t[1][1], torch.equal(out[0], out[1])))
