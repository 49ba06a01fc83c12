#!/usr/bin/env python
"""The fused AVERAGE render with and without SS_WARP_EPS_FOLD: us per 32-frame 720p clip (HIP events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from stabstitch2_amd import synth, pipeline, ops
dev = torch.device('cuda:0'); torch.set_grad_enabled(False)
nets, _ = bench.build_nets(dev)
hr, lr = synth.make_clip_device(32, 720, 1280, seed=0, device=dev)
acc = pipeline.estimate_meshes(nets, lr[0], lr[1])
res = {}
for fold in (False, True, False, True):
    ops.RENDER_EPS_FOLD = fold
    for _ in range(3):
        out = pipeline.render_frames([hr[0], hr[1]], [acc['smooth_mesh1'], acc['smooth_mesh2']])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        out = pipeline.render_frames([hr[0], hr[1]], [acc['smooth_mesh1'], acc['smooth_mesh2']])
    e1.record(); torch.cuda.synchronize()
    res.setdefault(fold, []).append(e0.elapsed_time(e1) / 20 * 1e3)
print({'render_stage_us_per_clip_default': [round(x, 1) for x in res[False]], 'with_eps_fold': [round(x, 1) for x in res[True]],
       'canvas': list(out[1:3])})
