import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from stabstitch2_amd import synth, pipeline, _hip as H
dev = torch.device('cuda:0'); torch.set_grad_enabled(False)
nets, _ = bench.build_nets(dev)
hr, lr = synth.make_clip_device(32, 720, 1280, seed=0, device=dev)
for _ in range(2): pipeline.run_two_view(hr[0], hr[1], lr[0], lr[1], nets)
orig = H.call
def spy(name, *a):
    if 'pack' in name:
        print('PACK', name); traceback.print_stack(limit=8)
    return orig(name, *a)
H.call = spy
import stabstitch2_amd.ops as ops
ops.H.call = spy
pipeline.run_two_view(hr[0], hr[1], lr[0], lr[1], nets)
