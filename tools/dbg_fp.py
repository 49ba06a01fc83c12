"""Fused render, forced tile classes (tuning build): time per frame when every tile takes one path."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabstitch2_amd import ops, pipeline, synth, _hip as H
import _tuning
lib = _tuning.lib()
from bench import build_nets
dev = torch.device('cuda:0'); torch.set_grad_enabled(False)
nets, _ = build_nets(dev)
hr, lr = synth.make_clip_device(8, 720, 1280, seed=0, views=2, device=dev)
acc = pipeline.estimate_meshes(nets, lr[0], lr[1])
hc, wc, src, T = pipeline.render_plan([acc['smooth_mesh1'], acc['smooth_mesh2']], 720, 1280, False)
fp = ops.render_footprints(src, T, 720, 1280, hc, wc)
imgs = [hr[0][0].contiguous(), hr[1][0].contiguous()]
arr = H.ptr_array(imgs)
out = torch.empty((3, hc, wc), device=dev)
for name, forced, f in (('full (no footprint)', 0, None), ('footprint', 0, fp[0]), ('forced both', 4, fp[0]), ('forced view 0 only', 2, fp[0]),
                        ('forced view 1 only', 3, fp[0]), ('forced none', 1, fp[0]), ('checkerboard single/both', 9, fp[0]),
                        ('left half single, right half both', 10, fp[0])):
    mode = forced << 8
    def run():
        H.call('ss_render_average', arr, H.dptr(src[0].contiguous()), H.dptr(T[0].contiguous()), H.dptr(f, True), H.dptr(out), 2, 720, 1280, hc, wc, mode, H.stream())
    for _ in range(3): run()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    print('%-22s %.1f us' % (name, e0.elapsed_time(e1) / 20 * 1e3))
