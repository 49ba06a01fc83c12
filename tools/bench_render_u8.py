import sys, torch
sys.path.insert(0, '/root/repo')
import bench
from stabstitch2_amd import synth, pipeline, ops, _hip
dev = torch.device('cuda:0'); _hip.lib(); torch.set_grad_enabled(False)
nets, sds = bench.build_nets(dev)
hr, lr = synth.make_clip_device(8, 720, 1280, seed=0, views=2, device=dev)
acc = pipeline.estimate_meshes(nets, lr[0], lr[1])
meshes = [acc['smooth_mesh1'], acc['smooth_mesh2']]
hc, wc, src, T = pipeline.render_plan(meshes, 720, 1280)
fp = ops.render_footprints(src, T, 720, 1280, hc, wc)
u8 = [hr[v].permute(0, 2, 3, 1).round().clamp(0, 255).to(torch.uint8).contiguous() for v in range(2)]
out = torch.empty((3, hc, wc), device=dev); out8 = torch.empty((hc, wc, 3), device=dev, dtype=torch.uint8)
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print('fp32 render %.1f us' % t(lambda: ops.render_average([hr[0][0], hr[1][0]], src[0], T[0], hc, wc, out=out, footprint=fp[0])))
print('canvas_to_u8 %.1f us' % t(lambda: ops.canvas_to_u8(out.unsqueeze(0))))
print('u8 render %.1f us' % t(lambda: ops.render_average_u8([u8[0][0], u8[1][0]], src[0], T[0], hc, wc, out=out8, footprint=fp[0])))
print('ingest hr+lr %.1f us, lr only %.1f us (8 frames)' % (t(lambda: ops.ingest_u8(u8[0])), t(lambda: ops.ingest_u8(u8[0], want_hr=False))))
