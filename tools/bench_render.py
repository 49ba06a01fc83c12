"""Fused AVERAGE render of one 720p frame: full evaluation vs footprint skipping (us per frame, skipped share)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabstitch2_amd import ops, pipeline, synth
from bench import build_nets
dev = torch.device('cuda:0')
torch.set_grad_enabled(False)
nets, _ = build_nets(dev)
for views in (2, 3):
    n = 8
    hr, lr = synth.make_clip_device(n, 720, 1280, seed=0, views=views, device=dev)
    if views == 2:
        acc = pipeline.estimate_meshes(nets, lr[0], lr[1])
        meshes, pres = [acc['smooth_mesh1'], acc['smooth_mesh2']], False
    else:
        a12 = pipeline.estimate_meshes(nets, lr[0], lr[1]); a23 = pipeline.estimate_meshes(nets, lr[1], lr[2])
        meshes, pres = list(pipeline.three_view_compose(a12['smooth_mesh1'], a12['smooth_mesh2'], a23['smooth_mesh1'], a23['smooth_mesh2'], 720, 1280)), True
    hc, wc, src, T = pipeline.render_plan(meshes, 720, 1280, pres)
    fp = ops.render_footprints(src, T, 720, 1280, hc, wc)
    out = torch.empty((3, hc, wc), device=dev)
    imgs = [hr[k][0] for k in range(views)]
    res = {}
    for name, f in (('full', None), ('footprint', fp[0])):
        for _ in range(3): ops.render_average(imgs, src[0], T[0], hc, wc, 'NORMAL', out=out, footprint=f)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.render_average(imgs, src[0], T[0], hc, wc, 'NORMAL', out=out, footprint=f)
        e1.record(); torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / 20 * 1e3
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); ops.render_footprints(src, T, 720, 1280, hc, wc); e1.record(); torch.cuda.synchronize()
    ny, nx = (hc + 7) // 8 + 1, (wc + 63) // 64 + 1
    mar = fp[0, views * ny * nx * 2:views * ny * nx * 2 + 4 * views].view(views, 4)
    print('%d views, canvas %dx%d: full %.1f us, with footprints %.1f us per frame; footprint pass for %d frames %.1f us; hulls (normalised canvas) %s'
          % (views, hc, wc, res['full'], res['footprint'], n, e0.elapsed_time(e1) * 1e3, mar.cpu().tolist()))
