"""HostClipRunner: uploads one or two clips ahead; run totals and the distribution of per-clip periods."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stabstitch2_amd  # noqa
import torch
import bench
from stabstitch2_amd import synth, pipeline, hostbind
dev = torch.device('cuda:0'); torch.cuda.set_device(0)
hostbind.bind_to_gpu(dev)
torch.set_grad_enabled(False)
nets, _ = bench.build_nets(dev)
hr, lr = synth.make_clip_device(32, 720, 1280, seed=0, device=dev)
u8 = [hr[v].permute(0, 2, 3, 1).round().clamp(0, 255).to(torch.uint8).contiguous().cpu().pin_memory() for v in range(2)]
del hr, lr
for rounds in range(4):
    for pf in (1, 2):
        r = pipeline.HostClipRunner(nets, dev, prefetch=pf)
        for _ in r.run((u8[0], u8[1]) for _ in range(4)): pass
        torch.cuda.synchronize()
        st = []; t0 = time.perf_counter()
        for _ in r.run((u8[0], u8[1]) for _ in range(40)): st.append(time.perf_counter())
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        g = sorted((b - a) * 1e3 for a, b in zip(st[:-1], st[1:]))
        print('prefetch %d: %.0f fps total; period ms median %.2f p90 %.2f max %.2f' % (pf, 32 * 40 / dt, g[len(g) // 2], g[int(len(g) * 0.9)], g[-1]))
