import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from stabstitch2_amd import synth, pipeline
dev = torch.device('cuda:0'); torch.set_grad_enabled(False)
nets, _ = bench.build_nets(dev)
hr, lr = synth.make_clip_device(32, 720, 1280, 0, device=dev)
ref = None
for rnd in range(5):
    for ov in (False, True):
        pipeline.OVERLAP_STREAMS = ov
        for _ in range(2): out = pipeline.run_two_view(hr[0], hr[1], lr[0], lr[1], nets)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): out = pipeline.run_two_view(hr[0], hr[1], lr[0], lr[1], nets)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        if ref is None: ref = out[3].clone()
        print('overlap=%d  %.3f ms/clip  %.1f fps  mesh equal %s' % (ov, dt * 1e3, 32 / dt, torch.equal(out[3], ref)), flush=True)
