"""Time the device metric harness (alignment PSNR/SSIM, stability, distortion) on a synthetic clip."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from stabstitch2_amd import synth, metrics
dev = torch.device('cuda:0'); torch.set_grad_enabled(False)
nets, _ = bench.build_nets(dev)
hr, lr = synth.make_clip_device(32, 360, 480, seed=0, device=dev)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = metrics.evaluate_clip(nets, lr[0], lr[1])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('evaluate_clip 32 frames: %.2f ms' % (dt * 1e3), {k: (round(float(v), 4) if not hasattr(v, 'shape') or v.numel() == 1 else tuple(v.shape)) for k, v in out.items()} if isinstance(out, dict) else type(out))
