"""The host->host path alone (for profiling): uint8 clips in pinned host memory -> stitched uint8 clips in pinned host memory.
    python tools/run_hostpath.py [clips] [fusion]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stabstitch2_amd  # noqa: F401  (GPU_MAX_HW_QUEUES before the runtime starts)
import torch
import bench
from stabstitch2_amd import synth, pipeline, hostbind

clips = int(sys.argv[1]) if len(sys.argv) > 1 else 10
fusion = sys.argv[2] if len(sys.argv) > 2 else 'AVERAGE'
dev = torch.device('cuda:0')
torch.cuda.set_device(0)
print(hostbind.bind_to_gpu(dev))
torch.set_grad_enabled(False)
nets, _ = bench.build_nets(dev)
hr, lr = synth.make_clip_device(32, 720, 1280, seed=0, device=dev)
u8 = [hr[v].permute(0, 2, 3, 1).round().clamp(0, 255).to(torch.uint8).contiguous().cpu().pin_memory() for v in range(2)]
del hr, lr
runner = pipeline.HostClipRunner(nets, dev, fusion_mode=fusion)
runner.timed = True


def run(k):
    t0 = time.perf_counter()
    for _ in runner.run((u8[0], u8[1]) for _ in range(k)):
        pass
    torch.cuda.synchronize()
    return time.perf_counter() - t0


run(3)
runner.copy_stats()
for _ in range(3):
    dt = run(clips)
    print('%.2f ms/clip  %.0f fps' % (dt / clips * 1e3, 32 * clips / dt), runner.copy_stats())
