"""LINEAR fusion: frames per launch group (workspace residency in the 256 MB MALL vs launch count)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stabstitch2_amd  # noqa
import torch
import bench
from stabstitch2_amd import synth, pipeline
dev = torch.device('cuda:0'); torch.set_grad_enabled(False)
nets, _ = bench.build_nets(dev)
views = int(sys.argv[1]) if len(sys.argv) > 1 else 2
hr, lr = synth.make_clip_device(32, 720, 1280, seed=0, views=views, device=dev)
def run():
    if views == 2:
        return pipeline.run_two_view(hr[0], hr[1], lr[0], lr[1], nets, 'NORMAL', 'LINEAR')
    return pipeline.run_three_view(hr[0], hr[1], hr[2], lr[0], lr[1], lr[2], nets, 'NORMAL', 'LINEAR')
from stabstitch2_amd import _hip
for rounds in range(2):
    for ch in (-1, 0, 48, 64, 128, 192, 368):
        _hip.lib().ss_linear_clip_set_rows(ch)
        for _ in range(3): run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): run()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print('views %d rows %3d: %.3f ms/clip  %.0f fps' % (views, ch, dt * 1e3, 32 / dt))
