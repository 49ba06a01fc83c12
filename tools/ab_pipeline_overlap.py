"""Render of clip k overlapped with the networks of clip k+1 (two HIP streams) vs back-to-back: frames/s at 720p."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabstitch2_amd import pipeline, synth
from bench import build_nets
dev = torch.device('cuda:0'); torch.set_grad_enabled(False)
nets, _ = build_nets(dev)
n = 32
hr, lr = synth.make_clip_device(n, 720, 1280, seed=0, device=dev)
side = torch.cuda.Stream(dev)
main = torch.cuda.current_stream(dev)

def seq():
    return pipeline.run_two_view(hr[0], hr[1], lr[0], lr[1], nets)

def overlapped(prio=None):
    acc = pipeline.estimate_meshes(nets, lr[0], lr[1])
    ev = torch.cuda.Event(); ev.record(main)
    side.wait_event(ev)
    with torch.cuda.stream(side):
        fr = pipeline.render_frames([hr[0], hr[1]], [acc['smooth_mesh1'], acc['smooth_mesh2']])
    return fr

for name, f in (('sequential', seq), ('render(k) || nets(k+1)', overlapped), ('sequential', seq), ('render(k) || nets(k+1)', overlapped)):
    for _ in range(3): f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(15): f()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('%-26s %.3f ms per clip, %.0f frames/s' % (name, dt / 15 * 1e3, 15 * n / dt))
