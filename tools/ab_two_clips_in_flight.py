"""Throughput experiment: two independent 720p clips in flight on two HIP streams driven by two host threads (ctypes and the
canvas-size read-back release the GIL), against the same clips back to back on one stream.   python tools/ab_two_clips_in_flight.py"""
import sys, os, time, threading, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from stabstitch2_amd import synth, pipeline
dev = torch.device('cuda:0')
torch.set_grad_enabled(False)
nets, _ = bench.build_nets(dev)
clips = [synth.make_clip_device(32, 720, 1280, seed=s, device=dev) for s in (0, 1)]
def run(k, reps, stream=None):
    hr, lr = clips[k]
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        for _ in range(reps):
            pipeline.run_two_view(hr[0], hr[1], lr[0], lr[1], nets)
for k in (0, 1):
    run(k, 3)
torch.cuda.synchronize()
R = 20
for rnd in range(3):
    t0 = time.perf_counter(); run(0, R); run(1, R); torch.cuda.synchronize(); t1 = time.perf_counter() - t0
    s = [torch.cuda.Stream(), torch.cuda.Stream()]
    th = [threading.Thread(target=run, args=(k, R, s[k])) for k in (0, 1)]
    t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]; torch.cuda.synchronize(); t2 = time.perf_counter() - t0
    print('one stream: %.1f frames/s   two clips in flight: %.1f frames/s (%.3fx)' % (2 * R * 32 / t1, 2 * R * 32 / t2, t1 / t2), flush=True)
