"""Where does HostClipRunner lose time?  Host-side timestamps per phase."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench
from stabstitch2_amd import synth, pipeline, ops

dev = torch.device('cuda:0')
torch.set_grad_enabled(False)
nets, _ = bench.build_nets(dev)
hr, lr = synth.make_clip_device(32, 720, 1280, seed=0, device=dev)
u8d = [hr[v].permute(0, 2, 3, 1).round().clamp(0, 255).to(torch.uint8).contiguous() for v in range(2)]
u8h = [t.cpu().pin_memory() for t in u8d]
del hr

R = pipeline.HostClipRunner(nets, dev)
marks = []
for name in ('_upload', '_compute', '_download'):
    fn = getattr(R, name)
    def wrap(*a, _fn=fn, _n=name, **k):
        t0 = time.perf_counter(); r = _fn(*a, **k); marks.append((_n, (time.perf_counter() - t0) * 1e3)); return r
    setattr(R, name, wrap)

def run(k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in R.run((u8h[0], u8h[1]) for _ in range(k)):
        pass
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3

run(2); marks.clear()
print('runner ms/clip', run(8))
for n in ('_upload', '_compute', '_download'):
    v = [m for k, m in marks if k == n]
    print(n, ' '.join('%.2f' % x for x in v))

def sync_u8(k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k):
        pipeline.run_two_view_u8(u8d[0], u8d[1], nets, device=dev)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3
sync_u8(2)
print('device-resident u8 ms/clip', sync_u8(8))
s = torch.cuda.Stream(dev)
with torch.cuda.stream(s):
    sync_u8(2)
    print('same on a side stream', sync_u8(8))
