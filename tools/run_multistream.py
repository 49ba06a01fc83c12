"""Steady state of the batch-of-streams streaming mode (for profiling):  python tools/run_multistream.py [S] [pushes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stabstitch2_amd  # noqa
import torch
import bench
from stabstitch2_amd import synth
from stabstitch2_amd.online import MultiOnlineStitcher, OnlineStitcher, PipelinedMultiOnlineStitcher
S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
pushes = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device('cuda:0'); torch.set_grad_enabled(False)
nets, _ = bench.build_nets(dev)
hr, lr = synth.make_clip_device(max(S, 2), 720, 1280, seed=0, device=dev)
a = [hr[0][:S].contiguous(), hr[1][:S].contiguous(), lr[0][:S].contiguous(), lr[1][:S].contiguous()]
shared = 'shared' in sys.argv[3:]
piped = 'pipelined' in sys.argv[3:]
canv = [(-20.0, 1880.0, -15.0, 745.0)] * S if shared else None          # one canvas size for all streams: one render launch per push
st = (PipelinedMultiOnlineStitcher if piped else MultiOnlineStitcher)(nets, 720, 1280, streams=S, canvases=canv) if S > 1 else None
if S == 1:
    one = OnlineStitcher(nets, 720, 1280)
    push = lambda: one.push(*a)
else:
    push = lambda: st.push(*a)
for _ in range(12):
    push()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(pushes):
    push()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print('S=%d%s%s: %.3f ms per push, %.0f frames/s aggregate' % (S, ' shared canvas size' if S > 1 and shared else '', ' two pushes in flight' if piped else '', dt / pushes * 1e3, S * pushes / dt))
