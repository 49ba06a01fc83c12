"""Every conv-engine launch of one 720p 2-view clip, aggregated by shape (bench.ConvProbe's table): rows M, cout, taps,
cin, launches, ms, direct-conv GFLOP, direct-conv-equivalent TF/s.      python tools/list_convs.py"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from stabstitch2_amd import synth, pipeline, _hip
dev = torch.device('cuda:0'); _hip.lib(); torch.set_grad_enabled(False)
nets, sds = bench.build_nets(dev)
hr, lr = synth.make_clip_device(32, 720, 1280, seed=0, views=2, device=dev)
probe = bench.ConvProbe(); probe.install()
step = lambda: pipeline.run_two_view(hr[0], hr[1], lr[0], lr[1], nets, 'NORMAL', 'AVERAGE')
for _ in range(3): step()
torch.cuda.synchronize()
probe.active = True; step(); torch.cuda.synchronize(); probe.active = False
probe.report()
ms, flop, n, nbytes, ex = probe.summary()
print('total %.3f ms in %d launches; %.1f GFLOP direct-equivalent, %.1f executed' % (ms, n, flop / 1e9, ex / 1e9), file=sys.stderr)
