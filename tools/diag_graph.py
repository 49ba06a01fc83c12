import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from stabstitch2_amd import synth, pipeline, ops
from stabstitch2_amd.spatial_network import build_SpatialNet
dev = torch.device('cuda:0'); torch.set_grad_enabled(False)
nets, _ = bench.build_nets(dev)
hr, lr = synth.make_clip_device(8, 720, 1280, 0, device=dev)
a = lr[0][:1].clone(); b = lr[1][:1].clone()
ref = build_SpatialNet(nets[0], a, b)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): build_SpatialNet(nets[0], a, b)
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    out = build_SpatialNet(nets[0], a, b)
g.replay(); torch.cuda.synchronize()
print('graph == eager:', torch.equal(out['motion1'], ref['motion1']), float((out['motion1'] - ref['motion1']).abs().max()))
a.copy_(lr[0][3:4]); b.copy_(lr[1][3:4]); g.replay(); torch.cuda.synchronize()
ref2 = build_SpatialNet(nets[0], lr[0][3:4], lr[1][3:4])
print('graph new input == eager:', float((out['motion1'] - ref2['motion1']).abs().max()))
t = time.perf_counter()
for _ in range(50): g.replay()
torch.cuda.synchronize(); print('graph replay %.3f ms' % ((time.perf_counter() - t) / 50 * 1e3))
t = time.perf_counter()
for _ in range(50): build_SpatialNet(nets[0], a, b)
torch.cuda.synchronize(); print('eager %.3f ms' % ((time.perf_counter() - t) / 50 * 1e3))
