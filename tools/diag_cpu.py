import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from stabstitch2_amd import synth
from oracle import nets as ON, pipeline as OP
nt = int(sys.argv[1]); torch.set_num_threads(nt); torch.set_grad_enabled(False)
nets = []
for cls in (ON.SpatialNet, ON.TemporalNet, ON.SmoothNet):
    m = cls().eval(); m.load_state_dict(synth.synthetic_state_dict(m)); nets.append(m)
n = 8
hr, lr = synth.make_clip_device(n, 720, 1280, seed=0, device='cpu')
L = lambda x: [x[i:i + 1] for i in range(n)]
t = time.time(); s = OP.spatial_stage(nets[0], L(lr[0]), L(lr[1])); t1 = time.time() - t
t = time.time(); a = OP.temporal_stage(nets[1], L(lr[0])); t2 = time.time() - t
t = time.time(); acc = OP.estimate_meshes(nets, L(lr[0]), L(lr[1])); t3 = time.time() - t
t = time.time(); fr = OP.get_stable_sqe(L(hr[0]), L(hr[1]), acc['smooth_mesh1'], acc['smooth_mesh2'], 'NORMAL', 'AVERAGE'); t4 = time.time() - t
print('threads %d: spatial %.2f temporal(1 view) %.2f estimate %.2f render %.2f -> %.3f fps' % (nt, t1, t2, t3, t4, n / (t3 + t4)), flush=True)
