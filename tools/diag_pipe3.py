import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
import torch, bench
from stabstitch2_amd import synth
from stabstitch2_amd.online import ThreeViewOnlineStitcher, PipelinedThreeViewOnlineStitcher, PipelinedOnlineStitcher
dev = torch.device('cuda:0'); torch.set_grad_enabled(False)
nets, _ = bench.build_nets(dev)
n = 32
hr, lr = synth.make_clip_device(n, 720, 1280, seed=0, views=3, device=dev)
def run(cls, tag, pushes=100):
    st = cls(nets, 720, 1280)
    for t in range(12):
        st.push(hr[0][t:t + 1], hr[1][t:t + 1], hr[2][t:t + 1], lr[0][t:t + 1], lr[1][t:t + 1], lr[2][t:t + 1])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(pushes):
        i = t % n
        st.push(hr[0][i:i + 1], hr[1][i:i + 1], hr[2][i:i + 1], lr[0][i:i + 1], lr[1][i:i + 1], lr[2][i:i + 1])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('%-10s %d pushes: %.3f ms per push (host enqueue %.3f ms per push)' % (tag, pushes, (t2 - t0) / pushes * 1e3, (t1 - t0) / pushes * 1e3))
    return st
order = sys.argv[1:] or ['pipe', 'plain', 'pipe']
keep = []
for o in order:
    keep.append(run(PipelinedThreeViewOnlineStitcher if o == 'pipe' else ThreeViewOnlineStitcher, o))
    if 'drop' in os.environ.get('MODE', ''):
        keep.clear()
