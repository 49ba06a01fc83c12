import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
import torch, bench
from stabstitch2_amd import synth
from stabstitch2_amd.online import PipelinedThreeViewOnlineStitcher
dev = torch.device('cuda:0'); torch.set_grad_enabled(False)
nets, _ = bench.build_nets(dev)
n = 32
hr, lr = synth.make_clip_device(n, 720, 1280, seed=0, views=3, device=dev)
def run(pushes=100):
    st = PipelinedThreeViewOnlineStitcher(nets, 720, 1280)
    for t in range(12):
        st.push(hr[0][t:t + 1], hr[1][t:t + 1], hr[2][t:t + 1], lr[0][t:t + 1], lr[1][t:t + 1], lr[2][t:t + 1])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(pushes):
        i = t % n
        st.push(hr[0][i:i + 1], hr[1][i:i + 1], hr[2][i:i + 1], lr[0][i:i + 1], lr[1][i:i + 1], lr[2][i:i + 1])
    torch.cuda.synchronize()
    run.probe = st.stream_probe_ms
    return (time.perf_counter() - t0) / pushes * 1e3
dummies = []
out = []
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    out.append(round(run(), 3))
    s = torch.cuda.Stream(); 
    with torch.cuda.stream(s):
        torch.zeros(1, device=dev)
    dummies.append(s)
print(getattr(run, "probe", None)); print("GPU_MAX_HW_QUEUES", os.environ['GPU_MAX_HW_QUEUES'], 'ms/push per successive stitcher (one dummy stream created between):', out)
