"""Do small pinned H2D / D2H copies on their own streams overlap a stream of graph replays?   python tools/diag_copy_overlap.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stabstitch2_amd  # noqa
import torch, bench
from stabstitch2_amd import synth
from stabstitch2_amd.online import OnlineStitcher
dev = torch.device('cuda:0'); torch.set_grad_enabled(False)
nets, _ = bench.build_nets(dev)
n = 32
hr, _ = synth.make_clip_device(n, 720, 1280, seed=0, device=dev)
u8 = [[hr[v][i].clamp(0, 255).to(torch.uint8).permute(1, 2, 0).contiguous() for i in range(n)] for v in range(2)]
hp = [[f.cpu().pin_memory() for f in v] for v in u8]
st = OnlineStitcher(nets, 720, 1280)
for t in range(12):
    st.push_u8(u8[0][t], u8[1][t])
torch.cuda.synchronize()
up, down, comp = torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev)
scratch = [torch.empty_like(u8[0][0]) for _ in range(4)]
hout = torch.empty((st.hc, st.wc, 3), dtype=torch.uint8).pin_memory()
dsrc = torch.empty((st.hc, st.wc, 3), dtype=torch.uint8, device=dev)
def run(mode, K=300):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(K):
        i = t % n
        if 'h2d' in mode:
            with torch.cuda.stream(up):
                scratch[(2 * t) % 4].copy_(hp[0][i], non_blocking=True)
                scratch[(2 * t + 1) % 4].copy_(hp[1][i], non_blocking=True)
        with torch.cuda.stream(comp if 'side' in mode else torch.cuda.current_stream(dev)):
            st.push_u8(u8[0][i], u8[1][i])
        if 'd2h' in mode:
            with torch.cuda.stream(down):
                hout.copy_(dsrc, non_blocking=True)
        if t % 8 == 7:
            torch.cuda.current_stream(dev).synchronize() if 'side' not in mode else comp.synchronize()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3
for mode in ('compute only', 'compute only side', 'h2d side', 'd2h side', 'h2d d2h side', 'h2d d2h'):
    print('%-22s %.3f ms per push' % (mode, run(mode)), flush=True)
# copies alone
torch.cuda.synchronize(); t0 = time.perf_counter()
for t in range(300):
    with torch.cuda.stream(up):
        scratch[0].copy_(hp[0][t % n], non_blocking=True); scratch[1].copy_(hp[1][t % n], non_blocking=True)
torch.cuda.synchronize(); print('h2d pair alone         %.3f ms' % ((time.perf_counter() - t0) / 300 * 1e3))
torch.cuda.synchronize(); t0 = time.perf_counter()
for t in range(300):
    with torch.cuda.stream(down):
        hout.copy_(dsrc, non_blocking=True)
torch.cuda.synchronize(); print('d2h frame alone        %.3f ms' % ((time.perf_counter() - t0) / 300 * 1e3))
