"""Which dependency edge of HostFrameStream serialises copies and compute?   python tools/diag_host_stream2.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stabstitch2_amd  # noqa
import torch, bench
from collections import deque
from stabstitch2_amd import synth
from stabstitch2_amd.online import OnlineStitcher, HostFrameStream
dev = torch.device('cuda:0'); torch.set_grad_enabled(False)
nets, _ = bench.build_nets(dev)
n = 32
hr, _ = synth.make_clip_device(n, 720, 1280, seed=0, device=dev)
hp = [[hr[v][i].clamp(0, 255).to(torch.uint8).permute(1, 2, 0).contiguous().cpu().pin_memory() for i in range(n)] for v in range(2)]
st = OnlineStitcher(nets, 720, 1280)
r = HostFrameStream(st)
for _ in r.run(tuple(hp[v][t % n] for v in range(2)) for t in range(40)):
    pass
torch.cuda.synchronize()
def loop(skip, K=300):
    staged, results = deque(), deque()
    def stage(k):
        j = r._k % 4; r._k += 1
        with torch.cuda.stream(r.up):
            if r._free[j] is not None and 'free' not in skip:
                r.up.wait_event(r._free[j])
            for dst, t in zip(r._in[j], (hp[0][k % n], hp[1][k % n])):
                dst.copy_(t, non_blocking=True)
            ev = torch.cuda.Event(); ev.record(r.up)
        return j, ev
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k = 0
    for _ in range(2):
        staged.append(stage(k)); k += 1
    while staged:
        j, ev = staged.popleft()
        if k < K:
            staged.append(stage(k)); k += 1
        with torch.cuda.stream(r.comp):
            if 'ready' not in skip:
                r.comp.wait_event(ev)
            outs = st.push_u8(*r._in[j])
            done = torch.cuda.Event(); done.record(r.comp)
        r._free[j] = done
        for o in outs:
            with torch.cuda.stream(r.down):
                if 'done' not in skip:
                    r.down.wait_event(done)
                h = r._host_slot(o.shape)
                h.copy_(o, non_blocking=True)
                if 'rec' not in skip:
                    o.record_stream(r.down)
                e = torch.cuda.Event(); e.record(r.down)
            results.append((e, h))
        while len(results) > r.depth:
            e, h = results.popleft(); e.synchronize()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3
for skip in ((), ('free',), ('ready',), ('done',), ('rec',), ('free', 'ready'), ('free', 'ready', 'done', 'rec')):
    print('skipped edges %-32s %.3f ms per push' % (skip, loop(skip)), flush=True)
