"""Where a HostFrameStream iteration spends its HOST time (perf_counter around the phases) -- python tools/diag_host_stream.py"""
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
T = {'stage': 0.0, 'push': 0.0, 'down': 0.0, 'wait': 0.0}
pc = time.perf_counter
staged, results = deque(), deque()
K = 300
t_all = pc()
k = 0
for _ in range(r.prefetch):
    a = pc(); staged.append(r._stage(tuple(hp[v][k % n] for v in range(2)))); T['stage'] += pc() - a; k += 1
while staged:
    j, ev = staged.popleft()
    if k < K:
        a = pc(); staged.append(r._stage(tuple(hp[v][k % n] for v in range(2)))); T['stage'] += pc() - a; k += 1
    a = pc()
    with torch.cuda.stream(r.comp):
        r.comp.wait_event(ev)
        outs = st.push_u8(*r._in[j])
        done = torch.cuda.Event(); done.record(r.comp)
    r._free[j] = done
    T['push'] += pc() - a
    a = pc()
    for o in outs:
        with torch.cuda.stream(r.down):
            r.down.wait_event(done)
            h = r._host_slot(o.shape)
            h.copy_(o, non_blocking=True)
            o.record_stream(r.down)
            e = torch.cuda.Event(); e.record(r.down)
        results.append((e, h))
    T['down'] += pc() - a
    a = pc()
    while len(results) > r.depth:
        e, h = results.popleft(); e.synchronize()
    T['wait'] += pc() - a
torch.cuda.synchronize()
tot = pc() - t_all
print('per iteration (ms): total %.3f | host time in: %s' % (tot / K * 1e3, {k_: round(v / K * 1e3, 3) for k_, v in T.items()}))
