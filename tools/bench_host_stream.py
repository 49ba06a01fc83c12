"""HostFrameStream in isolation: uint8 frames in pinned host memory -> uint8 frames in pinned host memory, one pair per push.
    python tools/bench_host_stream.py [--pushes 300] [--views 2|3] [--dummies K]   (K idle streams created first: queue lottery)"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stabstitch2_amd  # noqa
import torch
import bench
from stabstitch2_amd import synth
from stabstitch2_amd.online import OnlineStitcher, ThreeViewOnlineStitcher, HostFrameStream, PipelinedOnlineStitcher
ap = argparse.ArgumentParser()
ap.add_argument('--pushes', type=int, default=300)
ap.add_argument('--views', type=int, default=2)
ap.add_argument('--dummies', type=int, default=0)
ap.add_argument('--depth', type=int, default=4)
ap.add_argument('--prefetch', type=int, default=2)
ap.add_argument('--pipelined', action='store_true')
ap.add_argument('--no-up', action='store_true')
ap.add_argument('--no-down', action='store_true')
ap.add_argument('--cpu-only', action='store_true')
args = ap.parse_args()
dev = torch.device('cuda:0'); torch.set_grad_enabled(False)
nets, _ = bench.build_nets(dev)
n = 32
hr, _ = synth.make_clip_device(n, 720, 1280, seed=0, views=args.views, device=dev)
hp = [[hr[v][i].clamp(0, 255).to(torch.uint8).permute(1, 2, 0).contiguous().cpu().pin_memory() for i in range(n)] for v in range(args.views)]
dummies = []
for _ in range(args.dummies):
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        torch.zeros(1, device=dev)
    dummies.append(s)
st = (ThreeViewOnlineStitcher if args.views == 3 else (PipelinedOnlineStitcher if args.pipelined else OnlineStitcher))(nets, 720, 1280)
runner = HostFrameStream(st, depth=args.depth, prefetch=args.prefetch)
if args.no_up:          # experiment: the frames are uploaded once, later stages reuse the slots
    real_stage = runner._stage
    cache = {}
    def stage(frames):
        if len(cache) < args.prefetch + 2:
            j, ev = real_stage(frames)
            cache[j] = ev
            return j, ev
        j = runner._k % (args.prefetch + 2)
        runner._k += 1
        return j, cache[j]
    runner._stage = stage
if args.no_down:        # experiment: results stay on the device
    import types
    def run(self, source):
        it = iter(source)
        from collections import deque
        staged = deque()
        for _ in range(self.prefetch):
            nxt = next(it, None)
            if nxt is not None:
                staged.append(self._stage(nxt))
        while staged:
            j, ev = staged.popleft()
            nxt = next(it, None)
            if nxt is not None:
                staged.append(self._stage(nxt))
            with torch.cuda.stream(self.comp):
                self.comp.wait_event(ev)
                outs = self.st.push_u8(*self._in[j])
                done = torch.cuda.Event()
                done.record(self.comp)
            self._free[j] = done
            for o in outs:
                yield o
    runner.run = types.MethodType(run, runner)
seq = lambda k: (tuple(hp[v][t % n] for v in range(args.views)) for t in range(k))
for _ in runner.run(seq(40)):
    pass
torch.cuda.synchronize()
t0 = time.perf_counter()
nout = sum(1 for _ in runner.run(seq(args.pushes)))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print({'views': args.views, 'pushes': args.pushes, 'frames': nout, 'ms_per_push': round(dt / args.pushes * 1e3, 4), 'fps': round(nout / dt, 1),
       'dummies': args.dummies})
