#!/usr/bin/env python
"""Steady-state streaming (batch 1) in isolation: ms per push, graph node count, HIP-event time of the replay alone.
    python tools/bench_stream.py [--pushes 400] [--views 2|3] [--eager] [--height 720 --width 1280]
Same measurement as bench.py's `streaming ... fps_steady` (push = input copies + graph replay + clone of the frame), plus the
replay alone.  Used under rocprofv3 by tools/profile_stream.sh (short run) for tools/push_timeline.py."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
import torch

import bench
from stabstitch2_amd import synth
from stabstitch2_amd.online import OnlineStitcher, ThreeViewOnlineStitcher, PipelinedOnlineStitcher, PipelinedThreeViewOnlineStitcher


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--pushes', type=int, default=400)
    ap.add_argument('--views', type=int, default=2)
    ap.add_argument('--height', type=int, default=720)
    ap.add_argument('--width', type=int, default=1280)
    ap.add_argument('--eager', action='store_true')
    ap.add_argument('--fusion', default='AVERAGE')
    ap.add_argument('--deterministic', action='store_true')
    ap.add_argument('--pipelined', action='store_true')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    torch.set_grad_enabled(False)
    nets, _ = bench.build_nets(dev)
    n = 32
    hr, lr = synth.make_clip_device(n, args.height, args.width, seed=0, views=args.views, device=dev)
    if args.views == 3:
        if args.pipelined:
            st = PipelinedThreeViewOnlineStitcher(nets, args.height, args.width, fusion_mode=args.fusion)
        else:
            st = ThreeViewOnlineStitcher(nets, args.height, args.width, use_graph=not args.eager, fusion_mode=args.fusion)
        push = lambda i: st.push(hr[0][i:i + 1], hr[1][i:i + 1], hr[2][i:i + 1], lr[0][i:i + 1], lr[1][i:i + 1], lr[2][i:i + 1])
    else:
        if args.pipelined:
            st = PipelinedOnlineStitcher(nets, args.height, args.width, fusion_mode=args.fusion, deterministic=args.deterministic)
        else:
            st = OnlineStitcher(nets, args.height, args.width, use_graph=not args.eager, fusion_mode=args.fusion, deterministic=args.deterministic)
        push = lambda i: st.push(hr[0][i:i + 1], hr[1][i:i + 1], lr[0][i:i + 1], lr[1][i:i + 1])
    for t in range(12):
        push(t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(args.pushes):
        push(t % n)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res = {'pipelined': args.pipelined, 'deterministic': args.deterministic, 'views': args.views, 'pushes': args.pushes, 'ms_per_push': round(dt / args.pushes * 1e3, 4),
           'fps_steady': round(args.pushes / dt, 1), 'canvas': [st.hc, st.wc]}
    res['graph_nodes'] = getattr(st, 'graph_nodes', None)
    g = getattr(st, 'graph', None)
    if g is not None:
        reps = max(20, args.pushes // 4)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        res['ms_per_replay_events'] = round(e0.elapsed_time(e1) / reps, 4)
        nodes = getattr(st, 'graph_nodes', None)
        if nodes is not None:
            res['graph_nodes'] = nodes
    print(res)


if __name__ == '__main__':
    main()
