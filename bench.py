#!/usr/bin/env python
"""Benchmark of the StabStitch++ inference hot path on MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

A "step" = one 2-view clip of `--frames` synthetic 720x1280 frames through the whole path (SpatialNet, TemporalNet x2,
tsmotion, sliding SmoothNet windows, canvas, TPS warp + AVERAGE fusion; warp NORMAL -- the defaults of the reference's
StabStitch-D script, test_online_ssd.py:440-444), inputs resident in HBM, outputs left in HBM.  One clip per rank
(independent video pairs: no data-path collective); a single all_gather of per-rank records at the end.
Prints ONE JSON line (see DESIGN.md "Measurement").
"""
import argparse
import json
import os
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))

PEAK_FP32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_HBM_GBS = 8000.0


def build_nets(dev):
    from stabstitch2_amd import synth
    from stabstitch2_amd.spatial_network import SpatialNet
    from stabstitch2_amd.temporal_network import TemporalNet
    from stabstitch2_amd.smooth_network import SmoothNet
    nets, sds = [], []
    for cls in (SpatialNet, TemporalNet, SmoothNet):
        m = cls()
        sd = synth.synthetic_state_dict(m)
        m.load_state_dict(sd, strict=True)
        nets.append(m.to(dev))
        sds.append(sd)
    return nets, sds


class ConvProbe:
    """HIP-event timing of every ss_conv_nhwc launch (the dominant kernel family) on the launch stream."""

    def __init__(self):
        self.records = []
        self.active = False

    def install(self):
        from stabstitch2_amd import ops
        orig = ops.conv
        probe = self

        def timed_conv(x, wgt, *a, **k):
            if not probe.active:
                return orig(x, wgt, *a, **k)
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            out = orig(x, wgt, *a, **k)
            e1.record()
            m = out.numel() // out.shape[-1]
            cout, kt, kh, kw, cin = wgt.shape
            # algorithmic MACs use the channels that carry data (zero-padded taps excluded)
            res = k.get('res')
            nbytes = 4 * (x.numel() + wgt.numel() + out.numel() + (res.numel() if res is not None else 0))
            probe.records.append((e0, e1, m, cout, kt * kh * kw, cin, nbytes))
            return out
        ops.conv = timed_conv
        orig_g = ops.conv_grouped

        def timed_conv_grouped(x, wgt, *a, **k):
            if not probe.active:
                return orig_g(x, wgt, *a, **k)
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            out = orig_g(x, wgt, *a, **k)
            e1.record()
            g, cout, kt, kh, kw, cin = wgt.shape
            m = out.numel() // out.shape[-1]              # rows of all groups together
            res = k.get('res')
            nbytes = 4 * (x.numel() + wgt.numel() + out.numel() + (res.numel() if res is not None else 0))
            probe.records.append((e0, e1, m, cout, kt * kh * kw, cin, nbytes))
            return out
        ops.conv_grouped = timed_conv_grouped
        from stabstitch2_amd import layers, smooth_network
        layers.ops = ops
        smooth_network.ops = ops

    def report(self):
        real_cin = {4: 3, 124: 121, 52: 49}
        agg = {}
        for e0, e1, m, cout, taps, cin, nbytes in self.records:
            c = real_cin.get(cin, cin)
            if cin == 4 and taps == 9:
                c = 2
            key = (m, cout, taps, cin)
            a = agg.setdefault(key, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += e0.elapsed_time(e1)
            a[2] += 2.0 * m * cout * taps * c
        import sys
        print('%10s %5s %5s %5s %4s %9s %9s %7s' % ('M', 'cout', 'taps', 'cin', 'n', 'ms', 'GFLOP', 'TF/s'), file=sys.stderr)
        for key, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            print('%10d %5d %5d %5d %4d %9.3f %9.2f %7.1f' % (key + (a[0], a[1], a[2] / 1e9, a[2] / a[1] / 1e9)),
                  file=sys.stderr)

    def summary(self):
        real_cin = {4: 3, 124: 121, 52: 49}
        tot_ms, tot_flop, n, tot_bytes = 0.0, 0.0, 0, 0.0
        for e0, e1, m, cout, taps, cin, nbytes in self.records:
            ms = e0.elapsed_time(e1)
            c = real_cin.get(cin, cin)
            if cin == 4 and taps == 9:
                c = 2      # CCL flow (dx, dy) regressor input
            tot_ms += ms
            tot_flop += 2.0 * m * cout * taps * c
            tot_bytes += nbytes
            n += 1
        return tot_ms, tot_flop, n, tot_bytes


def cpu_baseline(sds, frames, height, width, threads):
    """The CPU oracle (a from-scratch PyTorch-CPU port of the reference path) on a bounded sample of the workload."""
    from oracle import nets as ON, pipeline as OP
    from stabstitch2_amd import synth
    torch.set_num_threads(threads)
    nets = []
    for cls, sd in zip((ON.SpatialNet, ON.TemporalNet, ON.SmoothNet), sds):
        m = cls().eval()
        m.load_state_dict(sd, strict=True)
        nets.append(m)
    hr, lr = synth.make_clip_device(frames, height, width, seed=0, device='cpu')
    hr1 = [hr[0, i:i + 1] for i in range(frames)]
    hr2 = [hr[1, i:i + 1] for i in range(frames)]
    lr1 = [lr[0, i:i + 1] for i in range(frames)]
    lr2 = [lr[1, i:i + 1] for i in range(frames)]
    t0 = time.perf_counter()
    with torch.no_grad():
        acc = OP.estimate_meshes(nets, lr1, lr2)                        # SpatialNet, TemporalNet, tsmotion, SmoothNet
        t1 = time.perf_counter()
        fr, wc, hc = OP.get_stable_sqe(hr1, hr2, acc['smooth_mesh1'], acc['smooth_mesh2'], 'NORMAL', 'AVERAGE')
    t2 = time.perf_counter()
    out = (fr, int(hc), int(wc), acc['smooth_mesh1'], acc['smooth_mesh2'])
    return frames / (t2 - t0), out, {'estimate_meshes_s': round(t1 - t0, 2), 'warp_and_fuse_s': round(t2 - t1, 2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--frames', type=int, default=32)
    ap.add_argument('--height', type=int, default=720)
    ap.add_argument('--width', type=int, default=1280)
    ap.add_argument('--views', type=int, default=2, choices=(2, 3))
    ap.add_argument('--online', action='store_true', help='streaming mode: one frame pair per push (batch 1), fixed canvas')
    ap.add_argument('--warp_mode', default='NORMAL')
    ap.add_argument('--fusion_mode', default='AVERAGE')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-frames', type=int, default=48)
    ap.add_argument('--cpu-threads', type=int, default=32)
    ap.add_argument('--conv-report', action='store_true')
    ap.add_argument('--io', default='f32', choices=('f32', 'u8', 'u8host'),
                    help="f32: fp32 frames resident in HBM (the headline metric); u8: uint8 frames resident, ingest + "
                         "uint8 sink inside the step; u8host: uint8 frames in pinned host memory, H2D + D2H inside the step")
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node %d)'
                         % (args.gpus, world, args.gpus))
    assert torch.cuda.is_available(), 'bench.py needs an MI355X'
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=dev)       # RCCL over xGMI

    from stabstitch2_amd import synth, pipeline, _hip
    _hip.lib()
    torch.set_grad_enabled(False)
    nets, sds = build_nets(dev)
    hr, lr = synth.make_clip_device(args.frames, args.height, args.width, seed=rank, views=args.views, device=dev)
    probe = ConvProbe()
    probe.install()

    def step_online(use_graph=True):
        from stabstitch2_amd.online import OnlineStitcher
        st = OnlineStitcher(nets, args.height, args.width, warp_mode=args.warp_mode, fusion_mode=args.fusion_mode,
                            use_graph=use_graph)
        last = None
        for t in range(args.frames):
            got = st.push(hr[0][t:t + 1], hr[1][t:t + 1], lr[0][t:t + 1], lr[1][t:t + 1])
            if got:
                last = got[-1]
        return last.unsqueeze(0), st.hc, st.wc

    u8 = None
    runner = None
    if args.io != 'f32':
        assert args.views == 2 and not args.online, '--io u8 covers the offline 2-view path'
        u8 = [hr[v].permute(0, 2, 3, 1).round().clamp(0, 255).to(torch.uint8).contiguous() for v in range(2)]
        if args.io == 'u8host':
            u8 = [t.cpu().pin_memory() for t in u8]
            runner = pipeline.HostClipRunner(nets, dev, args.warp_mode, args.fusion_mode)

    def step_u8():
        fr, hc_, wc_, _, _ = pipeline.run_two_view_u8(u8[0], u8[1], nets, args.warp_mode, args.fusion_mode, device=dev)
        return fr, hc_, wc_

    def steps_host(k):
        """k clips through the overlapped upload / compute / download pipeline; -> last (video, Hc, Wc)."""
        last = None
        for last in runner.run((u8[0], u8[1]) for _ in range(k)):
            pass
        return last

    def step():
        if args.online:
            return step_online()
        if u8 is not None:
            return step_u8()
        if args.views == 3:       # BASELINE configs[4]: two 2-view passes (v1,v2),(v2,v3) + three-view composition
            return pipeline.run_three_view(hr[0], hr[1], hr[2], lr[0], lr[1], lr[2], nets, args.warp_mode,
                                           args.fusion_mode)[:3]
        return pipeline.run_two_view(hr[0], hr[1], lr[0], lr[1], nets, args.warp_mode, args.fusion_mode)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    if runner is not None:
        if args.warmup:
            steps_host(args.warmup)
    else:
        for _ in range(args.warmup):
            out = step()
    sync()
    t0 = time.perf_counter()
    if runner is not None:
        out = steps_host(args.steps)
    else:
        for i in range(args.steps):
            # HIP events around every conv launch of the last timed step (not inside a HIP-graph capture: the streaming
            # mode is probed on an extra eager pass after the timed region)
            probe.active = (i == args.steps - 1) and not args.online
            out = step()
    probe.active = False
    sync()
    dt = time.perf_counter() - t0
    if args.online:
        probe.active = True
        step_online(use_graph=False)
        probe.active = False
        torch.cuda.synchronize()
    frames_out, hc, wc = out[0], out[1], out[2]

    from stabstitch2_amd import dist as ssdist
    rec = torch.tensor([float(args.frames * args.steps), dt, float(hc), float(wc)], dtype=torch.float64)
    allrec = ssdist.gather_records(rec, dist, dev)            # the only collective: result gather
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    tmax = float(allrec[:, 1].max())
    fps = ssdist.aggregate_fps(allrec)

    conv_ms, conv_flop, conv_n, conv_bytes = probe.summary()
    if args.conv_report:
        probe.report()
    achieved = conv_flop / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
    # algorithmic HBM bytes per stitched frame: fp32 frames in, fp32 canvas out, weights once
    io_bytes = args.views * (3 * args.height * args.width + 3 * 360 * 480) * 4 + 3 * hc * wc * 4 + 70.6e6
    # HBM bytes per conv launch from the committed PMC passes of this same command (profiles/r01_pmc_hbm.json;
    # FETCH_SIZE / WRITE_SIZE cannot be read live from inside the process)
    traffic = None
    try:
        with open(os.path.join(ROOT, 'profiles', 'r01_pmc_hbm.json')) as f:
            traffic = json.load(f)['kernels']['conv_igemm_kernel']['hbm_bytes_per_launch']
    except Exception:
        pass
    result = {
        'metric': 'stitched frames/sec, %dp %d-view (StabStitch++ inference hot path)' % (args.height, args.views),
        'value': round(fps, 3), 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(tmax / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': '%s: %dx%d %d-view, %d-frame clip per step per GPU, 7-frame SmoothWarp sliding '
                               'window, warp %s / fusion %s, synthetic checkpoints' % (
                                   ('streaming (batch 1) ' if args.online else '') + ('' if args.io == 'f32' else '[io=%s] ' % args.io) +
                                   ('configs[4]' if args.views == 3 else ('configs[2]' if args.height == 720 else 'configs[1]')),
                                   args.height, args.width, args.views, args.frames, args.warp_mode, args.fusion_mode),
                   'frames_per_step': args.frames, 'canvas': [int(hc), int(wc)], 'parallelism': 'streams%d' % world,
                   'published_reference': '28.3 fps on 1x RTX 4090 at 360x480 (README.md:30); different resolution '
                                          'and hardware, not comparable'},
        'roofline': {'bound': 'mfma', 'kernel': 'conv_igemm_kernel<WM,WN> (fp32 implicit-GEMM conv, %d launches/clip)'
                     % conv_n, 'achieved': round(achieved, 3), 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                     'frac': round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), 'traffic': traffic,
                     'traffic_unit': 'HBM bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, profiles/r01_pmc_hbm.json)',
                     'algorithmic_flop_per_launch': round(conv_flop / max(conv_n, 1)),
                     'algorithmic_bytes_per_launch': round(conv_bytes / max(conv_n, 1)),
                     'avg_launch_us': round(conv_ms * 1e3 / max(conv_n, 1), 2),
                     'kernel_ms_per_step': round(conv_ms, 3),
                     'path_hbm_frac': round(fps / world * io_bytes / 1e9 / PEAK_HBM_GBS, 5),
                     # whole path against the MFMA roof: 41.31 GFLOP of dense contraction per 2-view frame
                     'path_mfma_frac': round(fps / world * 41.31e9 / (PEAK_FP32_MFMA_TFLOPS * 1e12), 4) if args.views == 2 else None},
    }
    if world == 1 and not args.no_cpu_baseline and args.views == 2 and not args.online:
        threads = max(1, min(args.cpu_threads, os.cpu_count()))
        cfps, cout, csplit = cpu_baseline(sds, args.cpu_frames, args.height, args.width, threads)
        result['cpu_baseline'] = {'value': round(cfps, 4), 'unit': 'frames/s', 'cores': threads, 'kind': 'port',
                                  'sample': '%d-frame %dx%d 2-view clip (seed 0), NORMAL/AVERAGE, oracle/ on PyTorch-CPU'
                                            % (args.cpu_frames, args.height, args.width)}
        # parity at benchmark time: same clip through the HIP path
        n = args.cpu_frames
        result['cpu_baseline']['host_logical_cpus'] = os.cpu_count()
        result['cpu_baseline']['stage_seconds'] = csplit
        hr0, lr0 = synth.make_clip_device(n, args.height, args.width, seed=0, device=dev)
        g = pipeline.run_two_view(hr0[0], hr0[1], lr0[0], lr0[1], nets, args.warp_mode, args.fusion_mode)
        dm = max(float((g[3].cpu() - cout[3]).abs().max()), float((g[4].cpu() - cout[4]).abs().max()))
        par = {'mesh_max_abs_px': round(dm, 6), 'canvas_equal': (g[1], g[2]) == (cout[1], cout[2])}
        if par['canvas_equal']:
            import numpy as np
            d = np.abs(g[0][0].permute(1, 2, 0).cpu().numpy() - cout[0][0])
            par['frame0_median_abs'] = round(float(np.median(d)), 6)
            par['frame0_p999_abs'] = round(float(np.quantile(d, 0.999)), 6)
        # alignment PSNR / SSIM (test_metric_ssd.py:513-527) of the first frames from both sides, each with its own meshes
        from oracle import metrics as OM
        from stabstitch2_amd import metrics as GM
        k = min(4, n)
        lr_cpu = lr0.cpu()              # same frames as the meshes were estimated from
        c1 = OM.warp_lr_with_mask([lr_cpu[0, i:i + 1] for i in range(k)], cout[3][:, :k])
        c2 = OM.warp_lr_with_mask([lr_cpu[1, i:i + 1] for i in range(k)], cout[4][:, :k])
        cps = [OM.alignment_psnr_ssim(a, b) for a, b in zip(c1, c2)]
        gp, gs = GM.alignment_psnr_ssim(GM.warp_lr_planes(lr0[0][:k], g[3][:, :k]), GM.warp_lr_planes(lr0[1][:k], g[4][:, :k]))
        par['alignment_psnr_delta_db'] = round(max(abs(float(gp[i]) - cps[i][0]) for i in range(k)), 5)
        par['alignment_ssim_delta'] = round(max(abs(float(gs[i]) - cps[i][1]) for i in range(k)), 6)
        result['parity_vs_cpu'] = par
    print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
