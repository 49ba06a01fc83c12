#!/usr/bin/env python
"""Benchmark of the StabStitch++ inference hot path on MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python bench.py --gpus 8 --steps 5 --warmup 2          # self-launches 8 ranks (one per GPU) over RCCL
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

A "step" = one 2-view clip of `--frames` synthetic 720x1280 frames through the whole path (SpatialNet, TemporalNet x2,
tsmotion, sliding SmoothNet windows, canvas, TPS warp + AVERAGE fusion; warp NORMAL -- the defaults of the reference's
StabStitch-D script, test_online_ssd.py:440-444), inputs resident in HBM, outputs left in HBM.  One clip per rank
(independent video pairs, seed = rank: no data-path collective); a single all_gather of per-rank records at the end.
Prints ONE JSON line (see DESIGN.md "Measurement").  At N=1 the line also carries the other BASELINE.json
configurations (`other_configs`), the CPU oracle baseline with its per-stage split, and benchmark-time parity.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

# the HIP runtime multiplexes all streams of a process onto GPU_MAX_HW_QUEUES hardware queues (default 4); streams sharing a
# queue execute in order, which serialises the host->host path's PCIe copies with its kernels (stabstitch2_amd/__init__.py,
# tools/diag_overlap.py).  Must be in the environment before the runtime initialises, i.e. before the first device call.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))

PEAK_FP32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_HBM_GBS = 8000.0
PMC_PROFILE = os.path.join('profiles', 'r06_pmc_hbm.json')


def build_nets(dev, profile='default'):
    """The three networks on `dev`: real checkpoints when the reference's `Full_model_inference/full_model_{tra,ssd}/`
    (or $SS_MODEL_DIR) holds the three *.pth files (test_online_tra.py:173-194), else the deterministic synthetic
    checkpoints of stabstitch2_amd/synth.py (`--weights trained_like`: its harsh profile -- BN-folded channel scales over four
    decades, Student-t taps; pinned against the reference by tests/golden/g14_trained_like.npz).  -> (nets, state_dicts)."""
    from stabstitch2_amd import synth, pipeline
    from stabstitch2_amd.spatial_network import SpatialNet
    from stabstitch2_amd.temporal_network import TemporalNet
    from stabstitch2_amd.smooth_network import SmoothNet
    model_dir = os.environ.get('SS_MODEL_DIR') or pipeline.find_model_dir(ROOT)
    if model_dir:
        nets = pipeline.load_nets(model_dir, dev)
        build_nets.weights = 'pretrained (%s)' % model_dir
        return list(nets), [{k: v.detach().cpu() for k, v in m.state_dict().items()} for m in nets]
    build_nets.weights = 'synthetic checkpoints' + ('' if profile == 'default' else " (profile '%s')" % profile)
    nets, sds = [], []
    for cls in (SpatialNet, TemporalNet, SmoothNet):
        m = cls()
        sd = synth.synthetic_state_dict(m, profile=profile)
        m.load_state_dict(sd, strict=True)
        nets.append(m.to(dev))
        sds.append(sd)
    return nets, sds


build_nets.weights = 'synthetic checkpoints'


def _measured_limiters():
    """Kernels of the `secondary` list that are NOT HBM-bound although SURVEY.md 8d prices them against HBM: what the committed
    PMC passes (profiles/r06_pmc_render.json, profiles/r06_pmc_lds.json) measured instead.  Static, like the HBM bytes."""
    out = {}
    try:
        with open(os.path.join(ROOT, 'profiles', 'r06_pmc_render.json')) as f:
            r = json.load(f)
        out['render_average_kernel'] = {'unit': 'valu', 'valu_issue_active_frac': r['valu_active_frac'],
                                        'source': 'profiles/r06_pmc_render.json (SQ_ACTIVE_INST_VALU)'}
    except (OSError, KeyError, ValueError):
        pass
    try:
        with open(os.path.join(ROOT, 'profiles', 'r06_pmc_lds.json')) as f:
            l = json.load(f)['cost_volume']
        out['cost_volume_kernel'] = {'unit': 'valu issue (packed fp32 FMA + operand moves)', 'lds_active_frac': l['lds_busy_frac'],
                                     'lds_bank_conflict_frac': l['lds_bank_conflict_frac'],
                                     'source': 'profiles/r06_pmc_lds.json (SQ_LDS_IDX_ACTIVE, SQ_LDS_BANK_CONFLICT)'}
        with open(os.path.join(ROOT, 'profiles', 'r06_pmc_mfma.json')) as f:
            m = json.load(f)['cost_volume']['counters']
        out['cost_volume_kernel']['valu_issue_frac'] = round(m['SQ_INSTS_VALU'] * 4 / 1024.0 / (m['GRBM_GUI_ACTIVE'] / 8.0), 3)
    except (OSError, KeyError, ValueError):
        pass
    return out


class ConvProbe:
    """HIP-event timing of every conv-engine launch (the dominant kernel family) on the launch stream."""

    def __init__(self):
        self.records = []
        self.secondary = {}
        self.active = False

    def secondary_report(self, pmc):
        """-> {kernel: launches, ms per step, avg us (HIP events of the probed step); + HBM bytes per launch from the
        committed PMC passes, achieved GB/s and fraction of the 8 TB/s roof when the profile lists the kernel}."""
        out = {}
        for k, evs in self.secondary.items():
            ms = sum(a.elapsed_time(b) for a, b in evs)
            ent = {'launches_per_step': len(evs), 'ms_per_step': round(ms, 3), 'avg_launch_us': round(ms * 1e3 / len(evs), 2),
                   'bound': 'hbm'}
            pk = (pmc or {}).get('kernels', {}).get(k)
            if pk and pk.get('launches'):
                # bytes per STEP from the profile (its launches / its steps), so that a kernel whose launch count per step
                # changes between the profile and this run is not mispriced
                steps = max(pmc.get('steps_profiled', 0), 1)
                per_step = pk['hbm_bytes_per_launch'] * pk['launches'] / steps
                ent['hbm_bytes_per_step_static'] = round(per_step)
                ent['achieved_GBps'] = round(per_step / (ms * 1e-3) / 1e9, 1) if ms > 0 else 0.0
                ent['frac_of_hbm_peak'] = round(per_step / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4) if ms > 0 else 0.0
            lim = _measured_limiters().get(k)
            if lim:
                ent['measured_limiter_static'] = lim      # what the committed PMC passes say the kernel is bound by (not HBM)
            out[k] = ent
        return out

    def install(self):
        from stabstitch2_amd import ops
        probe = self

        def wrap(orig, grouped):
            def timed(x, wgt, *a, **k):
                if not probe.active:
                    return orig(x, wgt, *a, **k)
                if k.get('pool2') and not (ops.pool2_is_fused(x, wgt, k.get('stride', 1), k.get('pad', (0, 1, 1))) or
                                           ops.pool2_in_reduce(x, wgt, k.get('stride', 1), k.get('pad', (0, 1, 1)))):
                    return orig(x, wgt, *a, **k)      # two launches: the inner conv and the pool are probed on their own
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                out = orig(x, wgt, *a, **k)
                e1.record()
                cout, kt, kh, kw, cin = wgt.shape[-5:]
                m = out.numel() // out.shape[-1]              # rows (of all groups together)
                if k.get('pool2'):                            # conv + 2x2 max-pool in one kernel: the conv's rows, not the pooled map's
                    groups = wgt.shape[0] if wgt.dim() == 6 else 1
                    m = groups * x.shape[-4] * x.shape[-3] * x.shape[-2]        # 3x3 / stride 1 / pad 1: one output row per input pixel
                res = k.get('res')
                nbytes = 4 * (x.numel() + wgt.numel() + out.numel() + (res.numel() if res is not None else 0))
                stride = k.get('stride', 1)
                # executed / direct-convolution flop of the kernel the engine actually launched (ops records its dispatch)
                probe.records.append((e0, e1, m, cout, kt * kh * kw, cin, nbytes, {'wino': 16.0 / 36.0, 'wino43': 36.0 / 144.0}.get(ops.last_conv_path, 1.0)))
                return out
            return timed
        ops.conv = wrap(ops.conv, False)
        ops.conv_grouped = wrap(ops.conv_grouped, True)
        orig_stem = ops.conv_stem

        def timed_stem(buf, wgt, *a, **k):          # 7x7/2 stem on the row-packed 3-channel layout: 147 real products per output
            if not probe.active:
                return orig_stem(buf, wgt, *a, **k)
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            out = orig_stem(buf, wgt, *a, **k)
            e1.record()
            m = out.numel() // out.shape[-1]
            nbytes = 4 * (buf.numel() + wgt.numel() + out.numel())
            probe.records.append((e0, e1, m, wgt.shape[-3], 49, 3, nbytes, 1.0))
            return out
        ops.conv_stem = timed_stem
        orig_fused = ops.stem_pool

        def timed_fused(buf, wgt, *a, **k):         # the fused stem (conv 7x7/2 + BN + ReLU + max-pool, csrc/stem.hip): its conv's 147-tap flop
            if not probe.active:
                return orig_fused(buf, wgt, *a, **k)
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            out = orig_fused(buf, wgt, *a, **k)
            e1.record()
            n, h, wp8, _ = buf.shape
            ho, wo = (h - 1) // 2 + 1, (wp8 - 8 - 1) // 2 + 1
            cout = wgt.numel() // 168
            nbytes = 4 * (buf.numel() + wgt.numel() + out.numel())
            probe.records.append((e0, e1, n * ho * wo, cout, 49, 3, nbytes, 1.0, 'stem_pool_kernel'))
            return out
        ops.stem_pool = timed_fused
        # the HBM-side kernels SURVEY.md 8d judges against memory bandwidth (K7 homography sampler, K8 cost volume, max-pool,
        # K12/K13 fused render): HIP events around their launches in the same step
        def wrap_plain(name, orig):
            def timed(*a, **k):
                if not probe.active:
                    return orig(*a, **k)
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                out = orig(*a, **k)
                e1.record()
                probe.secondary.setdefault(name, []).append((e0, e1))
                return out
            return timed
        for fn, kern in (('render_average_clip', 'render_average_kernel'), ('render_average', 'render_average_kernel'),
                         ('cost_volume', 'cost_volume_kernel'), ('cost_volume_bidir', 'cost_volume_kernel'), ('maxpool', 'maxpool_kernel'), ('maxpool_split', 'maxpool_kernel'),
                         ('homo_warp_nhwc', 'homo_warp_kernel')):
            setattr(ops, fn, wrap_plain(kern, getattr(ops, fn)))

    @staticmethod
    def _real_cin(cin, taps):
        # algorithmic MACs use the channels that carry data (zero-padded taps excluded)
        if cin == 4 and taps == 9:
            return 2          # CCL flow (dx, dy) regressor input
        return {4: 3, 124: 121, 52: 49}.get(cin, cin)          # (the stem reports its 3 real channels itself)

    def report(self):
        agg = {}
        for e0, e1, m, cout, taps, cin, nbytes, xr, *_ in self.records:
            key = (m, cout, taps, cin)
            a = agg.setdefault(key, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += e0.elapsed_time(e1)
            a[2] += 2.0 * m * cout * taps * self._real_cin(cin, taps)
        print('%10s %5s %5s %5s %4s %9s %9s %7s' % ('M', 'cout', 'taps', 'cin', 'n', 'ms', 'GFLOP', 'TF/s'), file=sys.stderr)
        for key, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            print('%10d %5d %5d %5d %4d %9.3f %9.2f %7.1f' % (key + (a[0], a[1], a[2] / 1e9, a[2] / a[1] / 1e9)),
                  file=sys.stderr)

    def by_kernel(self):
        """-> {kernel family: (launches, ms, executed flop, direct-equivalent flop)}: 'conv_wino_kernel' = the launches the
        engine's dispatch rule sends to the Winograd F(2x2,3x3) kernel, 'conv_wino43_kernel' = those it sends to F(4x4,3x3),
        'conv_igemm_kernel' = the rest."""
        out = {}
        for e0, e1, m, cout, taps, cin, nbytes, xr, *name in self.records:
            k = name[0] if name else ('conv_wino43_kernel' if xr < 0.3 else ('conv_wino_kernel' if xr < 1.0 else 'conv_igemm_kernel'))
            a = out.setdefault(k, [0, 0.0, 0.0, 0.0])
            f = 2.0 * m * cout * taps * self._real_cin(cin, taps)
            a[0] += 1; a[1] += e0.elapsed_time(e1); a[2] += f * xr; a[3] += f
        return out

    def summary(self):
        """-> (ms, direct-conv-equivalent flop, launches, algorithmic bytes, executed MFMA flop)."""
        tot_ms, tot_flop, n, tot_bytes, tot_exec = 0.0, 0.0, 0, 0.0, 0.0
        for e0, e1, m, cout, taps, cin, nbytes, xr, *_ in self.records:
            tot_ms += e0.elapsed_time(e1)
            f = 2.0 * m * cout * taps * self._real_cin(cin, taps)
            tot_flop += f
            tot_exec += f * xr
            tot_bytes += nbytes
            n += 1
        return tot_ms, tot_flop, n, tot_bytes, tot_exec


def cpu_baseline(sds, frames, height, width, threads):
    """The CPU oracle (a from-scratch PyTorch-CPU port of the reference path) on a bounded sample of the workload,
    with the per-stage split BASELINE.md 3 names (spatial / temporal / tsmotion / smooth / warp+blend)."""
    from oracle import nets as ON, pipeline as OP
    from stabstitch2_amd import synth
    torch.set_num_threads(threads)
    nets = []
    for cls, sd in zip((ON.SpatialNet, ON.TemporalNet, ON.SmoothNet), sds):
        m = cls().eval()
        m.load_state_dict(sd, strict=True)
        nets.append(m)
    hr, lr = synth.make_clip_device(frames, height, width, seed=0, device='cpu')
    sl = lambda t: [t[i:i + 1] for i in range(frames)]
    hr1, hr2, lr1, lr2 = sl(hr[0]), sl(hr[1]), sl(lr[0]), sl(lr[1])
    st = {}
    with torch.no_grad():
        t0 = time.perf_counter()
        s1, s2 = OP.spatial_stage(nets[0], lr1, lr2)
        t1 = time.perf_counter()
        tm1, tm2 = OP.temporal_stage(nets[1], lr1), OP.temporal_stage(nets[1], lr2)
        t2 = time.perf_counter()
        smesh1, tsm1 = OP.tsmotion_prepare(s1, tm1)
        smesh2, tsm2 = OP.tsmotion_prepare(s2, tm2)
        t3 = time.perf_counter()
        acc = OP.smooth_stage(nets[2], tsm1, tsm2, smesh1, smesh2)
        t4 = time.perf_counter()
        fr, wc, hc = OP.get_stable_sqe(hr1, hr2, acc['smooth_mesh1'], acc['smooth_mesh2'], 'NORMAL', 'AVERAGE')
        t5 = time.perf_counter()
    st = {'spatial_s': round(t1 - t0, 2), 'temporal_s': round(t2 - t1, 2), 'tsmotion_s': round(t3 - t2, 2),
          'smooth_s': round(t4 - t3, 2), 'warp_blend_s': round(t5 - t4, 2)}
    out = (fr, int(hc), int(wc), acc['smooth_mesh1'], acc['smooth_mesh2'])
    return frames / (t5 - t0), out, st


def cpu_thread_sweep(sds, height, width, candidates):
    """Seconds for one SpatialNet pair + one two-image TPS warp per thread count; -> (best, {threads: seconds}).
    BASELINE.md 3 names os.cpu_count() threads; on the 2 x 64-core GPU hosts fewer threads are faster for these
    small-batch CPU convolutions, so the sample runs with the fastest of the candidates and the sweep is reported."""
    from oracle import nets as ON, samplers as OS, geometry as OG
    from stabstitch2_amd import synth
    sp = ON.SpatialNet().eval()
    sp.load_state_dict(sds[0], strict=True)
    hr, lr = synth.make_clip_device(1, height, width, seed=0, device='cpu')
    nr = OG.norm_mesh(OG.rigid_mesh(1, height, width), height, width)
    src = torch.cat((nr, nr), 0)
    img = torch.cat((hr[0, 0:1], hr[1, 0:1]), 0)
    res = {}
    x = torch.randn(1, 64, 90, 120)
    w = torch.randn(64, 64, 3, 3)
    with torch.no_grad():
        for th in candidates:
            torch.set_num_threads(th)
            if th > 64:
                # every hardware thread of a 2 x 64-core host: measured 136 s for the unit below (oversubscribed
                # small-batch convolutions), so this candidate is probed on ONE layer1 convolution and scaled
                torch.nn.functional.conv2d(x, w, padding=1)
                t0 = time.perf_counter()
                for _ in range(3):
                    torch.nn.functional.conv2d(x, w, padding=1)
                res['%d (one 3x3 64->64 conv at 90x120, x3)' % th] = round(time.perf_counter() - t0, 3)
                torch.set_num_threads(min(candidates))
                t0 = time.perf_counter()
                for _ in range(3):
                    torch.nn.functional.conv2d(x, w, padding=1)
                res['%d (same conv probe)' % min(candidates)] = round(time.perf_counter() - t0, 3)
                continue
            ON.build_SpatialNet(sp, lr[0, 0:1], lr[1, 0:1])            # warm
            t0 = time.perf_counter()
            ON.build_SpatialNet(sp, lr[0, 0:1], lr[1, 0:1])
            OS.tps_warp(img, src, src, (height, width), 'NORMAL')
            res[th] = round(time.perf_counter() - t0, 3)
    full = {k: v for k, v in res.items() if isinstance(k, int)}
    return min(full, key=full.get), res


def path_traffic(pmc, algorithmic_bytes_per_step):
    """Static: HBM bytes per clip summed over every kernel the PMC passes list (conv engine, render, cost volume, pools, FC,
    homography sampler) against the compulsory bytes of SURVEY.md 8d (frames in, canvas out, weights once)."""
    if not pmc or not pmc.get('kernels'):
        return None
    steps = max(pmc.get('steps_profiled', 0), 1)
    tot = sum(v['hbm_bytes_per_launch'] * v['launches'] for v in pmc['kernels'].values()) / steps
    return {'hbm_bytes_per_step_static': round(tot), 'algorithmic_bytes_per_step': round(algorithmic_bytes_per_step),
            'ratio': round(tot / algorithmic_bytes_per_step, 2), 'source': 'static: ' + PMC_PROFILE}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--frames', type=int, default=32)
    ap.add_argument('--height', type=int, default=720)
    ap.add_argument('--width', type=int, default=1280)
    ap.add_argument('--views', type=int, default=2, choices=(2, 3))
    ap.add_argument('--online', action='store_true', help='streaming mode: one frame pair per push (batch 1), fixed canvas')
    ap.add_argument('--warp_mode', default='NORMAL')
    ap.add_argument('--fusion_mode', default='AVERAGE')
    ap.add_argument('--weights', default='default', choices=('default', 'trained_like'),
                    help="synthetic checkpoint profile (stabstitch2_amd/synth.py); trained_like = the conv engine's adversary")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-other-configs', action='store_true', help='skip the other BASELINE.json configurations')
    ap.add_argument('--cpu-frames', type=int, default=24)
    ap.add_argument('--cpu-threads', type=int, default=0, help='0 = short sweep over {cpu_count, 64, 32, 16}')
    ap.add_argument('--conv-report', action='store_true')
    ap.add_argument('--io', default='f32', choices=('f32', 'u8', 'u8host'),
                    help="f32: fp32 frames resident in HBM (the headline metric); u8: uint8 frames resident, ingest + "
                         "uint8 sink inside the step; u8host: uint8 frames in pinned host memory, H2D + D2H inside the step")
    ap.add_argument('--backend', default='nccl', choices=('nccl', 'gloo'), help='nccl = RCCL over xGMI; gloo for the CPU launcher test')
    ap.add_argument('--force-collective', action='store_true',
                    help='initialise the process group and run the result all_gather even with ONE rank (RCCL smoke on a 1-GPU box)')
    ap.add_argument('--share-device', action='store_true',
                    help='multi-rank readiness check on a 1-GPU box: every rank drives cuda:0 (real kernels, per-rank clip seeds, '
                         'the gather and the N > 1 JSON line; use with --backend gloo -- RCCL refuses two ranks on one device)')
    ap.add_argument('--also-360', action='store_true',
                    help='N > 1: append the 360x480 64-frame configuration (configs[1]) measured on ALL ranks (north_star: both '
                         'resolutions at 1 / 2 / 4 / 8 GPUs); N = 1 has it under other_configs already')
    ap.add_argument('--stub-step-ms', type=float, default=0.0,
                    help='launcher self-test without GPUs: every step is a sleep of this many ms (backend gloo)')
    return ap.parse_args(argv)


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks (one process per GPU) under torch.distributed.run on
    this node and hand them the same command line; rank 0 prints the JSON line."""
    if not args.stub_step_ms:
        assert torch.cuda.is_available(), 'bench.py needs MI355X GPUs'
        have = torch.cuda.device_count()
        if have < args.gpus and not args.share_device:
            raise SystemExit('--gpus %d but only %d GPU(s) visible on this node' % (args.gpus, have))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: what RCCL needs on this driver
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // max(args.gpus, 1))))
    return subprocess.run(cmd, env=env).returncode


def measure(step, sync, warmup, steps, per_step=False):
    """warm-up, then `steps` timed calls bracketed by sync(); -> (seconds, last result[, per-step seconds]).
    per_step=True appends a second pass of `steps` calls with a sync after each one (median / min of single steps:
    a one-off -- a first-touch allocation, a late compile -- shows up there and not in a 3-step average)."""
    out = None
    for _ in range(warmup):
        out = step()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    sync()
    dt = time.perf_counter() - t0
    if not per_step:
        return dt, out
    single = []
    for _ in range(steps):
        t1 = time.perf_counter()
        out = step()
        sync()
        single.append(time.perf_counter() - t1)
    return dt, out, single


def other_configs(nets, dev, args):
    """The remaining BASELINE.json configurations and I/O variants on this GPU (same build, same process), each with 2+
    warm-up steps (the caching allocator then holds both result buffers a `out = step()` loop alternates between: the
    first-touch hipMalloc of a 710 MB three-view canvas inside a 3-step timed region was round 2's 24 ms outlier) and 10
    timed steps: fps over the 10, plus median / min of single synchronised steps.  The headline `value` stays configs[2]."""
    from stabstitch2_amd import synth, pipeline
    from stabstitch2_amd.online import OnlineStitcher
    res = {}
    sync = torch.cuda.synchronize
    K, W = 10, 2

    def entry(name, frames_per_step, seconds, steps, hc, wc, single=None, note=None):
        res[name] = {'fps': round(frames_per_step * steps / seconds, 1), 'ms_per_step': round(seconds / steps * 1e3, 3),
                     'frames_per_step': frames_per_step, 'steps': steps, 'canvas': [int(hc), int(wc)]}
        if single:
            ss = sorted(single)
            res[name]['ms_per_step_median'] = round(ss[len(ss) // 2] * 1e3, 3)
            res[name]['ms_per_step_min'] = round(ss[0] * 1e3, 3)
            res[name]['ms_per_step_max'] = round(ss[-1] * 1e3, 3)
        if note:
            res[name]['note'] = note

    # configs[1]: 360x480 2-view, 64-frame clip
    hr, lr = synth.make_clip_device(64, 360, 480, seed=0, device=dev)
    dt, o, sg = measure(lambda: pipeline.run_two_view(hr[0], hr[1], lr[0], lr[1], nets), sync, W, K, True)
    entry('configs[1] 360x480 2-view 64-frame clip', 64, dt, K, o[1], o[2], sg)
    del o
    # 720p clip shared by the 720p variants
    n = args.frames
    hr, lr = synth.make_clip_device(n, 720, 1280, seed=0, views=3, device=dev)
    # configs[4]: 3-view 720p (two 2-view passes + composition + 3-image render)
    dt, o, sg = measure(lambda: pipeline.run_three_view(hr[0], hr[1], hr[2], lr[0], lr[1], lr[2], nets), sync, W, K, True)
    entry('configs[4] 720p 3-view', n, dt, K, o[1], o[2], sg)
    del o
    # fusion LINEAR (default of test_online_tra.py), warp FAST
    dt, o, sg = measure(lambda: pipeline.run_two_view(hr[0], hr[1], lr[0], lr[1], nets, 'NORMAL', 'LINEAR'), sync, W, K, True)
    entry('720p 2-view fusion LINEAR', n, dt, K, o[1], o[2], sg)
    dt, o, sg = measure(lambda: pipeline.run_two_view(hr[0], hr[1], lr[0], lr[1], nets, 'FAST', 'AVERAGE'), sync, W, K, True)
    entry('720p 2-view warp FAST', n, dt, K, o[1], o[2], sg)
    # three views with the reference's default fusion (test_online_tra_threeview.py:541): warp once, two chained blend passes
    dt, o, sg = measure(lambda: pipeline.run_three_view(hr[0], hr[1], hr[2], lr[0], lr[1], lr[2], nets, 'NORMAL', 'LINEAR'),
                        sync, W, K, True)
    entry('720p 3-view fusion LINEAR', n, dt, K, o[1], o[2], sg)
    del o
    # opt-in: TemporalNet's trunk on a second HIP stream beside SpatialNet's chain of small launches (pipeline.QUAD_OVERLAP)
    old_ov = pipeline.QUAD_OVERLAP
    pipeline.QUAD_OVERLAP = True
    try:
        dt, o, sg = measure(lambda: pipeline.run_two_view(hr[0], hr[1], lr[0], lr[1], nets), sync, W, K, True)
        entry('720p 2-view, TemporalNet trunk beside the SpatialNet regressor chain on a second stream (opt-in SS_QUAD_OVERLAP=1)',
              n, dt, K, o[1], o[2], sg, 'bit-identical frames; per-launch durations overlap, so the headline (and its roofline) run without it')
    finally:
        pipeline.QUAD_OVERLAP = old_ov
    del o
    # round 6 opt-ins (never the headline): the geometry-only kernel policy (a frame's bits independent of its batch), the one-block-
    # per-workgroup F(4x4,3x3) kernel (A/B of the persistent one), the render with the reference's + 1e-6 folded into its row table
    from stabstitch2_amd import ops as _ops
    dt, o, sg = measure(lambda: pipeline.run_two_view(hr[0], hr[1], lr[0], lr[1], nets, deterministic=True), sync, W, K, True)
    entry('720p 2-view, deterministic kernel policy (opt-in: deterministic=True)', n, dt, K, o[1], o[2], sg,
          note='every kernel chosen by layer geometry alone: resident clip == chunked passes == stream, bit for bit')
    del o
    _ops.WINO43_PERSIST = False
    try:
        dt, o, sg = measure(lambda: pipeline.run_two_view(hr[0], hr[1], lr[0], lr[1], nets), sync, W, K, True)
        entry('720p 2-view, F(4x4,3x3) with one workgroup per tile block (A/B: SS_WINO43_PERSIST=0)', n, dt, K, o[1], o[2], sg,
              note='the round-5 kernel; the headline runs the persistent one (bit-identical)')
    finally:
        _ops.WINO43_PERSIST = True
    del o
    _ops.RENDER_EPS_FOLD = True
    try:
        dt, o, sg = measure(lambda: pipeline.run_two_view(hr[0], hr[1], lr[0], lr[1], nets), sync, W, K, True)
        entry('720p 2-view, render with the + 1e-6 folded into the row table (opt-in SS_RENDER_EPS_FOLD=1)', n, dt, K, o[1], o[2], sg,
              note='NOT the reference arithmetic (a log a instead of d2 log(d2 + 1e-6), ~1e-3 px): passes the G7 / G9 / G13 gates '
                   '(tests/test_gpu_round6.py), never the default')
    finally:
        _ops.RENDER_EPS_FOLD = False
    del o
    # opt-in arithmetic of the Winograd GEMMs: fp32 products formed exactly from three bf16 slices per operand (nine slice
    # products) on the bf16 matrix pipe, fp32 accumulation (ops.WINO_MATH, csrc/wino.hip SLICED).  NOT the headline.
    from stabstitch2_amd import ops
    base = pipeline.run_two_view(hr[0], hr[1], lr[0], lr[1], nets)
    old_math = ops.WINO_MATH
    ops.WINO_MATH = 'bf16x9'
    try:
        dt, o, sg = measure(lambda: pipeline.run_two_view(hr[0], hr[1], lr[0], lr[1], nets), sync, W, K, True)
        dm = max(float((o[3] - base[3]).abs().max()), float((o[4] - base[4]).abs().max()))
        entry('720p 2-view, Winograd GEMMs as exact bf16x9 slice products (opt-in SS_WINO_MATH=bf16x9)', n, dt, K, o[1], o[2], sg,
              'every fp32 x fp32 product from 3 bf16 slices per operand, all 9 slice products, fp32 accumulation; smooth meshes '
              'differ from the fp32-MFMA path by %.1e px (max); canvas %s' % (dm, 'equal' if (o[1], o[2]) == (base[1], base[2]) else 'DIFFERENT'))
    finally:
        ops.WINO_MATH = old_math
    del o, base
    # host-to-host: uint8 frames in pinned host memory -> stitched uint8 frames in pinned host memory (H2D + ingest +
    # path + uint8 sink + D2H of every fused frame, as the reference's printed fps includes .cpu())
    u8 = [hr[v].permute(0, 2, 3, 1).round().clamp(0, 255).to(torch.uint8).contiguous().cpu().pin_memory() for v in range(2)]
    from stabstitch2_amd import hostbind
    place = hostbind.report(dev) or {}
    for fusion in ('AVERAGE', 'LINEAR'):
        runner = pipeline.HostClipRunner(nets, dev, fusion_mode=fusion)

        stamps = []

        def host_steps(k):
            last = None
            for last in runner.run((u8[0], u8[1]) for _ in range(k)):
                stamps.append(time.perf_counter())
            return last
        host_steps(3)
        sync()
        runner.timed = True
        KH = 3 * K                      # the first upload and the last download of a run are not hidden: a long run
        del stamps[:]
        t0 = time.perf_counter()
        last = host_steps(KH)
        sync()
        name = '720p 2-view uint8 host->host incl. D2H of every fused frame' + ('' if fusion == 'AVERAGE' else ', fusion LINEAR')
        entry(name, n, time.perf_counter() - t0, KH, last[1], last[2],
              note='PCIe both ways (5.5 MB in + 3.1 MB out per frame), copies overlapped with compute on three HIP streams; '
                   'the clips of one run are pipelined: fps includes the un-hidden first upload and last download of the run, '
                   'ms_per_clip_steady = median time between two delivered clips; h2d / d2h = HIP events around the '
                   'copies on their own streams while the compute stream runs the neighbouring clip')
        gaps = sorted(b - a for a, b in zip(stamps[:-2], stamps[1:-1]))
        res[name]['ms_per_clip_steady'] = round(gaps[len(gaps) // 2] * 1e3, 3)
        res[name]['fps_steady'] = round(n / gaps[len(gaps) // 2], 1)
        res[name].update(runner.copy_stats())
        res[name]['numa_node'] = place.get('numa_node')
        res[name]['cpus_bound'] = place.get('cpus_bound')
        res[name]['GPU_MAX_HW_QUEUES'] = os.environ.get('GPU_MAX_HW_QUEUES')
        if fusion == 'AVERAGE':
            # the same copies with nothing else on the GPU
            dbuf = torch.empty_like(u8[0], device=dev)
            hout = torch.empty(tuple(last[0].shape), dtype=torch.uint8).pin_memory()
            dout = torch.empty(tuple(last[0].shape), dtype=torch.uint8, device=dev)
            for key, fn, nb in (('h2d_alone_GBps', lambda: dbuf.copy_(u8[0], non_blocking=True), u8[0].numel()),
                                ('d2h_alone_GBps', lambda: hout.copy_(dout, non_blocking=True), hout.numel())):
                fn(); sync()
                t1 = time.perf_counter()
                for _ in range(4):
                    fn()
                sync()
                res[name][key] = round(4 * nb / (time.perf_counter() - t1) / 1e9, 1)
            del dbuf, hout, dout
        del runner
    # synchronous variant: fp32 fused frames copied to the host after every clip (the reference's .cpu() per frame)
    def step_d2h():
        o = pipeline.run_two_view(hr[0], hr[1], lr[0], lr[1], nets)
        o[0].cpu()
        return o
    dt, o = measure(step_d2h, sync, 1, 3)
    entry('720p 2-view fp32 resident in, fp32 frames D2H (blocking .cpu())', n, dt, 3, o[1], o[2])
    del o
    # streaming, batch 1, HIP graph steady state
    pushes = 96
    def stream_once():
        st = OnlineStitcher(nets, 720, 1280)
        out = None
        for t in range(pushes):
            i = t % n
            got = st.push(hr[0][i:i + 1], hr[1][i:i + 1], lr[0][i:i + 1], lr[1][i:i + 1])
            if got:
                out = got[-1]
        return out, st.hc, st.wc
    dt, o, sg = measure(stream_once, sync, 1, 4, True)
    entry('720p 2-view streaming (batch 1, one pair per push)', pushes, dt, 4, o[1], o[2], sg,
          note='fps = 96-push streams INCLUDING the 7 eager window-fill pushes and the graph capture (the figure of rounds 1-3); '
               'fps_steady = graph replays only')
    st1 = OnlineStitcher(nets, 720, 1280)
    for t in range(12):
        st1.push(hr[0][t:t + 1], hr[1][t:t + 1], lr[0][t:t + 1], lr[1][t:t + 1])
    sync()
    t0 = time.perf_counter()
    for t in range(200):
        i = t % n
        st1.push(hr[0][i:i + 1], hr[1][i:i + 1], lr[0][i:i + 1], lr[1][i:i + 1])
    sync()
    dts = time.perf_counter() - t0
    res['720p 2-view streaming (batch 1, one pair per push)']['canvas_overflow'] = st1.overflow_report()      # device-side watcher (read after the clock)
    res['720p 2-view streaming (batch 1, one pair per push)']['fps_steady'] = round(200 / dts, 1)
    res['720p 2-view streaming (batch 1, one pair per push)']['ms_per_push_steady'] = round(dts / 200 * 1e3, 4)
    res['720p 2-view streaming (batch 1, one pair per push)']['graph_nodes'] = st1.graph_nodes      # (hipGraphGetNodes of the captured step)
    del st1
    # two pushes in flight (opt-in; frames bit-identical, handed out one push late): PipelinedOnlineStitcher
    from stabstitch2_amd.online import PipelinedOnlineStitcher
    stp = PipelinedOnlineStitcher(nets, 720, 1280)
    for t in range(12):
        stp.push(hr[0][t:t + 1], hr[1][t:t + 1], lr[0][t:t + 1], lr[1][t:t + 1])
    sync()
    t0 = time.perf_counter()
    for t in range(200):
        i = t % n
        stp.push(hr[0][i:i + 1], hr[1][i:i + 1], lr[0][i:i + 1], lr[1][i:i + 1])
    stp.flush()
    sync()
    entry('720p 2-view streaming, two pushes in flight (opt-in PipelinedOnlineStitcher)', 200, time.perf_counter() - t0, 1, stp.hc, stp.wc,
          note='push t + 1`s trunks and stage-1 heads on a second HIP stream beside push t`s regressor heads, smoothing and render; '
               'frames bit-identical to OnlineStitcher, handed out one push late (flush() for the last)')
    res['720p 2-view streaming, two pushes in flight (opt-in PipelinedOnlineStitcher)']['graph_nodes'] = stp.graph_nodes
    del stp
    # the reference's own frame loop shape: decoded uint8 frames in (cv2.imread's layout), uint8 video frames out (push_u8)
    st8 = OnlineStitcher(nets, 720, 1280)
    u8 = [[hr[v][i].clamp(0, 255).to(torch.uint8).permute(1, 2, 0).contiguous() for i in range(n)] for v in range(2)]
    for t in range(12):
        st8.push_u8(u8[0][t], u8[1][t])
    sync()
    t0 = time.perf_counter()
    for t in range(200):
        i = t % n
        st8.push_u8(u8[0][i], u8[1][i])
    sync()
    entry('720p 2-view streaming from decoded uint8 frames to uint8 video frames (push_u8)', 200, time.perf_counter() - t0, 1, st8.hc, st8.wc,
          note='device-resident uint8 [H,W,3] frames in, uint8 [Hc,Wc,3] out; the cv2-exact resize feeds the graph, the render samples '
               'the uint8 frames and writes the uint8 frame: byte for byte ingest_u8 -> push -> canvas_to_u8')
    # ... and the whole loop of the reference: uint8 frames in pinned HOST memory in, uint8 frames in pinned host memory out, one pair
    # per iteration, PCIe both ways beside the pushes (HostFrameStream: upload / compute / download streams)
    from stabstitch2_amd.online import HostFrameStream
    hp = [[f.cpu().pin_memory() for f in v] for v in u8]
    runner = HostFrameStream(OnlineStitcher(nets, 720, 1280))
    for _ in runner.run(tuple(hp[v][t % n] for v in range(2)) for t in range(40)):      # window fill, capture, first replays
        pass
    sync()
    t0 = time.perf_counter()
    nout = sum(1 for _ in runner.run(tuple(hp[v][t % n] for v in range(2)) for t in range(300)))
    sync()
    dth = time.perf_counter() - t0
    assert nout == 300
    entry('720p 2-view streaming from host memory to host memory: pinned uint8 frames in, one pair per push, pinned uint8 frames out', 300,
          dth, 1, runner.st.hc, runner.st.wc,
          note='the reference`s loop shape end to end (test_online_tra.py:250-417) at batch 1: 5.5 MB up + ~4.7 MB down per push over PCIe '
               'on their own HIP streams beside the push (push_u8 on the uploaded frames); frames byte-identical to push_u8')
    runner2 = HostFrameStream(PipelinedOnlineStitcher(nets, 720, 1280))
    for _ in runner2.run(tuple(hp[v][t % n] for v in range(2)) for t in range(40)):
        pass
    sync()
    t0 = time.perf_counter()
    nout = sum(1 for _ in runner2.run(tuple(hp[v][t % n] for v in range(2)) for t in range(300)))
    sync()
    dth = time.perf_counter() - t0
    assert nout == 300
    entry('720p 2-view streaming from host memory to host memory, two pushes in flight (HostFrameStream over PipelinedOnlineStitcher)', 300,
          dth, 1, runner2.st.hc, runner2.st.wc, note='the same loop with push t + 1`s first half beside push t`s second half; frames byte-identical')
    del st8, u8, hp, runner, runner2
    std = OnlineStitcher(nets, 720, 1280, deterministic=True)
    for t in range(12):
        std.push(hr[0][t:t + 1], hr[1][t:t + 1], lr[0][t:t + 1], lr[1][t:t + 1])
    sync()
    t0 = time.perf_counter()
    for t in range(100):
        i = t % n
        std.push(hr[0][i:i + 1], hr[1][i:i + 1], lr[0][i:i + 1], lr[1][i:i + 1])
    sync()
    entry('720p 2-view streaming, deterministic kernel policy (opt-in)', 100, time.perf_counter() - t0, 1, std.hc, std.wc,
          note='OnlineStitcher(deterministic=True), steady state: frames bit-identical to the resident clip; no split-K at batch 1')
    res['720p 2-view streaming, deterministic kernel policy (opt-in)']['graph_nodes'] = std.graph_nodes
    del std
    # the three-view script as a stream: two pair chains + per-frame composition + three-image render, one HIP graph per push
    from stabstitch2_amd.online import ThreeViewOnlineStitcher
    st3 = ThreeViewOnlineStitcher(nets, 720, 1280)
    for t in range(12):
        st3.push(hr[0][t:t + 1], hr[1][t:t + 1], hr[2][t:t + 1], lr[0][t:t + 1], lr[1][t:t + 1], lr[2][t:t + 1])
    sync()
    t0 = time.perf_counter()
    for t in range(100):
        i = t % n
        st3.push(hr[0][i:i + 1], hr[1][i:i + 1], hr[2][i:i + 1], lr[0][i:i + 1], lr[1][i:i + 1], lr[2][i:i + 1])
    sync()
    entry('720p 3-view streaming (batch 1, one triple per push)', 100, time.perf_counter() - t0, 1, st3.hc, st3.wc,
          note='ThreeViewOnlineStitcher, steady state (graph replays), 100 pushes; the middle view passes the trunks once')
    res['720p 3-view streaming (batch 1, one triple per push)']['graph_nodes'] = st3.graph_nodes
    del st3
    # the three-view loop host to host: pinned uint8 triples in, pinned uint8 frames out
    from stabstitch2_amd.online import HostFrameStream
    hp3 = [[hr[v][i].clamp(0, 255).to(torch.uint8).permute(1, 2, 0).contiguous().cpu().pin_memory() for i in range(n)] for v in range(3)]
    r3 = HostFrameStream(ThreeViewOnlineStitcher(nets, 720, 1280))
    for _ in r3.run(tuple(hp3[v][t % n] for v in range(3)) for t in range(30)):
        pass
    sync()
    t0 = time.perf_counter()
    nout = sum(1 for _ in r3.run(tuple(hp3[v][t % n] for v in range(3)) for t in range(150)))
    sync()
    entry('720p 3-view streaming from host memory to host memory: pinned uint8 triples in, pinned uint8 frames out', nout, time.perf_counter() - t0,
          1, r3.st.hc, r3.st.wc, note='HostFrameStream over ThreeViewOnlineStitcher (push_u8), PCIe both ways beside the pushes')
    del hp3, r3
    from stabstitch2_amd.online import PipelinedThreeViewOnlineStitcher
    st3 = PipelinedThreeViewOnlineStitcher(nets, 720, 1280)
    for t in range(12):
        st3.push(hr[0][t:t + 1], hr[1][t:t + 1], hr[2][t:t + 1], lr[0][t:t + 1], lr[1][t:t + 1], lr[2][t:t + 1])
    sync()
    t0 = time.perf_counter()
    for t in range(100):
        i = t % n
        st3.push(hr[0][i:i + 1], hr[1][i:i + 1], hr[2][i:i + 1], lr[0][i:i + 1], lr[1][i:i + 1], lr[2][i:i + 1])
    st3.flush()
    sync()
    entry('720p 3-view streaming, two pushes in flight (opt-in PipelinedThreeViewOnlineStitcher)', 100, time.perf_counter() - t0, 1, st3.hc, st3.wc,
          note='frames bit-identical to ThreeViewOnlineStitcher, handed out one push late')
    del st3
    # batch of S independent live streams advancing together (one graph launch per push of S pairs)
    from stabstitch2_amd.online import MultiOnlineStitcher
    S = 8
    mh1, mh2 = hr[0][:S].contiguous(), hr[1][:S].contiguous()        # the S streams' current frames: [S,3,H,W]
    ml1, ml2 = lr[0][:S].contiguous(), lr[1][:S].contiguous()
    def multi_once():
        st = MultiOnlineStitcher(nets, 720, 1280, streams=S)
        out = None
        for t in range(7 + 40):
            got = st.push(mh1, mh2, ml1, ml2)
            if got[0]:
                out = got[0][-1]
        return out, st.canvas_sizes[0][0], st.canvas_sizes[0][1], st
    multi_once()
    sync()
    # time the steady state only (the first window of every stream runs through the single-stream code)
    _, hc_, wc_, stm = multi_once()
    sync()
    t0 = time.perf_counter()
    for _ in range(40):
        stm.push(mh1, mh2, ml1, ml2)
    sync()
    entry('720p 2-view streaming, %d streams per push (aggregate over the streams)' % S, 40 * S, time.perf_counter() - t0, 1, hc_, wc_,
          note='MultiOnlineStitcher: S independent live pairs advance one frame per push as one batch (one HIP graph); '
               'steady state, 40 pushes of %d pairs' % S)
    del stm
    # ... and with two pushes in flight (opt-in PipelinedMultiOnlineStitcher: frames bit-identical, handed out one push late)
    from stabstitch2_amd.online import PipelinedMultiOnlineStitcher
    stm = PipelinedMultiOnlineStitcher(nets, 720, 1280, streams=S)
    for _ in range(7 + 8):
        stm.push(mh1, mh2, ml1, ml2)
    sync()
    t0 = time.perf_counter()
    for _ in range(40):
        stm.push(mh1, mh2, ml1, ml2)
    stm.flush()
    sync()
    entry('720p 2-view streaming, %d streams per push, two pushes in flight (opt-in, aggregate)' % S, 40 * S, time.perf_counter() - t0, 1,
          stm.canvas_sizes[0][0], stm.canvas_sizes[0][1], note='PipelinedMultiOnlineStitcher; steady state, 40 pushes of %d pairs' % S)
    del stm
    # the same with 16 streams whose canvases the caller fixed to ONE size (a rig of identical cameras): one render launch per push
    S2 = 16
    mh1, mh2 = hr[0][:S2].contiguous(), hr[1][:S2].contiguous()
    ml1, ml2 = lr[0][:S2].contiguous(), lr[1][:S2].contiguous()
    stm = MultiOnlineStitcher(nets, 720, 1280, streams=S2, canvases=[(-20.0, 1880.0, -15.0, 745.0)] * S2)
    for _ in range(7 + 8):
        stm.push(mh1, mh2, ml1, ml2)
    sync()
    t0 = time.perf_counter()
    for _ in range(40):
        stm.push(mh1, mh2, ml1, ml2)
    sync()
    entry('720p 2-view streaming, %d streams per push on canvases of one size (aggregate)' % S2, 40 * S2, time.perf_counter() - t0, 1,
          stm.canvas_sizes[0][0], stm.canvas_sizes[0][1], note='one clip-style render launch per push; steady state, 40 pushes of %d pairs' % S2)
    del stm
    return res


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(self_launch(args, argv))

    claim_stdout()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    stub = args.stub_step_ms > 0
    dist = None
    host = None
    if stub:
        dev = torch.device('cpu')
    else:
        assert torch.cuda.is_available(), 'bench.py needs an MI355X'
        local_rank = local
        if args.share_device:
            local = 0
        torch.cuda.set_device(local)
        dev = torch.device('cuda', local)
        # this rank's launch thread, copy threads and pinned buffers on the GPU's NUMA node (before anything is pinned); ranks that
        # share one device split that node's CPUs into disjoint slices
        from stabstitch2_amd import hostbind
        host = hostbind.bind_to_gpu(dev, local_rank, int(os.environ.get('LOCAL_WORLD_SIZE', world)),
                                    share=world if args.share_device else 0)
    if world > 1 or args.force_collective:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if world == 1:
            os.environ.setdefault('MASTER_PORT', str(free_port()))
            os.environ.setdefault('RANK', '0')
            os.environ.setdefault('WORLD_SIZE', '1')
        if args.backend == 'nccl' and not stub:
            dist.init_process_group('nccl', device_id=dev)       # RCCL over xGMI
        else:
            dist.init_process_group('gloo')

    def sync():
        if not stub:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            if not stub:
                torch.cuda.synchronize()

    from stabstitch2_amd import dist as ssdist
    if stub:
        # launcher / gather self-test (tests/test_host_logic.py): same control flow, the step is a sleep
        for _ in range(args.warmup):
            time.sleep(args.stub_step_ms * 1e-3)
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            time.sleep(args.stub_step_ms * 1e-3 * (1 + rank))
        sync()
        dt = time.perf_counter() - t0
        rec = torch.tensor([float(args.frames * args.steps), dt, 0.0, 0.0, float(rank)], dtype=torch.float64)
        allrec = ssdist.gather_records(rec, dist, None, args.force_collective)
        # (self-test of the device-identity check: SS_STUB_PCI = comma-separated fake PCI bus ids, one per rank)
        fake = os.environ.get('SS_STUB_PCI', '').split(',')
        idents = ssdist.gather_objects({'pci_bus_id': fake[rank] if rank < len(fake) and fake[rank] else 'stub:%02d' % rank,
                                        'uuid': None, 'name': 'stub', 'hostname': socket.gethostname()}, dist)
        clash = ssdist.shared_devices(idents)
        if clash and not args.share_device:
            if dist is not None:
                dist.destroy_process_group()
            raise SystemExit('bench.py: ranks %s drive ONE physical device (%s); refusing to print a multi-GPU line (pass '
                             '--share-device for the single-GPU readiness check)' % (clash, idents[clash[0][0]]['pci_bus_id']))
        if rank == 0:
            emit(({'metric': 'launcher self-test (stubbed step)', 'value': round(ssdist.aggregate_fps(allrec), 3),
                              'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                              'ms_per_step': round(float(allrec[:, 1].max()) / args.steps * 1e3, 3), 'scaling': 'weak',
                              'ranks': dist.get_world_size() if dist is not None else 1, 'backend': 'gloo',
                              'ranks_seen_by_backend': dist.get_world_size() if dist is not None else 1,
                              'per_rank_pci_bus_id': [d['pci_bus_id'] for d in idents],
                              'per_rank_seconds': [round(float(x), 4) for x in allrec[:, 1]],
                              'clip_seeds': [int(x) for x in allrec[:, 4]]}))
        if dist is not None:
            dist.destroy_process_group()
        return

    from stabstitch2_amd import synth, pipeline, _hip
    _hip.lib()
    torch.set_grad_enabled(False)
    nets, sds = build_nets(dev, args.weights)
    # configs[3]: rank r stitches its own clip (seed = rank): N distinct video pairs on N GPUs
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

    vis = os.environ.get('HIP_VISIBLE_DEVICES', os.environ.get('ROCR_VISIBLE_DEVICES', ''))
    rec = torch.tensor([float(args.frames * args.steps), dt, float(hc), float(wc), float(rank), float(local),
                        float(-1 if not host or host.get('numa_node') is None else host['numa_node']),
                        float(0 if not host or not host.get('cpus_bound') else host['cpus_bound']),
                        float(int(vis.split(',')[0]) if vis.split(',')[0].strip().isdigit() else -1),
                        float(-1 if not host or host.get('cpu_first') is None else host['cpu_first']),
                        float(-1 if not host or host.get('cpu_last') is None else host['cpu_last'])], dtype=torch.float64)
    # the only collective: result gather (RCCL takes the record from device memory, gloo from the host)
    allrec = ssdist.gather_records(rec, dist, dev if args.backend == 'nccl' else None, args.force_collective)
    # physical identity of every rank's GPU (PCI bus id, UUID): the local index above is an echo of the launcher's numbering
    idents = ssdist.gather_objects(ssdist.device_identity(dev), dist)
    clash = ssdist.shared_devices(idents)
    seen = dist.get_world_size() if dist is not None else 1
    if (clash and not args.share_device) or seen != args.gpus:
        if dist is not None:
            dist.destroy_process_group()
        if seen != args.gpus:
            raise SystemExit('bench.py: --gpus %d but the process group has %d ranks' % (args.gpus, seen))
        raise SystemExit('bench.py: ranks %s drive ONE physical device (%s); refusing to print a multi-GPU line (pass '
                         '--share-device for the single-GPU readiness check)' % (clash, idents[clash[0][0]]))
    also360 = None
    if args.also_360 and world > 1:
        # configs[1] on every rank (its own clip seed), same bracket: barrier + synchronize on both sides, max over ranks
        hr3, lr3 = synth.make_clip_device(64, 360, 480, seed=rank, device=dev)
        for _ in range(2):
            pipeline.run_two_view(hr3[0], hr3[1], lr3[0], lr3[1], nets)
        sync()
        t3 = time.perf_counter()
        for _ in range(10):
            o3 = pipeline.run_two_view(hr3[0], hr3[1], lr3[0], lr3[1], nets)
        sync()
        rec3 = torch.tensor([640.0, time.perf_counter() - t3, float(o3[1]), float(o3[2]), float(rank)], dtype=torch.float64)
        all3 = ssdist.gather_records(rec3, dist, dev if args.backend == 'nccl' else None, False)
        also360 = {'workload': 'configs[1]: 360x480 2-view, 64-frame clip per step per GPU, 10 steps after 2 warm-up steps',
                   'value': round(ssdist.aggregate_fps(all3), 1), 'unit': 'frames/s', 'n_gpus': world,
                   'per_rank_seconds': [round(float(x), 4) for x in all3[:, 1]], 'ms_per_step': round(float(all3[:, 1].max()) * 100, 3)}
        del hr3, lr3, o3
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    tmax = float(allrec[:, 1].max())
    fps = ssdist.aggregate_fps(allrec)

    conv_ms, conv_flop, conv_n, conv_bytes, conv_exec = probe.summary()
    if args.conv_report:
        probe.report()
    executed = conv_exec / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
    equivalent = conv_flop / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
    # algorithmic HBM bytes per stitched frame: fp32 frames in, fp32 canvas out, weights once
    io_bytes = args.views * (3 * args.height * args.width + 3 * 360 * 480) * 4 + 3 * hc * wc * 4 + 70.6e6
    # HBM bytes per conv launch: STATIC, from the committed PMC passes of this same command (FETCH_SIZE / WRITE_SIZE
    # are profiler counters and cannot be read from inside the process)
    traffic = None
    pmc = None
    try:
        with open(os.path.join(ROOT, PMC_PROFILE)) as f:
            pmc = json.load(f)
        traffic = pmc['conv_family']['hbm_bytes_per_launch']
    except Exception:
        pass
    result = {
        'metric': 'stitched frames/sec, %dp %d-view (StabStitch++ inference hot path)' % (args.height, args.views),
        'value': round(fps, 3), 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(tmax / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': '%s: %dx%d %d-view, %d-frame clip per step per GPU, 7-frame SmoothWarp sliding '
                               'window, warp %s / fusion %s, %s' % (
                                   ('streaming (batch 1) ' if args.online else '') + ('' if args.io == 'f32' else '[io=%s] ' % args.io) +
                                   ('configs[4]' if args.views == 3 else ('configs[2]' if (args.height, args.width) == (720, 1280) else
                                    ('configs[1]' if (args.height, args.width) == (360, 480) else 'custom size')))
                                   + (' x %d GPUs = configs[3]' % world if world > 1 else ''),
                                   args.height, args.width, args.views, args.frames, args.warp_mode, args.fusion_mode,
                                   build_nets.weights),
                   'frames_per_step': args.frames, 'canvas': [int(hc), int(wc)], 'parallelism': 'streams%d' % world,
                   'published_reference': '28.3 fps on 1x RTX 4090 at 360x480 (README.md:30); different resolution '
                                          'and hardware, not comparable'},
        'roofline': {'bound': 'mfma', 'kernel': 'conv engine: conv_wino43_kernel / conv_wino_kernel / conv_igemm_kernel / stem_pool_kernel (fp32 MFMA, %d launches/clip)'
                     % conv_n, 'achieved': round(executed, 3), 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                     'frac': round(executed / PEAK_FP32_MFMA_TFLOPS, 4),
                     'achieved_is': 'EXECUTED MFMA flop (Winograd F(2x2,3x3) layers count 16/36 of their direct-conv flop, '
                                    'F(4x4,3x3) layers 36/144; since round 5 layer1 / layer2 / layer3 all run F(4x4,3x3), 18 of the 44 '
                                    'launches) / summed launch durations (HIP events); direct_conv_equivalent_tflops prices the same '
                                    'time on direct-convolution flop',
                     'direct_conv_equivalent_tflops': round(equivalent, 3),
                     'traffic': traffic, 'traffic_source': 'static: %s (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE passes of this '
                                                           'command), not measured in this run' % PMC_PROFILE,
                     'algorithmic_flop_per_launch': round(conv_flop / max(conv_n, 1)),
                     'executed_flop_per_launch': round(conv_exec / max(conv_n, 1)),
                     'algorithmic_bytes_per_launch': round(conv_bytes / max(conv_n, 1)),
                     'avg_launch_us': round(conv_ms * 1e3 / max(conv_n, 1), 2),
                     'kernel_ms_per_step': round(conv_ms, 3),
                     # the two kernels of the engine separately (HIP events of the same step; profiles/r06_kernel_stats.txt holds
                     # the rocprofv3 kernel trace of this same command, profiled and un-profiled clocks stated there)
                     'per_kernel': {k: {'launches_per_step': v[0], 'avg_launch_us': round(v[1] * 1e3 / max(v[0], 1), 2),
                                        'ms_per_step': round(v[1], 3),
                                        'achieved_tflops_executed': round(v[2] / (v[1] * 1e-3) / 1e12, 2) if v[1] > 0 else 0.0,
                                        'frac': round(v[2] / (v[1] * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4) if v[1] > 0 else 0.0,
                                        'direct_conv_equivalent_tflops': round(v[3] / (v[1] * 1e-3) / 1e12, 2) if v[1] > 0 else 0.0}
                                    for k, v in probe.by_kernel().items()},
                     # the HBM-side kernels (K7 / K8 / max-pool / K12-K13), HIP-event time in this run x STATIC
                     # HBM bytes of the committed PMC passes / 8 TB/s
                     'secondary': probe.secondary_report(pmc),
                     # all profiled kernels together: HBM bytes per clip from the committed PMC passes against the compulsory bytes
                     'path_hbm_traffic': path_traffic(pmc, io_bytes * args.frames),
                     'path_hbm_frac': round(fps / world * io_bytes / 1e9 / PEAK_HBM_GBS, 5),
                     # whole path against the MFMA roof: 41.31 GFLOP of dense contraction per 2-view frame
                     'path_mfma_frac': round(fps / world * 41.31e9 / (PEAK_FP32_MFMA_TFLOPS * 1e12), 4) if args.views == 2 else None,
                     'path_mfma_frac_is': 'DIRECT-convolution flop of the path (41.31 GFLOP per frame) x frames/s / peak: may exceed 1 -- the '
                                          'Winograd layers execute 16/36 (F(2x2,3x3)) and 36/144 (F(4x4,3x3)) of their direct flop; `frac` above '
                                          'is on executed flop'},
    }
    if dist is not None:
        result['ranks'] = dist.get_world_size()
        result['backend'] = (ssdist.collective_backend_version() or 'rccl') if args.backend == 'nccl' else args.backend
        result['per_rank_seconds'] = [round(float(x), 4) for x in allrec[:, 1]]
        result['clip_seeds'] = [int(x) for x in allrec[:, 4]]
        result['per_rank_device'] = [int(x) for x in allrec[:, 5]]
        result['per_rank_numa_node'] = [int(x) for x in allrec[:, 6]]
        result['per_rank_cpus_bound'] = [int(x) for x in allrec[:, 7]]
        result['per_rank_first_visible_device'] = [int(x) for x in allrec[:, 8]]
        result['per_rank_cpu_range'] = [[int(a), int(b)] for a, b in zip(allrec[:, 9], allrec[:, 10])]
        result['ranks_seen_by_backend'] = seen
        result['per_rank_pci_bus_id'] = [d.get('pci_bus_id') for d in idents]
        result['per_rank_device_uuid'] = [d.get('uuid') for d in idents]
        result['per_rank_hostname'] = [d.get('hostname') for d in idents]
        result['distinct_physical_devices'] = len({(d.get('hostname'), d.get('pci_bus_id') or d.get('uuid') or 'rank%d' % r)
                                                   for r, d in enumerate(idents)})
        if also360 is not None:
            result['also_360'] = also360
        if args.share_device:
            result['share_device'] = True
            result['ranks_sharing_a_device'] = clash
    result['host'] = {'placement': host, 'HIP_VISIBLE_DEVICES': vis or None, 'GPU_MAX_HW_QUEUES': os.environ.get('GPU_MAX_HW_QUEUES'),
                      'logical_cpus': os.cpu_count(), 'rccl_version': ssdist.collective_backend_version()}
    base = world == 1 and args.views == 2 and not args.online and args.io == 'f32' and not args.force_collective and not args.share_device
    if base and not args.no_other_configs:
        result['other_configs'] = other_configs(nets, dev, args)
        for k, v in result['other_configs'].items():          # the reference's fps definition (D2H inside the clock) beside the resident path
            if 'host->host' in k:
                ref_fps = result['other_configs']['720p 2-view fusion LINEAR']['fps'] if 'LINEAR' in k else fps
                v['frac_of_resident_path'] = round(v['fps'] / ref_fps, 3)
                v['frac_of_resident_path_steady'] = round(v['fps_steady'] / ref_fps, 3)
    # Flat scalars of the per-kernel table (the driver's record keeps scalars of `roofline` / `config` / `cpu_baseline` only)
    for kname, tag in (('conv_wino43_kernel', 'wino43'), ('conv_wino_kernel', 'wino22'), ('conv_igemm_kernel', 'igemm'),
                       ('stem_pool_kernel', 'stem')):
        pk = result['roofline']['per_kernel'].get(kname)
        if pk:
            result['roofline'][tag + '_launches'] = pk['launches_per_step']
            result['roofline'][tag + '_avg_us'] = pk['avg_launch_us']
            result['roofline'][tag + '_frac'] = pk['frac']
    if 'other_configs' in result:
        # the reference-definition figures (uint8 host -> uint8 host, D2H inside the clock: test_online_tra.py:152,
        # 402-403) and the other configurations as FLAT scalars of `config`, so that they survive any truncation of the line
        oc = result['other_configs']

        def pick(sub, key='fps'):
            for k, v in oc.items():
                if sub in k:
                    return v.get(key)
            return None
        hh = '720p 2-view uint8 host->host incl. D2H of every fused frame'
        summ = {'host_u8_fps': oc.get(hh, {}).get('fps'), 'host_u8_frac_of_resident': oc.get(hh, {}).get('frac_of_resident_path'),
                'host_u8_fps_steady': oc.get(hh, {}).get('fps_steady'),
                'host_u8_linear_fps': oc.get(hh + ', fusion LINEAR', {}).get('fps'),
                'h2d_GBps': oc.get(hh, {}).get('h2d_GBps'), 'd2h_GBps': oc.get(hh, {}).get('d2h_GBps'),
                'linear_fps': oc.get('720p 2-view fusion LINEAR', {}).get('fps'),
                'three_view_fps': pick('configs[4]'), 'three_view_linear_fps': oc.get('720p 3-view fusion LINEAR', {}).get('fps'),
                'configs1_360x480_fps': pick('configs[1]'), 'warp_fast_fps': oc.get('720p 2-view warp FAST', {}).get('fps'),
                'streaming_fps_incl_fill': pick('streaming (batch 1'), 'streaming_steady_fps': pick('streaming (batch 1', 'fps_steady'),
                'three_view_streaming_steady_fps': pick('3-view streaming'), 'streaming_8_streams_fps': pick('8 streams per push'), 'streaming_16_streams_fps': pick('16 streams per push'),
                'streaming_graph_nodes': pick('streaming (batch 1', 'graph_nodes'), 'three_view_streaming_graph_nodes': pick('3-view streaming', 'graph_nodes'),
                'streaming_pipelined_fps': pick('streaming, two pushes in flight'), 'streaming_u8_fps': pick('streaming from decoded uint8'), 'streaming_host_u8_fps': pick('streaming from host memory to host memory: pinned'), 'streaming_host_u8_pipelined_fps': pick('streaming from host memory to host memory, two pushes'),
                'three_view_streaming_host_u8_fps': pick('3-view streaming from host memory'),
                'streaming_8_streams_pipelined_fps': pick('streams per push, two pushes in flight'),
                'three_view_streaming_pipelined_fps': pick('3-view streaming, two pushes in flight'),
                'deterministic_clip_fps': pick('2-view, deterministic kernel policy'), 'deterministic_streaming_fps': pick('streaming, deterministic'),
                'wino43_one_block_per_workgroup_fps': pick('one workgroup per tile block'), 'render_eps_fold_fps': pick('folded into the row table')}
        for k, v in summ.items():
            result['config']['summary_' + k] = v
    if base and not args.no_cpu_baseline:
        ncpu = os.cpu_count() or 1
        if args.cpu_threads > 0:
            threads, sweep = max(1, min(args.cpu_threads, ncpu)), None
        else:
            threads, sweep = cpu_thread_sweep(sds, args.height, args.width, sorted({ncpu, min(64, ncpu), min(32, ncpu), min(16, ncpu)}))
        cfps, cout, csplit = cpu_baseline(sds, args.cpu_frames, args.height, args.width, threads)
        result['cpu_baseline'] = {'value': round(cfps, 4), 'unit': 'frames/s', 'cores': threads, 'kind': 'port',
                                  'sample': '%d-frame %dx%d 2-view clip (seed 0), NORMAL/AVERAGE, oracle/ on PyTorch-CPU'
                                            % (args.cpu_frames, args.height, args.width),
                                  'host_logical_cpus': ncpu, 'stage_seconds': csplit}
        if sweep is not None:
            result['cpu_baseline']['thread_sweep_seconds'] = {str(k): v for k, v in sweep.items()}
            result['cpu_baseline']['thread_sweep_sample'] = '1 SpatialNet pair + 1 two-image TPS warp per thread count; ' \
                                                            'the clip sample runs with the fastest'
        # parity at benchmark time: same clip through the HIP path
        n = args.cpu_frames
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
        for k, v in par.items():                   # flat copies where the driver's record keeps them
            result['cpu_baseline']['parity_' + k] = v
    emit(result)
    if dist is not None:
        dist.destroy_process_group()


_JSON_FD = None


def claim_stdout():
    """From here on file descriptor 1 of this rank IS stderr, and the original stdout is kept aside for the one JSON line:
    whatever native libraries print through C stdio (RCCL writes a version banner when its communicator is created -- on every
    rank, flushed at exit) can no longer land in front of, behind or instead of the line the driver parses."""
    global _JSON_FD
    if _JSON_FD is None:
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)


def emit(result):
    """The ONE JSON line of the run, on the real stdout (rank 0 only calls this)."""
    line = (json.dumps(result) + '\n').encode()
    if _JSON_FD is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        os.write(_JSON_FD, line)


if __name__ == '__main__':
    main()
