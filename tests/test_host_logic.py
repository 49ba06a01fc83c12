"""CPU-only checks of the host side: C-ABI surface, checkpoint layout, weight repacking, layout rules, sharding."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def built_lib():
    from stabstitch2_amd import _hip
    if not os.path.exists(_hip.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _hip.lib()


def test_cabi_exports_exactly_the_declared_symbols(built_lib):
    """Two-sided: header == ctypes table == what the shared library exports under the ss_ prefix (the library is built
    with -fvisibility=hidden; tuning knobs such as ss_debug_set exist only in the tools/ build)."""
    from stabstitch2_amd import _hip
    hdr = open(os.path.join(ROOT, 'include', 'stabstitch_hip.h')).read()
    declared = sorted(set(re.findall(r'\b(ss_[a-z0-9_]+)\s*\(', hdr)))
    assert len(declared) >= 38
    for name in declared:
        assert hasattr(built_lib, name), name
    assert sorted(_hip.SIGNATURES) == declared, 'ctypes table and header disagree'
    nm = subprocess.run(['nm', '-D', '--defined-only', _hip.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted({ln.split()[-1] for ln in nm.splitlines() if ln.split() and ln.split()[-1].startswith('ss_')})
    assert exported == declared, ('library exports != header', sorted(set(exported) ^ set(declared)))
    assert 'getenv' not in nm                                   # no environment overrides inside the product library
    und = subprocess.run(['nm', '-D', '--undefined-only', _hip.LIB_PATH], capture_output=True, text=True).stdout
    assert 'getenv' not in und
    assert built_lib.ss_version() >= 300
    assert built_lib.ss_error_string(-1) == b'bad argument'
    # argument validation happens before any device work, so it can be exercised without a GPU
    assert built_lib.ss_conv_nhwc(None, None, None, None, None, *([1] * 15), 1, 0, 0, 0, None, 0, None) == -1
    assert built_lib.ss_maxpool_nhwc(None, None, 1, 4, 4, 4, 2, 2, 0, None) == -1
    assert built_lib.ss_tps_solve(None, None, None, 1, None) == -1
    assert built_lib.ss_ccl_workspace_floats(2, 23, 30, 256) == 2 * 690 * (512 + 690) + 64      # + slack for 16-byte reads at shifted columns
    assert built_lib.ss_tsmotion_workspace_floats(10) == 126 + 10 * 384
    # split-K plan: large launches need no workspace, the tiny-map regressor tail does
    assert built_lib.ss_conv_workspace_need(64, 1, 90, 120, 64, 64, 1, 3, 3, 1, 0, 1, 1, 1) == 0
    need = built_lib.ss_conv_workspace_need(32, 1, 5, 7, 256, 256, 1, 3, 3, 1, 0, 1, 1, 1)
    assert need > 0 and need % (32 * 5 * 7 * 256) == 0


def test_no_matrix_kernel_spills_or_uses_scratch(built_lib):
    """Every gfx950 code object of the built library, read back from its metadata notes (tools/kernel_resources.py): a kernel that
    issues MFMA instructions must have .vgpr_spill_count = .sgpr_spill_count = 0 and no private segment.  Twice a spill shipped
    unnoticed (round 4: 47 registers of conv_wino43_kernel, 114 KB of scratch stores per workgroup; round 5: 7 of
    stem_pool_kernel_half with a scratch_load inside its MFMA stream)."""
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    try:
        import kernel_resources
    finally:
        sys.path.pop(0)
    from stabstitch2_amd import _hip
    if not os.path.exists(os.path.join(kernel_resources.LLVM, 'llvm-readelf')):
        pytest.skip('no llvm-readelf in this image')
    ks = kernel_resources.kernels(_hip.LIB_PATH)
    mfma = {k: v for k, v in ks.items() if v['mfma_instructions'] > 0}
    assert len(ks) >= 60 and len(mfma) >= 30, (len(ks), len(mfma))
    for want in ('conv_igemm_kernel', 'conv_wino_kernel', 'conv_wino43p_kernel', 'stem_pool_kernel_half'):
        assert any(want in k for k in mfma), want
    # (SGPR spills go to VGPR lanes -- v_writelane / v_readlane, no memory: the persistent F(4x4,3x3) kernel parks up to six
    # block-loop scalars that way outside its K loop; what must never ship is a VGPR spilled to scratch memory)
    bad = {k: (v.get('.vgpr_spill_count', 0), v.get('.private_segment_fixed_size', 0))
           for k, v in mfma.items() if v.get('.vgpr_spill_count', 0) or v.get('.private_segment_fixed_size', 0)}
    assert not bad, 'MFMA kernels with VGPR spills / scratch (vgpr spills, scratch bytes): %s' % bad
    assert max(v.get('.sgpr_spill_count', 0) for v in mfma.values()) <= 8
    # the register budgets the occupancy arguments of DESIGN.md rest on
    half = next(v for k, v in ks.items() if 'stem_pool_kernel_half' in k)
    assert half['.vgpr_count'] + half.get('.agpr_count', 0) <= 128          # four workgroups of 256 threads per CU
    for k, v in ks.items():
        if 'conv_wino43' in k:               # (both the persistent and the one-block-per-workgroup kernel)
            assert v['.vgpr_count'] <= 256 and v.get('.vgpr_spill_count', 0) == 0, k


def test_product_never_imports_oracle_and_fails_without_gpu():
    pkg = os.path.join(ROOT, 'stabstitch2_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, re.M), os.path.join(dirpath, f)
    assert 'TEST INFRASTRUCTURE ONLY' in open(os.path.join(ROOT, 'oracle', '__init__.py')).read()
    from stabstitch2_amd import ops, _hip
    with pytest.raises(_hip.HipError):
        ops.linear(torch.zeros(1, 4), torch.zeros(2, 4))          # CPU tensors are refused: no CPU path
    from stabstitch2_amd.spatial_network import SpatialNet
    with pytest.raises(_hip.HipError):
        SpatialNet()(torch.zeros(1, 3, 360, 480), torch.zeros(1, 3, 360, 480))


def test_checkpoint_layout_roundtrip(tmp_path):
    from stabstitch2_amd import synth
    from stabstitch2_amd.spatial_network import SpatialNet
    from stabstitch2_amd.temporal_network import TemporalNet
    from stabstitch2_amd.smooth_network import SmoothNet
    from oracle import nets as ON
    nets = [SpatialNet(), TemporalNet(), SmoothNet()]
    synth.write_synthetic_checkpoints(str(tmp_path), *nets)
    assert sorted(os.listdir(tmp_path)) == ['smooth_warp.pth', 'spatial_warp.pth', 'temporal_warp.pth']
    for name, net, ocls, count in (('spatial_warp', nets[0], ON.SpatialNet, 130),
                                   ('temporal_warp', nets[1], ON.TemporalNet, 104),
                                   ('smooth_warp', nets[2], ON.SmoothNet, 14)):
        ck = torch.load(os.path.join(tmp_path, name + '.pth'))
        assert list(ck) == ['model'] and len(ck['model']) == count
        net.load_state_dict(ck['model'], strict=True)
        ocls().load_state_dict(ck['model'], strict=True)          # same keys as the oracle (= the reference layout)
    keys = set(nets[0].state_dict())
    for k in ('regressNet1_part1.12.weight', 'regressNet2_part1_ref.17.weight', 'regressNet2_part2_tgt.4.bias',
              'feature_extractor_stage1.0.weight', 'feature_extractor_stage1.1.running_var',
              'feature_extractor_stage1.5.0.downsample.1.num_batches_tracked',
              'feature_extractor_stage2.0.1.bn2.weight'):
        assert k in keys, k
    assert 'MotionPre.embedding2.0.weight' in nets[2].state_dict()      # unused by forward, required by strict load


def test_weight_repacking_matches_eval_semantics():
    from stabstitch2_amd import layers as L
    torch.manual_seed(0)
    conv = torch.nn.Conv2d(3, 8, 3, 2, 1, bias=False)
    bn = torch.nn.BatchNorm2d(8).eval()
    bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2); bn.weight.data.normal_(); bn.bias.data.normal_()
    x = torch.randn(2, 3, 9, 11)
    w, b = L.pack_conv2d(conv, bn)
    assert w.shape == (8, 1, 3, 3, 4) and float(w[..., 3].abs().max()) == 0.0
    y = F.conv2d(x, w[:, 0, :, :, :3].permute(0, 3, 1, 2), b, stride=2, padding=1)
    assert torch.allclose(y, bn(conv(x)), atol=1e-5)
    lin = torch.nn.Linear(5 * 6, 7)
    act = torch.randn(2, 5, 2, 3)                                   # NCHW feature map, c=5, hw=6
    w2, b2 = L.pack_fc_first(lin, 5, 6)
    nhwc_flat = act.permute(0, 2, 3, 1).reshape(2, -1)
    assert torch.allclose(F.linear(nhwc_flat, w2, b2), lin(act.reshape(2, -1)), atol=1e-6)
    stem = torch.nn.Conv2d(3, 8, 7, 2, 3, bias=False)
    bn7 = torch.nn.BatchNorm2d(8).eval()
    bn7.running_mean.normal_(); bn7.running_var.uniform_(0.5, 2); bn7.weight.data.normal_(); bn7.bias.data.normal_()
    ws, bs = L.pack_stem3(stem, bn7)                                 # [cout, 7 rows, 24 = 7 taps x 3 channels + 3 zeros]
    assert ws.shape == (8, 7, 24) and float(ws[..., 21:].abs().max()) == 0.0
    w_back = ws[..., :21].reshape(8, 7, 7, 3).permute(0, 3, 1, 2)    # [co][c][dh][dw]
    xs = torch.randn(1, 3, 20, 24)
    assert torch.allclose(F.conv2d(xs, w_back, bs, stride=2, padding=3), bn7(stem(xs)), atol=1e-5)
    c3 = torch.nn.Conv3d(4, 6, (5, 3, 3), padding=(2, 1, 1))
    w3, b3 = L.pack_conv3d(c3)
    assert w3.shape == (6, 5, 3, 3, 4) and torch.equal(w3[2, 4, 1, 0, 3], c3.weight[2, 3, 4, 1, 0])


def test_synthetic_inputs_are_deterministic():
    from stabstitch2_amd import synth
    a = synth.make_clip(2, 72, 96, seed=3)
    b = synth.make_clip(2, 72, 96, seed=3)
    assert torch.equal(a[0][1][1], b[0][1][1]) and a[1][0][0].shape == (1, 3, 360, 480)
    hr, lr = synth.make_clip_device(2, 72, 96, seed=3, device='cpu')
    assert float((hr[1, 1] - a[0][1][1][0]).abs().max()) < 1e-3
    assert float(lr.abs().max()) <= 1.0 + 1e-6
    from oracle import nets as ON
    sd1 = synth.synthetic_state_dict(ON.SmoothNet())
    sd2 = synth.synthetic_state_dict(ON.SmoothNet())
    assert all(torch.equal(sd1[k], sd2[k]) for k in sd1)


def test_stream_sharding_rules():
    from stabstitch2_amd import dist as D
    assert D.shard_streams(8, 3, 8) == [3]
    assert D.shard_streams(10, 1, 4) == [1, 5, 9]
    assert sorted(sum((D.shard_streams(11, r, 4) for r in range(4)), [])) == list(range(11))
    with pytest.raises(ValueError):
        D.shard_streams(4, 4, 4)
    rec = torch.tensor([[64.0, 2.0], [64.0, 4.0]], dtype=torch.float64)
    assert D.aggregate_fps(rec) == 32.0


def _gloo_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from stabstitch2_amd import dist as D
    mine = D.shard_streams(5, rank, world)
    rec = torch.tensor([32.0 * len(mine), 1.0 + rank, 740.0, 1882.0 + rank], dtype=torch.float64)
    allrec = D.gather_records(rec, dist)
    dist.barrier()
    q.put((rank, mine, allrec.tolist()))
    dist.destroy_process_group()


def test_two_rank_gloo_gather():
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 2, 4] and res[1][1] == [1, 3]
    for _, _, allrec in res:
        assert allrec == [[96.0, 1.0, 740.0, 1882.0], [64.0, 2.0, 740.0, 1883.0]]
    from stabstitch2_amd import dist as D
    assert D.aggregate_fps(torch.tensor(res[0][2])) == 160.0 / 2.0


def test_bench_self_launches_ranks_and_gathers():
    """`python bench.py --gpus 2` with no launcher around it starts two ranks itself (torch.distributed.run, 127.0.0.1),
    each with its own clip seed, gathers one record per rank and prints ONE JSON line from rank 0.  Driven here with the
    gloo backend and a stubbed step (no GPU in this container); on a GPU node the same code path initialises RCCL."""
    import json
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None); env.pop('RANK', None); env.pop('LOCAL_RANK', None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
                        '--frames', '32', '--backend', 'gloo', '--stub-step-ms', '20'], capture_output=True, text=True,
                       env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['ranks'] == 2 and out['clip_seeds'] == [0, 1] and len(out['per_rank_seconds']) == 2
    # whole-job value = all ranks' frames / slowest rank (rank 1 sleeps twice as long per step)
    assert out['per_rank_seconds'][1] >= out['per_rank_seconds'][0] * 0.9
    assert abs(out['value'] - 2 * 32 * 3 / max(out['per_rank_seconds'])) / out['value'] < 0.05
    assert out['ranks_seen_by_backend'] == 2 and out['backend'] == 'gloo' and len(set(out['per_rank_pci_bus_id'])) == 2
    # two ranks that report ONE physical device (same host, same PCI bus id): refused without --share-device, printed with it
    env3 = dict(env, SS_STUB_PCI='0000:05:00.0,0000:05:00.0')
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--frames', '8',
           '--backend', 'gloo', '--stub-step-ms', '5']
    r3 = subprocess.run(cmd, capture_output=True, text=True, env=env3, timeout=300)
    assert r3.returncode != 0 and 'ONE physical device' in (r3.stderr + r3.stdout)
    assert not [ln for ln in r3.stdout.splitlines() if ln.startswith('{')]
    r4 = subprocess.run(cmd + ['--share-device'], capture_output=True, text=True, env=env3, timeout=300)
    assert r4.returncode == 0 and len([ln for ln in r4.stdout.splitlines() if ln.startswith('{')]) == 1, r4.stderr[-2000:]
    # N = 1 never spawns; a launcher-provided WORLD_SIZE that disagrees with --gpus is refused
    env2 = dict(env, WORLD_SIZE='4', RANK='0', LOCAL_RANK='0')
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--stub-step-ms', '1'],
                        capture_output=True, text=True, env=env2, timeout=120)
    assert r2.returncode != 0 and 'WORLD_SIZE=4' in (r2.stderr + r2.stdout)


def test_ranks_sharing_a_physical_device_are_found():
    """dist.shared_devices: ranks are grouped by (hostname, PCI bus id) -- or the UUID where the platform reports no bus id; the local
    device index plays no part (HIP_VISIBLE_DEVICES remaps it), a rank that reports neither is left alone."""
    from stabstitch2_amd import dist as D
    ident = lambda host, pci=None, uuid=None: {'hostname': host, 'pci_bus_id': pci, 'uuid': uuid, 'name': 'x'}
    eight = [ident('n0', '0000:%02x:00.0' % (5 + 8 * r)) for r in range(8)]
    assert D.shared_devices(eight) == []
    assert D.shared_devices(eight[:3] + [eight[1]]) == [[1, 3]]
    assert D.shared_devices([ident('n0', '0000:05:00.0'), ident('n1', '0000:05:00.0')]) == []        # same slot, different hosts
    assert D.shared_devices([ident('n0', None, 'GPU-a'), ident('n0', None, 'GPU-a'), ident('n0', None, 'GPU-b')]) == [[0, 1]]
    assert D.shared_devices([ident('n0'), ident('n0')]) == []
    assert D.gather_objects({'a': 1}) == [{'a': 1}]                  # no process group: identity
    assert D.shard_streams(10, 1, 4) == [1, 5, 9]


def test_default_switches_are_the_measured_configuration():
    """The headline number is plain fp32 MFMA arithmetic with every default optimisation on; the opt-in bf16-slice
    products must stay opt-in (a default flipped by accident would change what `dtype: f32` of the bench line means)."""
    if any(k in os.environ for k in ('SS_WINO_MATH', 'SS_WINOGRAD', 'SS_SKIP_OUTSIDE', 'SS_U8_FUSED')):
        pytest.skip('an A/B switch is set in the environment')
    from stabstitch2_amd import ops, pipeline
    assert ops.WINO_MATH == 'f32' and ops.WINOGRAD is True
    if 'SS_WINO43' not in os.environ and 'SS_WINO43_MIN_CIN' not in os.environ:
        assert ops.WINO43 == 'auto' and ops.WINO43_MIN_CIN == 0 and ops.WINO43_MIN_WGS == 0      # 0 = the library's thresholds
        from stabstitch2_amd import _hip
        rule = _hip.lib().ss_conv_uses_wino43                    # (pure host function: callable without a GPU)
        assert rule(1, 3, 3, 1, 64, 64, 90, 120, 64, 1, 0, 0, 0) == 1 and rule(1, 3, 3, 1, 128, 128, 45, 60, 64, 1, 0, 0, 0) == 1
        assert rule(1, 3, 3, 1, 128, 128, 45, 60, 16, 1, 0, 0, 0) == 0          # 384 workgroups < 512
        assert rule(1, 3, 3, 1, 128, 128, 45, 60, 16, 1, 1, 1, 1) == 1          # thresholds pinned to 1: geometry only
        assert rule(1, 3, 3, 1, 32, 64, 90, 120, 64, 1, 0, 0, 0) == 0           # cin < 64
        assert rule(1, 3, 3, 1, 256, 256, 23, 30, 64, 1, 0, 0, 0) == 1          # narrow map: the 16 x 32 block geometry (67 % filled >= 60 %)
        assert rule(1, 3, 3, 1, 256, 256, 23, 30, 32, 1, 0, 0, 0) == 0          # ... when the launch is deep enough (256 workgroups < 512)
        assert rule(1, 3, 3, 1, 128, 128, 11, 15, 248, 1, 0, 0, 0) == 0         # 11 x 15 fills 32 % of a 16 x 32 block
        assert rule(1, 3, 3, 2, 64, 128, 45, 60, 64, 1, 0, 0, 0) == 0 and rule(1, 3, 3, 1, 24, 64, 90, 120, 64, 1, 1, 1, 1) == 0
        assert rule(1, 3, 3, 1, 64, 64, 90, 120, 16, 4, 0, 0, 0) == 1           # groups count towards the launch depth
    assert pipeline.SKIP_OUTSIDE is True and pipeline.U8_FUSED is True
    if not any(k in os.environ for k in ('SS_DETERMINISTIC', 'SS_RENDER_EPS_FOLD', 'SS_WINO43_PERSIST')):
        # round 6: the deterministic kernel policy and the render's eps fold are opt-ins (the fold is not the reference's arithmetic);
        # the persistent F(4x4,3x3) launch is the default; the policy context restores what it found
        assert ops.DETERMINISTIC is False and ops.RENDER_EPS_FOLD is False and ops.WINO43_PERSIST is True
        assert ops._avg_mode('NORMAL') == 0 and ops._avg_mode('FAST') == 1
        import threading
        seen = []
        with ops.deterministic():
            assert ops.is_deterministic() and ops._conv_ws_need(1, 1, 5, 7, 256, 256, 1, 3, 3, 1, 0, 1, 1, 4) == 0
            with ops.deterministic(False):
                assert ops.is_deterministic()
            th = threading.Thread(target=lambda: seen.append(ops.is_deterministic()))        # the policy is per thread
            th.start(); th.join()
        assert seen == [False]
        assert not ops.is_deterministic() and ops._conv_ws_need(1, 1, 5, 7, 256, 256, 1, 3, 3, 1, 0, 1, 1, 4) > 0
    src = open(os.path.join(ROOT, 'bench.py')).read()
    assert "'dtype': 'f32'" in src


# ------------------------------------------------------------------ round 4: host placement, new entry points (no GPU needed)
def test_hostbind_sysfs_parsing_and_opt_out(monkeypatch):
    """stabstitch2_amd.hostbind: sysfs cpulist parsing, node discovery, and the switches that leave the process alone."""
    from stabstitch2_amd import hostbind
    assert hostbind.parse_cpulist('0-3,8,10-11') == [0, 1, 2, 3, 8, 10, 11]
    assert hostbind.parse_cpulist('') == [] and hostbind.parse_cpulist(None) == []
    nodes = hostbind.numa_nodes()
    assert isinstance(nodes, dict) and all(isinstance(k, int) and isinstance(v, list) for k, v in nodes.items())
    before = os.sched_getaffinity(0)
    monkeypatch.setenv('SS_NUMA_BIND', '0')
    rep = hostbind.bind_to_gpu('cpu')                       # no GPU here: nothing is bound, the report says why
    assert rep['skipped'] == 'SS_NUMA_BIND=0' and os.sched_getaffinity(0) == before
    monkeypatch.delenv('SS_NUMA_BIND')
    rep = hostbind.bind_to_gpu('cpu')                       # no PCI function behind 'cpu' -> no node -> skipped, affinity untouched
    assert 'skipped' in rep and os.sched_getaffinity(0) == before
    assert hostbind.report('cpu') is rep
    # MPOL_DEFAULT round trip through the raw syscall (harmless; checks the ctypes plumbing)
    assert hostbind.set_mempolicy(hostbind.MPOL_DEFAULT) == 0


def test_package_import_raises_the_hw_queue_budget():
    """Importing the package sets GPU_MAX_HW_QUEUES (the host-fed runners need a hardware queue per stream) unless the user
    set it; checked in a fresh interpreter because this process may have it already."""
    env = dict(os.environ)
    env.pop('GPU_MAX_HW_QUEUES', None)
    out = subprocess.run([sys.executable, '-c', 'import os, stabstitch2_amd; print(os.environ["GPU_MAX_HW_QUEUES"])'],
                         capture_output=True, text=True, cwd=ROOT, env=env, check=True).stdout.strip()
    assert out == '16'
    env['GPU_MAX_HW_QUEUES'] = '6'
    out = subprocess.run([sys.executable, '-c', 'import os, stabstitch2_amd; print(os.environ["GPU_MAX_HW_QUEUES"])'],
                         capture_output=True, text=True, cwd=ROOT, env=env, check=True).stdout.strip()
    assert out == '6'


def test_round4_entry_points_validate_arguments(built_lib):
    """Argument validation of the round-4 entry points happens before any device work."""
    L = built_lib
    assert L.ss_linear_clip_workspace_floats(32, 2, 740, 1882) == 32 * (2 * 4 * 740 * 1882 + 32 + 30 * 93 * 4 * 8) + 1
    assert L.ss_linear_clip_workspace_floats(1, 4, 100, 100) == 0
    assert L.ss_render_linear_clip(None, None, None, None, None, 1, 2, 8, 8, 32, 32, 0, None, None) == -1
    assert L.ss_linear_grouped(None, 0, None, None, None, 4, 1, 8, 8, 0, None) == -1
    assert L.ss_conv_pool2_nhwc(None, None, None, None, *([1] * 13), 0, 0, 0, None, 0, None) == -1
    assert L.ss_tsmotion_lag(None, None, None, None, 4, 0, 360.0, 480.0, None, None, None) == -1
    assert L.ss_cost_volume_set_tile(5) == -1 and L.ss_cost_volume_set_tile(0) == 0
    assert L.ss_linear_clip_set_rows(0) == 0
    assert L.ss_version() >= 400
    # Winograd F(4x4,3x3): pack sizes, geometry it does not take
    assert L.ss_wino43_packed_floats(128, 128) == 4 * 8 * 36 * 2 * 64 * 4
    assert L.ss_wino43_packed_floats(128, 120) == 0 and L.ss_wino43_packed_floats(96, 80) == 4608 * 3 * 5 * 4
    assert L.ss_wino43_pack(None, None, 64, 64, 1, None) == -1
    assert L.ss_conv3x3_wino43_nhwc(None, None, None, None, None, 1, 8, 8, 64, 64, 0, 64, 1, 0, 0, 0, None) == -1


def test_wino43_dispatch_rule(monkeypatch):
    """ops._uses_wino43 (host logic, no device work): the deep launches of the 60 / 120-wide trunk maps, nothing else."""
    from stabstitch2_amd import ops
    monkeypatch.setattr(ops, 'WINO43', 'auto')
    monkeypatch.setattr(ops, 'WINO43_MIN_CIN', 64)
    monkeypatch.setattr(ops, 'WINOGRAD', True)
    monkeypatch.setattr(ops, 'WINO_MATH', 'f32')
    yes = lambda *a, **k: ops._uses_wino43(1, 3, 3, 1, (0, 1, 1), *a, **k)
    assert yes(128, 128, 45, 60, 64) and yes(64, 64, 90, 120, 64) and yes(64, 64, 90, 120, 32)
    assert not yes(128, 128, 45, 60, 32)            # 384 workgroups: less than two rounds of the chip
    assert yes(256, 256, 23, 30, 64)                # layer3: the 16 x 32 block geometry of narrow maps
    assert not yes(256, 256, 23, 30, 32) and not yes(128, 128, 11, 15, 248)
    assert not yes(128, 128, 45, 60, 2)             # streaming
    assert not yes(120, 128, 45, 60, 64) and not yes(128, 96, 45, 60, 64)
    assert not ops._uses_wino43(1, 3, 3, 2, (0, 1, 1), 64, 128, 45, 60, 64)
    assert not ops._uses_wino43(5, 3, 3, 1, (2, 1, 1), 128, 128, 7, 9, 64)
    assert yes(128, 128, 45, 60, 16, groups=4)      # grouped launches count every group's workgroups
    monkeypatch.setattr(ops, 'WINO_MATH', 'bf16x9')
    assert not yes(128, 128, 45, 60, 64)            # the exact-product variant exists for F(2x2,3x3) only
    monkeypatch.setattr(ops, 'WINO_MATH', 'f32')
    monkeypatch.setattr(ops, 'WINO43', '0')
    assert not yes(128, 128, 45, 60, 64)
    monkeypatch.setattr(ops, 'WINO43', '1')
    assert yes(256, 256, 23, 30, 1) and not yes(120, 128, 45, 60, 64)


def test_wino43_capability_fallback(monkeypatch):
    """A device that cannot give the F(4x4,3x3) kernel its 144 KB of LDS answers SS_ERR_DEVICE at the first launch (nothing has been
    launched): ops.conv remembers THAT device (others keep the kernel, the process-wide switch is untouched) and takes the next
    kernel; a launch the kernel cannot address (SS_ERR_UNSUPPORTED) falls through for that launch only; with SS_WINO43=1 (forced)
    either error is raised; any other error is raised."""
    from stabstitch2_amd import ops, _hip

    class X:                       # stands in for a tensor: only .device.index is looked at before the launch
        def __init__(self, index):
            self.device = type('D', (), {'index': index})()

    def refuse(code):
        def f(*a, **k):
            e = _hip.HipError('ss_conv3x3_wino43_nhwc failed (%d)' % code)
            e.code = code
            raise e
        return f
    monkeypatch.setattr(ops, 'WINO43', 'auto')
    monkeypatch.setattr(ops, '_WINO43_REFUSED', set())
    monkeypatch.setattr(ops, 'conv_winograd43', refuse(-3))
    assert ops._try_wino43(X(0), None, None, None, False, None) is None
    assert ops.WINO43 == 'auto' and not ops._WINO43_REFUSED          # one odd launch switches nothing off
    monkeypatch.setattr(ops, 'conv_winograd43', refuse(-4))
    assert ops._try_wino43(X(1), None, None, None, False, None) is None
    assert ops.WINO43 == 'auto' and ops._WINO43_REFUSED == {1}
    calls = []
    monkeypatch.setattr(ops, 'conv_winograd43', lambda *a, **k: calls.append(1) or 'ran')
    assert ops._try_wino43(X(1), None, None, None, False, None) is None and not calls       # device 1: not tried again
    assert ops._try_wino43(X(0), None, None, None, False, None) == 'ran'                    # device 0 keeps the kernel
    monkeypatch.setattr(ops, 'WINO43', '1')
    monkeypatch.setattr(ops, 'conv_winograd43', refuse(-4))
    with pytest.raises(_hip.HipError):
        ops._try_wino43(X(0), None, None, None, False, None)
    monkeypatch.setattr(ops, 'WINO43', 'auto')
    monkeypatch.setattr(ops, 'conv_winograd43', refuse(-2))
    with pytest.raises(_hip.HipError):
        ops._try_wino43(X(0), None, None, None, False, None)
    assert ops.WINO43 == 'auto'
