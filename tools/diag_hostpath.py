"""Where does the host->host path (uint8 pinned in, uint8 pinned out, HostClipRunner) lose its time?
One child process per placement (none / GPU's node / the other node): H2D and D2H GB/s alone, the runner's per-clip wall
times, device allocations inside the timed region, the resident path beside it.
    python tools/diag_hostpath.py            # parent: runs the children
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(mode):
    import torch
    from stabstitch2_amd import hostbind
    dev = torch.device('cuda:0')
    torch.cuda.set_device(0)
    nodes = hostbind.numa_nodes()
    gnode = hostbind.gpu_numa_node(dev)
    info = {'mode': mode, 'nodes': {k: len(v) for k, v in nodes.items()}, 'gpu_node': gnode,
            'pci': hostbind.gpu_pci_bus_id(dev), 'affinity_before': len(os.sched_getaffinity(0))}
    if mode == 'local':
        info['bind'] = hostbind.bind_to_gpu(dev)
    elif mode == 'remote' and gnode is not None and len(nodes) > 1:
        other = [k for k in nodes if k != gnode][0]
        os.environ['SS_NUMA_NODE'] = str(other)
        info['bind'] = hostbind.bind_to_gpu(dev)
    elif mode == 'cpuonly' and gnode is not None:
        os.sched_setaffinity(0, nodes[gnode])
    info['affinity_after'] = len(os.sched_getaffinity(0))
    import bench
    from stabstitch2_amd import synth, pipeline
    torch.set_grad_enabled(False)
    nets, _ = bench.build_nets(dev)
    n = 32
    hr, lr = synth.make_clip_device(n, 720, 1280, seed=0, device=dev)
    u8 = [hr[v].permute(0, 2, 3, 1).round().clamp(0, 255).to(torch.uint8).contiguous().cpu().pin_memory() for v in range(2)]
    # copies alone
    dbuf = torch.empty_like(u8[0], device=dev)
    hout = torch.empty((n, 730, 1862, 3), dtype=torch.uint8).pin_memory()
    dout = torch.empty((n, 730, 1862, 3), dtype=torch.uint8, device=dev)

    def tm(fn, k=8):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / k
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

    def h2d():
        with torch.cuda.stream(s1):
            dbuf.copy_(u8[0], non_blocking=True)

    def d2h():
        with torch.cuda.stream(s2):
            hout.copy_(dout, non_blocking=True)
    info['h2d_GBps'] = round(u8[0].numel() / tm(h2d) / 1e9, 1)
    info['d2h_GBps'] = round(hout.numel() / tm(d2h) / 1e9, 1)
    t_both = tm(lambda: (h2d(), d2h()))
    info['both_GBps'] = round((u8[0].numel() + hout.numel()) / t_both / 1e9, 1)
    # resident path
    for _ in range(3):
        pipeline.run_two_view(hr[0], hr[1], lr[0], lr[1], nets)
    info['resident_ms'] = round(tm(lambda: pipeline.run_two_view(hr[0], hr[1], lr[0], lr[1], nets), 10) * 1e3, 2)
    runner = pipeline.HostClipRunner(nets, dev)

    def run(k):
        stamps = []
        t0 = time.perf_counter()
        for _ in runner.run((u8[0], u8[1]) for _ in range(k)):
            stamps.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, stamps
    run(3)
    reps = []
    for _ in range(4):
        a0 = torch.cuda.memory_stats(dev).get('num_device_alloc', 0)
        dt, stamps = run(10)
        a1 = torch.cuda.memory_stats(dev).get('num_device_alloc', 0)
        gaps = [round((stamps[i] - stamps[i - 1]) * 1e3, 2) for i in range(1, len(stamps))]
        reps.append({'ms_per_clip': round(dt / 10 * 1e3, 2), 'fps': round(320 / dt, 1), 'device_allocs': a1 - a0, 'yield_gaps_ms': gaps})
    info['runner'] = reps
    print('DIAG ' + json.dumps(info), flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        for f in ('/sys/devices/system/node/online', '/proc/cpuinfo'):
            pass
        print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)), flush=True)
        os.system('ls /sys/devices/system/node/ 2>&1 | head; cat /sys/class/drm/card*/device/numa_node 2>&1 | head; '
                  'lscpu 2>/dev/null | grep -i -E "numa|socket|model name|^cpu\\(s\\)"; nproc')
        for mode in ('none', 'local', 'remote', 'cpuonly', 'none'):
            subprocess.run([sys.executable, os.path.abspath(__file__), mode])
