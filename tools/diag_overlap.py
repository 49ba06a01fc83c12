"""Why do the PCIe copies of HostClipRunner not hide behind the compute?  One child per (environment, stream layout).
    python tools/diag_overlap.py
"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(layout):
    import torch
    dev = torch.device('cuda:0')
    torch.cuda.set_device(0)
    import bench
    from stabstitch2_amd import synth, pipeline
    torch.set_grad_enabled(False)
    nets, _ = bench.build_nets(dev)
    n = 32
    hr, lr = synth.make_clip_device(n, 720, 1280, seed=0, device=dev)
    u8 = [hr[v].permute(0, 2, 3, 1).round().clamp(0, 255).to(torch.uint8).contiguous().cpu().pin_memory() for v in range(2)]
    S = torch.cuda.Stream
    pre = []
    for _ in range(int(os.environ.get('DIAG_PRE', '0'))):        # streams someone else used before (each takes a HW queue)
        st = S(dev)
        with torch.cuda.stream(st):
            torch.zeros(1024, device=dev).add_(1)
        pre.append(st)
    torch.cuda.synchronize()
    if layout == 'normal3':
        streams = (S(dev), S(dev), S(dev))
    elif layout == 'io_high':
        streams = (S(dev, priority=-1), S(dev), S(dev, priority=-1))
    elif layout == 'io_high_shared':
        io = S(dev, priority=-1)
        streams = (io, S(dev), io)
    elif layout == 'comp_high':
        streams = (S(dev), S(dev, priority=-1), S(dev))
    elif layout == 'comp_default':
        streams = (S(dev), torch.cuda.default_stream(dev), S(dev))
    elif layout == 'comp_default_io_high':
        streams = (S(dev, priority=-1), torch.cuda.default_stream(dev), S(dev, priority=-1))
    elif layout == 'io_shared':
        io = S(dev)
        streams = (io, S(dev), io)
    runner = pipeline.HostClipRunner(nets, dev, streams=streams)

    def run(k):
        stamps = []
        t0 = time.perf_counter()
        for _ in runner.run((u8[0], u8[1]) for _ in range(k)):
            stamps.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, stamps
    run(3)
    reps = []
    for _ in range(5):
        dt, stamps = run(10)
        reps.append(round(dt / 10 * 1e3, 2))
    print('OVL %-22s %-40s ms/clip %s' % (layout, os.environ.get('DIAG_ENV', ''), reps), flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        envs = [{'DIAG_PRE': '2'}, {'DIAG_PRE': '2', 'GPU_MAX_HW_QUEUES': '8'}, {'DIAG_PRE': '5'}, {'DIAG_PRE': '5', 'GPU_MAX_HW_QUEUES': '16'}]
        layouts = ['normal3', 'io_high', 'io_high_shared', 'comp_high', 'comp_default', 'comp_default_io_high', 'io_shared']
        for e in envs:
            for l in layouts:
                env = dict(os.environ)
                env.update(e)
                env['DIAG_ENV'] = ' '.join('%s=%s' % kv for kv in e.items())
                subprocess.run([sys.executable, os.path.abspath(__file__), l], env=env)
