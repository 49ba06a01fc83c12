"""HIP-event phases of the compute stream inside HostClipRunner (un-profiled): ingest / estimate / render per clip, with and
without the download and the upload running beside them."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stabstitch2_amd  # noqa
import torch
import bench
from stabstitch2_amd import synth, pipeline, ops

dev = torch.device('cuda:0')
torch.cuda.set_device(0)
torch.set_grad_enabled(False)
nets, _ = bench.build_nets(dev)
hr, lr = synth.make_clip_device(32, 720, 1280, seed=0, device=dev)
u8 = [hr[v].permute(0, 2, 3, 1).round().clamp(0, 255).to(torch.uint8).contiguous().cpu().pin_memory() for v in range(2)]
del hr, lr
E = lambda: torch.cuda.Event(enable_timing=True)


class Probe(pipeline.HostClipRunner):
    marks = []
    no_download = False

    def _compute(self, d, ev):
        self.comp.wait_event(ev)
        with torch.cuda.stream(self.comp):
            for t in d:
                t.record_stream(self.comp)
            e = [E() for _ in range(4)]
            e[0].record()
            _, lr1 = ops.ingest_u8(d[0], want_hr=False)
            _, lr2 = ops.ingest_u8(d[1], want_hr=False)
            e[1].record()
            acc = pipeline.estimate_meshes(self.nets, lr1, lr2)
            e[2].record()
            u8o, hc, wc = pipeline.render_frames_u8(d, [acc['smooth_mesh1'], acc['smooth_mesh2']], self.warp_mode)
            e[3].record()
            self.marks.append(e)
            ev2 = torch.cuda.Event()
            ev2.record(self.comp)
        return u8o, hc, wc, ev2

    def _download(self, k, u8o, ev):
        if self.no_download:
            done = torch.cuda.Event(); done.record(self.comp)
            return u8o, done
        return super()._download(k, u8o, ev)


def run(r, k, src):
    r.marks = []
    t0 = time.perf_counter()
    for _ in r.run(src() for _ in range(k)):
        pass
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / k * 1e3
    m = r.marks[1:-1] if k > 4 else r.marks
    f = lambda i: sum(a[i].elapsed_time(a[i + 1]) for a in m) / len(m)
    gap = sum(m[i][3].elapsed_time(m[i + 1][0]) for i in range(len(m) - 1)) / max(len(m) - 1, 1)
    return '%.2f ms/clip | ingest %.3f estimate %.3f render %.3f | gap render->next ingest %.3f' % (dt, f(0), f(1), f(2), gap)


d_res = [t.to(dev) for t in u8]
for name, nodl, src in (('host in, host out', False, lambda: (u8[0], u8[1])), ('host in, no download', True, lambda: (u8[0], u8[1])),
                        ('device in, host out', False, lambda: (d_res[0], d_res[1])), ('device in, no download', True, lambda: (d_res[0], d_res[1]))):
    r = Probe(nets, dev)
    r.no_download = nodl
    run(r, 3, src)
    for _ in range(2):
        print('%-24s %s' % (name, run(r, 10, src)))
