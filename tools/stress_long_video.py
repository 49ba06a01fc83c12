"""Repeatability of the three-stream long-video path (upload / compute / download overlap): the same host video through
run_two_view_long / run_three_view_long with several chunk sizes, several times, must give the same bytes every time -- and, at the
resident path's own chunking (32), the resident clip's bytes.  A race between the streams would show up as a mismatch.
    python tools/stress_long_video.py [rounds]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from stabstitch2_amd import synth, pipeline
dev = torch.device('cuda:0'); torch.set_grad_enabled(False)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
nets, _ = bench.build_nets(dev)
n, h, w = 75, 240, 400
hr, _ = synth.make_clip_device(n, h, w, seed=21, views=3, device=dev)
u8 = [hr[v].permute(0, 2, 3, 1).round().clamp(0, 255).to(torch.uint8).contiguous() for v in range(3)]
host = [t.cpu().numpy() for t in u8]
want2 = pipeline.run_two_view_u8(u8[0], u8[1], nets, device=dev)[0].cpu().numpy()
want3 = pipeline.run_three_view_u8(u8[0], u8[1], u8[2], nets, device=dev)[0].cpu().numpy()
bad = 0
for chunk in (32, 7, 16, 75):
    ref2 = ref3 = None
    for r in range(rounds):
        g2 = pipeline.run_two_view_long(host[0], host[1], nets, device=dev, chunk=chunk)[0]
        g3 = pipeline.run_three_view_long(host[0], host[1], host[2], nets, device=dev, chunk=chunk)[0]
        if ref2 is None:
            ref2, ref3 = g2.copy(), g3.copy()
            if chunk == 32:
                ok = np.array_equal(g2, want2) and np.array_equal(g3, want3)
                print('chunk 32 equals the resident clip: %s' % ok); bad += not ok
            else:
                def cmp(a, b):
                    if a.shape != b.shape:
                        return 'canvas differs %s vs %s' % (a.shape[1:3], b.shape[1:3])
                    d = np.abs(a.astype(np.int16) - b)
                    return 'bytes differing %.2e (by > 1: %.2e, max %d)' % ((d > 0).mean(), (d > 1).mean(), d.max())
                print('chunk %2d vs resident (chunk 32; kernel variants differ with the batch, fp32 rounding moves a view\'s border '
                      'pixels): 2-view %s; 3-view %s' % (chunk, cmp(g2, want2), cmp(g3, want3)))
        else:
            ok = np.array_equal(g2, ref2) and np.array_equal(g3, ref3)
            bad += not ok
            print('chunk %2d round %d repeatable: %s' % (chunk, r, ok))
print('FAILED' if bad else 'all repeatable')
sys.exit(1 if bad else 0)
