"""Per-workgroup phase timeline of one F(4x4,3x3) launch (s_memtime stamps, tuning build).   python tools/diag_wino43.py layer2,layer1 [ablation masks, e.g. 0,1,2,4,8,15,16]"""
import sys, os, torch, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabstitch2_amd import ops, _hip
import _tuning
lib = _tuning.lib()
dev = torch.device('cuda:0')
SHAPES = {'layer1': (64, 90, 120, 64, 64), 'layer2': (64, 45, 60, 128, 128), 'layer3': (64, 23, 30, 256, 256)}
ABL = [int(a) for a in sys.argv[2].split(',')] if len(sys.argv) > 2 else [0]
for name, abl in ((n_, a_) for n_ in sys.argv[1].split(',') for a_ in ABL):
    lib.ss_debug_set(21, abl)
    n, h, w, cin, cout = SHAPES[name]
    x = torch.randn(n, h, w, cin, device=dev); wt = torch.randn(cout, 1, 3, 3, cin, device=dev) * 0.05; b = torch.randn(cout, device=dev)
    out = ops.conv_winograd43(x, wt, b, None, relu=True)
    res = torch.randn_like(out)
    for use_res in ((False,) if abl else (False, True)):
        r = res if use_res else None
        for _ in range(3): ops.conv_winograd43(x, wt, b, r, relu=True, out=out)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): ops.conv_winograd43(x, wt, b, r, relu=True, out=out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        dbg = torch.zeros((1 << 16, 12), dtype=torch.int64, device=dev)
        lib.ss_debug_ptr(ctypes.c_void_p(dbg.data_ptr()))
        ops.conv_winograd43(x, wt, b, r, relu=True, out=out)
        torch.cuda.synchronize()
        lib.ss_debug_ptr(None)
        d = dbg.cpu().numpy().astype(np.int64)
        d = d[d[:, 0] > 0]
        med = lambda a: int(np.median(a))
        hw = d[:, 10] & 0xFFFFFFFF; xcc = (d[:, 10] >> 32) & 0xF
        cu = ((hw >> 8) & 0xF) | (((hw >> 13) & 0x7) << 4) | (((hw >> 12) & 1) << 7) | (xcc << 8)
        spans, busy, cnt = [], [], []
        for c in np.unique(cu):
            sel = d[cu == c]
            spans.append(sel[:, 5].max() - sel[:, 0].min()); busy.append((sel[:, 5] - sel[:, 0]).sum()); cnt.append(len(sel))
        rt = (d[:, 11].max() - d[:, 11].min())        # s_memrealtime ticks (100 MHz) between the first and the last workgroup's end
        print('   %d CUs ran %d..%d workgroups; per CU: span (first start -> last end) median %d max %d ticks, workgroups resident %.3f of the span; launch: first start -> last end %d ticks; realtime ticks between first and last end %d'
              % (len(spans), min(cnt), max(cnt), int(np.median(spans)), int(np.max(spans)), float(np.sum(busy)) / float(np.sum(spans)), int(d[:, 5].max() - d[:, 0].min()), int(rt)))
        print('%s abl=%d res=%d: %.1f us per launch | %d blocks; median ticks: prologue %d (setup %d, loads + row transform %d, barrier %d) | first / later rounds prologue %d / %d | K loop %d (%d chunks: %d per chunk) | wait %d, dump 0 %d, barrier %d | combine 0 + dump 1 %d | combine 1 %d | total %d | launch span %d'
              % (name, abl, use_res, ms * 1e3, len(d), med(d[:, 1] - d[:, 0]), med(d[:, 8] - d[:, 0]), med(d[:, 9] - d[:, 8]), med(d[:, 1] - d[:, 9]), med((d[:, 1] - d[:, 0])[d[:, 0] < np.sort(d[:, 0])[255]]), med((d[:, 1] - d[:, 0])[d[:, 0] >= np.sort(d[:, 0])[256]]), med(d[:, 2] - d[:, 1]), cin // 16, med(d[:, 2] - d[:, 1]) // (cin // 16),
                 med(d[:, 6] - d[:, 2]), med(d[:, 7] - d[:, 6]), med(d[:, 3] - d[:, 7]), med(d[:, 4] - d[:, 3]), med(d[:, 5] - d[:, 4]), med(d[:, 5] - d[:, 0]), d[:, 5].max() - d[:, 0].min()), flush=True)
