"""Per-workgroup phase timeline of one conv launch (s_memtime stamps written through ss_debug_ptr)."""
import sys, os, torch, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabstitch2_amd import ops, _hip
import _tuning
lib = _tuning.lib()
dev = torch.device('cuda:0')
SHAPES = {'l2s2': (64, 90, 120, 64, 128, 3, 2, 1), 'l3s2': (64, 45, 60, 128, 256, 3, 2, 1), 'layer1': (64, 90, 120, 64, 64, 3, 1, 1), 'layer2': (64, 45, 60, 128, 128, 3, 1, 1), 'layer3': (64, 23, 30, 256, 256, 3, 1, 1)}
for name in sys.argv[1].split(','):
    if name == 'stem':          # the 7x7/2 stem on the row-packed 3-channel layout (two trunks' filters: cout 128)
        xs = [torch.randn(16, 3, 360, 480, device=dev)]
        buf = ops.stem_input(xs); wt = torch.randn(128, 7, 24, device=dev) * 0.05; b = torch.randn(128, device=dev)
        run = lambda: ops.conv_stem(buf, wt, b, relu=True)
        out = run(); cout = 128; m = out.numel() // cout
        nblk = ((m + 127) // 128) * ((cout + 63) // 64) + 64
    else:
        n, h, w, cin, cout, k, s, p = SHAPES[name]
        x = torch.randn(n, h, w, cin, device=dev); wt = torch.randn(cout, 1, k, k, cin, device=dev) * 0.05; b = torch.randn(cout, device=dev)
        out = ops.conv(x, wt, b, stride=s, pad=(0, p, p), relu=True)
        res = torch.randn_like(out)
        run = lambda: ops.conv(x, wt, b, res=res, stride=s, pad=(0, p, p), relu=True, out=out)
        m = out.numel() // cout
        nblk = ((m + 63) // 64) * ((cout + 63) // 64)
    dbg = torch.zeros((nblk, 8), dtype=torch.int64, device=dev)
    for _ in range(3): run()
    torch.cuda.synchronize()
    lib.ss_debug_ptr(ctypes.c_void_p(dbg.data_ptr()))
    run()
    torch.cuda.synchronize()
    lib.ss_debug_ptr(None)
    d = dbg.cpu().numpy().astype(np.int64)
    d = d[d[:, 0] > 0]; nblk = len(d)
    t0 = d[:, 0].min()
    pro, loop, epi = d[:, 1] - d[:, 0], d[:, 2] - d[:, 1], d[:, 3] - d[:, 2]
    total = d[:, 3].max() - t0
    hw = d[:, 4] & 0xFFFFFFFF; xcc = (d[:, 4] >> 32) & 0xF
    cu = ((hw >> 8) & 0xF) | (((hw >> 13) & 0x7) << 4) | (((hw >> 12) & 1) << 7) | (xcc << 8)
    print('%s: %d blocks, kernel span %d ticks; per block median ticks: prologue %d  loop %d  epilogue %d  (sum %d)' % (
        name, nblk, total, np.median(pro), np.median(loop), np.median(epi), np.median(d[:, 3] - d[:, 0])))
    print('   p10/p90: prologue %d/%d  loop %d/%d  epilogue %d/%d' % (np.percentile(pro, 10), np.percentile(pro, 90),
          np.percentile(loop, 10), np.percentile(loop, 90), np.percentile(epi, 10), np.percentile(epi, 90)))
    print('   prologue split (median ticks): row setup %d  table build+barrier %d  first loads issued %d  wait+store+barrier %d' % (
        np.median(d[:, 5] - d[:, 0]), np.median(d[:, 6] - d[:, 5]), np.median(d[:, 7] - d[:, 6]), np.median(d[:, 1] - d[:, 7])))
    ucu = np.unique(cu)
    print('   distinct CU ids seen: %d; blocks per CU min/median/max: %s' % (len(ucu), np.percentile(np.bincount(np.searchsorted(ucu, cu)), [0, 50, 100])))
    # concurrency on one CU: how many blocks are in each phase at sampled times
    c0 = ucu[len(ucu) // 2]
    sel = d[cu == c0]
    ts = np.linspace(sel[:, 0].min(), sel[:, 3].max(), 400)
    inpro = [(np.sum((sel[:, 0] <= t) & (t < sel[:, 1]))) for t in ts]
    inloop = [(np.sum((sel[:, 1] <= t) & (t < sel[:, 2]))) for t in ts]
    inepi = [(np.sum((sel[:, 2] <= t) & (t < sel[:, 3]))) for t in ts]
    print('   CU %d: %d blocks; time-avg resident blocks in prologue %.2f  loop %.2f  epilogue %.2f; fraction of time with <2 blocks in loop: %.3f'
          % (c0, len(sel), np.mean(inpro), np.mean(inloop), np.mean(inepi), np.mean(np.array(inloop) < 2)))
    starts = np.sort(sel[:, 0] - t0)
    print('   CU %d block start ticks (first 24): %s' % (c0, starts[:24].tolist()))
