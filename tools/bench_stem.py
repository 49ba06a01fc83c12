"""The fused stem (ss_stem_pool: conv 7x7/2 + BN + ReLU + max-pool in one kernel) against conv_stem + maxpool_split, 64 images of
360x480, both trunks' filter banks (the shared-stem launch of a 32-frame 2-view clip).      python tools/bench_stem.py"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabstitch2_amd import ops, layers as L
dev = torch.device('cuda:0')
torch.manual_seed(0)
n = 64
x = torch.randn(n, 3, 360, 480, device=dev)
w = torch.zeros(128, 7, 24, device=dev); w[:, :, :21] = torch.randn(128, 7, 21, device=dev) * 0.1
b = torch.randn(128, device=dev)
buf = ops.stem_input(x)
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
def two():
    old = ops.STEM_FUSED; ops.STEM_FUSED = False
    try: return L.run_stem_shared([x], (w, b))
    finally: ops.STEM_FUSED = old
flop = 2.0 * n * 180 * 240 * 128 * 147
for rnd in range(3):
    a = t(lambda: ops.stem_pool(buf, w, b)); c = t(two)
    print('fused %.1f us (%.1f TF/s on the conv\'s 147-tap flop; %.1f executed incl. halo + K padding)   conv_stem + maxpool_split (16-image sub-chunks, incl. layout kernel) %.1f us'
          % (a, flop / a / 1e6, flop / a / 1e6 * 512 / 432 * 154 / 147, c))
