"""Stem conv (7x7/2, row-packed 3-channel layout, two trunks' 128 filters, 16 images) under forced tile variants
(tuning build, ss_debug_set(0, key)): 0 = the dispatch rule (128x64, peeled partial K tile), 5 = 128x128, 7 = 128x64
without the peeled tile, 16 = 64x128."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabstitch2_amd import ops
import _tuning
lib = _tuning.lib()
dev = torch.device('cuda:0')
xs = [torch.randn(16, 3, 360, 480, device=dev)]
buf = ops.stem_input(xs); wt = torch.randn(128, 7, 24, device=dev) * 0.05; b = torch.randn(128, device=dev)
ref = None
for key in (0, 16):
    lib.ss_debug_set(0, key)
    for _ in range(3): out = ops.conv_stem(buf, wt, b, relu=True)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): out = ops.conv_stem(buf, wt, b, relu=True)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 30 * 1e3
    if ref is None: ref = out.clone()
    print('tile key %2d: %.1f us  (%.1f TF/s on 147 real products)  max|diff vs rule| %.3g' % (key, us, 2.0 * out.numel() * 147 / us / 1e6, (out - ref).abs().max().item()))
lib.ss_debug_set(0, 0)
for ab in (0, 64, 0, 64):
    lib.ss_debug_set(1, ab)
    for _ in range(3): out = ops.conv_stem(buf, wt, b, relu=True)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): out = ops.conv_stem(buf, wt, b, relu=True)
    e1.record(); torch.cuda.synchronize()
    print('ablation %d: %.1f us' % (ab, e0.elapsed_time(e1) / 30 * 1e3))
lib.ss_debug_set(1, 0)
