import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabstitch2_amd import ops, _hip
dev = torch.device('cuda:0')
for ty, n, r in ((4, 32, 5), (8, 32, 5), (4, 62, 3), (8, 62, 3), (4, 32, 5), (8, 32, 5)):
    _hip.lib().ss_cost_volume_set_tile(ty)
    a = torch.randn(n, 45, 60, 128, device=dev); b = torch.randn(n, 45, 60, 128, device=dev)
    out = ops.cost_volume(a, b, r); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(50): ops.cost_volume(a, b, r, out=out)
    e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / 50
    fl = 2.0 * n * 2700 * (2 * r + 1) ** 2 * 128
    print('cost volume TY=%d n=%d r=%d: %.1f us  %.1f TFLOP/s' % (ty, n, r, ms * 1e3, fl / ms / 1e9))
