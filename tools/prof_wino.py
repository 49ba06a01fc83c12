"""Isolated launches of the Winograd and the implicit-GEMM kernels on the trunk's layer shapes, for rocprofv3 PMC passes:
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d DIR -- python tools/prof_wino.py"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabstitch2_amd import ops
dev = torch.device('cuda:0')
SHAPES = {'layer1': (64, 90, 120, 64, 64), 'layer2': (64, 45, 60, 128, 128), 'layer3': (64, 23, 30, 256, 256)}
for name, (n, h, w, cin, cout) in SHAPES.items():
    x = torch.randn(n, h, w, cin, device=dev); wt = torch.randn(cout, 1, 3, 3, cin, device=dev) * 0.05; b = torch.randn(cout, device=dev)
    out = ops.conv_winograd(x, wt, b, None, relu=True)
    res = torch.randn_like(out)
    for _ in range(6):
        ops.conv_winograd(x, wt, b, res, relu=True, out=out)
    ops.WINOGRAD = False
    for _ in range(6):
        ops.conv(x, wt, b, res=res, stride=1, pad=(0, 1, 1), relu=True, out=out)
    ops.WINOGRAD = True
    torch.cuda.synchronize()
