"""Per-layer rounding error of the conv engine's three 3x3 kernels under both synthetic checkpoints.

    python tools/wino43_trained_like.py [default|trained_like] [frames]

Runs SpatialNet's trunk + regressors on a clip with the F(4x4,3x3) kernel off, captures the operands of every launch the
F(4x4,3x3) kernel could take (3x3, stride 1, cin % 16 == 0, cout % 64 == 0), and recomputes each with F(4x4,3x3), F(2x2,3x3) and the
implicit GEMM against an fp64 convolution of the same operands.  Columns: max|y|; max error / max|y| per kernel; and the worst
CHANNEL-relative error (max error in a channel / that channel's max|y|) -- the quiet channels of a BN-folded layer are where an
error that scales with the loud ones shows."""
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, '.')
from stabstitch2_amd import ops, synth                      # noqa: E402
from stabstitch2_amd.spatial_network import SpatialNet, build_SpatialNet   # noqa: E402

torch.set_grad_enabled(False)
profile = sys.argv[1] if len(sys.argv) > 1 else 'trained_like'
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device('cuda:0')
sp = SpatialNet()
sp.load_state_dict(synth.synthetic_state_dict(sp, profile=profile), strict=True)
sp = sp.to(dev)
_, lr = synth.make_clip(frames, 360, 480, seed=5)
lr1 = torch.cat(lr[0], 0).to(dev)
lr2 = torch.cat(lr[1], 0).to(dev)

captured = []
real_conv, real_grouped = ops.conv, ops.conv_grouped


def eligible(x, wgt, stride, pad):
    return wgt.dim() == 5 and x.dim() == 4 and tuple(wgt.shape[1:4]) == (1, 3, 3) and stride == 1 and tuple(pad) == (0, 1, 1) \
        and wgt.shape[-1] % 16 == 0 and wgt.shape[0] % 64 == 0


def conv(x, wgt, bias=None, res=None, stride=1, pad=(0, 1, 1), relu=False, out=None, pool2=False):
    if eligible(x, wgt, stride, pad) and not pool2:
        captured.append((x.clone(), wgt, bias, None if res is None else res.clone(), relu))
    return real_conv(x, wgt, bias, res, stride, pad, relu, out, pool2)


ops.conv = conv
ops.WINO43 = '0'
build_SpatialNet(sp, lr1, lr2)
ops.conv = real_conv

print('profile %s, %d images; eligible 3x3 launches: %d' % (profile, 2 * frames, len(captured)))
print('%-4s %-22s %10s | %10s %10s %10s | %10s %10s %10s' % ('#', 'geometry', 'max|y|', 'F43/max', 'F22/max', 'igemm/max',
                                                            'F43 ch-rel', 'F22 ch-rel', 'igemm ch'))
for i, (x, wgt, bias, res, relu) in enumerate(captured):
    n, h, w, cin = x.shape
    cout = wgt.shape[0]
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), wgt[:, 0].permute(0, 3, 1, 2).double(),
                   None if bias is None else bias.double(), padding=1).permute(0, 2, 3, 1)
    if res is not None:
        ref = ref + res.double()
    if relu:
        ref = torch.relu(ref)
    ymax = float(ref.abs().max())
    chmax = ref.abs().amax(dim=(0, 1, 2)).clamp_min(1e-30)
    row = []
    rel = []
    for fn in (ops.conv_winograd43, ops.conv_winograd, None):
        if fn is None:
            out = torch.empty((n, h, w, cout), device=dev)
            ws = ops.conv_workspace(dev, ops._conv_ws_need(n, 1, h, w, cin, cout, 1, 3, 3, 1, 0, 1, 1, 1))
            ops.H.call('ss_conv_nhwc', ops.H.dptr(x), ops.H.dptr(wgt), ops.H.dptr(bias, True), ops.H.dptr(res, True), ops.H.dptr(out),
                       n, 1, h, w, cin, cout, 1, 3, 3, 1, 0, 1, 1, int(relu), cout, 1, 0, 0, 0, ops.H.dptr(ws, True),
                       0 if ws is None else ws.numel(), ops.H.stream())
            y = out
        else:
            y = fn(x, wgt, bias, res, relu)
        e = (y.double() - ref).abs()
        row.append(float(e.max()) / ymax)
        rel.append(float((e.amax(dim=(0, 1, 2)) / chmax).max()))
    print('%-4d %-22s %10.3e | %10.2e %10.2e %10.2e | %10.2e %10.2e %10.2e' %
          (i, '%dx%dx%d %d->%d%s' % (n, h, w, cin, cout, '+res' if res is not None else ''), ymax, *row, *rel))
