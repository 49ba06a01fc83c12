"""Parity of the HIP path (through the C ABI) against the CPU oracle and the reference-generated
golden fixtures.  Runs on the MI355X box:  python -m pytest tests -m gpu"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import cases
from oracle import geometry as G, samplers as S, nets as N, pipeline as P, metrics as M
from stabstitch2_amd import synth

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'gpu tests need a GPU'
    return torch.device('cuda:0')


def close_boxes(got, ref, iqr, tol, what='', k=16, cover=0.6):
    """box medians compared where the golden box holds no discontinuity (cases.smooth_boxes)."""
    ok = cases.smooth_boxes(iqr, k)
    assert ok.mean() > cover, (what, ok.mean())
    return close(np.where(ok, got, 0.0), np.where(ok, ref, 0.0), tol, what)


def close(a, b, tol, what=''):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64)))) if a.size else 0.0
    assert np.isfinite(a).all(), what + ' has non-finite values'
    if os.environ.get('SS_VERBOSE'):
        print('  [close] %-40s max|diff| %.3e  (tol %.1e)' % (what, err, tol))
    assert err <= tol, '%s max|diff| %.3e > %.1e' % (what, err, tol)
    return err


def close_grad(a, b, tol_px, base, what=''):
    """Image comparison with a gradient-aware bound: |a-b| <= base + tol_px * G, where G is the largest jump to a
    4-neighbour in the reference image b (a sampling-coordinate error of tol_px moves the value by at most ~tol_px*G;
    at zero-padded borders G is the full edge step)."""
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    bp = np.pad(b, [(0, 0)] * (b.ndim - 2) + [(1, 1), (1, 1)], mode='edge')
    g = np.zeros_like(b)
    for dy, dx in ((0, 1), (2, 1), (1, 0), (1, 2)):
        g = np.maximum(g, np.abs(bp[..., dy:dy + b.shape[-2], dx:dx + b.shape[-1]] - b))
    excess = np.abs(a.astype(np.float64) - b) - (base + tol_px * g)
    if os.environ.get('SS_VERBOSE'):
        print('  [close_grad] %-35s max|diff| %.3e  worst excess %.3e' % (what, np.abs(a - b).max(), excess.max()))
    assert excess.max() <= 0, '%s exceeds base %.1e + %.1e px * gradient by %.3e' % (what, base, tol_px, excess.max())


# ------------------------------------------------------------------ conv engine
def _pack(w, cin_pad):
    cout, cin = w.shape[:2]
    if w.dim() == 4:
        out = torch.zeros(cout, 1, w.shape[2], w.shape[3], cin_pad)
        out[:, 0, :, :, :cin] = w.permute(0, 2, 3, 1)
    else:
        out = torch.zeros(cout, w.shape[2], w.shape[3], w.shape[4], cin_pad)
        out[..., :cin] = w.permute(0, 2, 3, 4, 1)
    return out.contiguous()


@pytest.mark.parametrize('n,cin,cout,h,w,k,s,p,bias,res,relu', [
    (2, 3, 64, 72, 96, 7, 2, 3, True, False, True),      # conv1 shape class (cin padded 3->4, K tail)
    (1, 64, 64, 90, 120, 3, 1, 1, True, True, True),     # layer1 block tail
    (3, 64, 128, 45, 60, 3, 2, 1, True, False, True),    # strided
    (3, 64, 128, 45, 60, 1, 2, 0, True, False, False),   # downsample 1x1
    (2, 121, 64, 45, 60, 3, 1, 1, False, False, True),   # cost-volume regressor (121 -> 124 ch)
    (5, 128, 256, 5, 7, 3, 1, 1, False, False, True),    # tiny map, M tail
    (16, 64, 64, 90, 120, 3, 1, 1, True, True, True),    # large M -> 128-row tiles
    (16, 128, 128, 45, 60, 3, 1, 1, True, False, True),  # 128x128 tiles
])
def test_conv2d(dev, n, cin, cout, h, w, k, s, p, bias, res, relu):
    from stabstitch2_amd import ops
    rs = np.random.RandomState(n * 1000 + cin + cout)
    x = torch.from_numpy(rs.normal(0, 1, (n, cin, h, w)).astype(np.float32))
    wt = torch.from_numpy((rs.normal(0, 1, (cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32))
    b = torch.from_numpy(rs.normal(0, 1, cout).astype(np.float32)) if bias else None
    ref = F.conv2d(x, wt, b, stride=s, padding=p)
    r = torch.from_numpy(rs.normal(0, 1, tuple(ref.shape)).astype(np.float32)) if res else None
    if res:
        ref = ref + r
    if relu:
        ref = F.relu(ref)
    cp = (cin + 3) // 4 * 4
    xd = ops.nchw_to_nhwc(x.to(dev), cp)
    rd = ops.nchw_to_nhwc(r.to(dev)) if res else None
    out = ops.conv(xd, _pack(wt, cp).to(dev), b.to(dev) if bias else None, rd, stride=s, pad=(0, p, p), relu=relu)
    close(ops.nhwc_to_nchw(out), ref, 2e-5 * max(1.0, float(ref.abs().max())), 'conv2d')


def test_conv3d(dev):
    from stabstitch2_amd import ops
    rs = np.random.RandomState(7)
    x = torch.from_numpy(rs.normal(0, 1, (3, 128, 7, 7, 9)).astype(np.float32))
    wt = torch.from_numpy((rs.normal(0, 1, (128, 128, 5, 3, 3)) / np.sqrt(128 * 45)).astype(np.float32))
    b = torch.from_numpy(rs.normal(0, 1, 128).astype(np.float32))
    ref = F.relu(F.conv3d(x, wt, b, padding=(2, 1, 1)))
    out = ops.conv(x.permute(0, 2, 3, 4, 1).contiguous().to(dev), _pack(wt, 128).to(dev), b.to(dev), None,
                   stride=1, pad=(2, 1, 1), relu=True)
    close(out.permute(0, 4, 1, 2, 3), ref, 5e-5, 'conv3d')


def test_pool_linear_layout(dev):
    from stabstitch2_amd import ops
    rs = np.random.RandomState(8)
    x = torch.from_numpy(rs.normal(0, 1, (2, 64, 45, 61)).astype(np.float32))
    xd = ops.nchw_to_nhwc(x.to(dev))
    close(ops.nhwc_to_nchw(xd), x, 0, 'layout round trip')
    close(ops.nhwc_to_nchw(ops.maxpool(xd, 3, 2, 1)), F.max_pool2d(x, 3, 2, 1), 0, 'maxpool 3/2/1')
    close(ops.nhwc_to_nchw(ops.maxpool(xd, 2, 2, 0)), F.max_pool2d(x, 2, 2), 0, 'maxpool 2/2 floor')
    for m in (1, 3, 19):
        a = torch.from_numpy(rs.normal(0, 1, (m, 1536)).astype(np.float32))
        w = torch.from_numpy((rs.normal(0, 1, (1024, 1536)) / 39.0).astype(np.float32))
        b = torch.from_numpy(rs.normal(0, 1, 1024).astype(np.float32))
        close(ops.linear(a.to(dev), w.to(dev), b.to(dev), relu=True), F.relu(F.linear(a, w, b)), 5e-5, 'linear')
    a = torch.from_numpy(rs.normal(0, 1, (5, 2)).astype(np.float32))
    w = torch.from_numpy(rs.normal(0, 1, (32, 2)).astype(np.float32))
    close(ops.linear(a.to(dev), w.to(dev), None), F.linear(a, w), 1e-6, 'linear k=2')


# ------------------------------------------------------------------ correlation
def test_cost_volume(dev, golden):
    from stabstitch2_amd import ops
    g = golden('g3_costvol')
    a, b = cases.g3_inputs(False)
    for r, key in ((5, 'cv5'), (3, 'cv3')):
        d = (2 * r + 1) ** 2
        out = ops.cost_volume(ops.nchw_to_nhwc(a.to(dev)), ops.nchw_to_nhwc(b.to(dev)), r)
        assert float(out[..., d:].abs().max()) == 0.0
        close(ops.nhwc_to_nchw(out, d), g[key], 1e-5, 'cost volume r=%d vs reference' % r)
    fa, fb = cases.g3_inputs(True)
    for r, rows, chs, row in ((5, 'full5_rows', 'full5_chsum', 22), (3, 'full3_rows', 'full3_chsum', 0)):
        d = (2 * r + 1) ** 2
        out = ops.nhwc_to_nchw(ops.cost_volume(ops.nchw_to_nhwc(fa.to(dev)), ops.nchw_to_nhwc(fb.to(dev)), r), d)
        close(out[0, :, row, :], g[rows], 1e-5, 'full rows')
        close(out.sum(dim=(2, 3)), g[chs], 2e-3, 'full channel sums')
        close(out, N.cost_volume(fa, fb, r), 1e-5, 'full vs oracle')
    # direction symmetry as a size-independent property on a batch
    x1 = torch.randn(3, 45, 60, 128, device=dev)
    x2 = torch.randn(3, 45, 60, 128, device=dev)
    c12 = ops.cost_volume(x1, x2, 5)[..., :121].view(3, 45, 60, 11, 11)
    c21 = ops.cost_volume(x2, x1, 5)[..., :121].view(3, 45, 60, 11, 11)
    assert torch.allclose(c12[:, 10, 10, 7, 8], c21[:, 12, 13, 3, 2], atol=1e-6)


def test_ccl(dev, golden):
    from stabstitch2_amd import ops
    g = golden('g4_ccl')
    for full, key in ((False, 'flow'), (True, 'flow_full')):
        a, b = cases.g4_inputs(full)
        flow, flow4 = ops.ccl(ops.nchw_to_nhwc(a.to(dev)), ops.nchw_to_nhwc(b.to(dev)), 10.0)
        close(flow, g[key], 1e-4, 'ccl vs reference')
        close(flow4[..., :2].permute(0, 3, 1, 2), g[key], 1e-4, 'ccl nhwc4')
        assert float(flow4[..., 2:].abs().max()) == 0.0


# ------------------------------------------------------------------ geometry
def test_dlt_decomposition_meshes(dev, golden):
    from stabstitch2_amd import ops
    from stabstitch2_amd.utils import torch_DLT
    g = golden('g1_dlt')
    off = cases.g1_offsets()
    b = off.shape[0]
    c = torch.tensor([[0., 0.], [480., 0.], [0., 360.], [480., 360.]]).unsqueeze(0).expand(b, -1, -1)
    H = torch_DLT.tensor_DLT(c.to(dev), (c + off.reshape(b, 4, 2)).to(dev)).cpu()
    pts = torch.tensor([[0., 0., 1.], [480., 0., 1.], [0., 360., 1.], [480., 360., 1.], [240., 180., 1.]]).T
    pa, pb = H @ pts, torch.from_numpy(g['H_full']) @ pts
    close(pa[:, :2] / pa[:, 2:3], pb[:, :2] / pb[:, 2:3], 5e-3, 'DLT action on corners (px)')
    zero = torch.zeros(b, 126, device=dev)
    m1, m2 = ops.spatial_meshes(off.to(dev), zero, zero, 360, 480)
    rigid = torch.from_numpy(g['rigid'])
    close(m1.cpu() + rigid, g['mesh_ref'], 5e-2, 'mesh_ref vs reference')
    close(m2.cpu() + rigid, g['mesh_tgt'], 5e-2, 'mesh_tgt vs reference')
    # against an fp64 evaluation of the same algebra the device solve must be far tighter
    _, Ht, Hr = G.decompose(off.double(), 360, 480, 1.0)
    close(m1.cpu() + rigid, G.homography_to_mesh(Hr, rigid.double()).float(), 2e-4, 'mesh_ref vs fp64')
    th_ref, th_tgt = ops.spatial_decompose(off.to(dev), 360, 480)
    _, Ht8, Hr8 = G.decompose(off.double(), 360, 480, 8.0)
    Mm = torch.tensor([[30., 0., 30.], [0., 22.5, 22.5], [0., 0., 1.]], dtype=torch.float64)
    close(th_ref, (torch.inverse(Mm) @ Hr8 @ Mm).float(), 2e-5, 'theta_ref')
    close(th_tgt, (torch.inverse(Mm) @ Ht8 @ Mm).float(), 2e-5, 'theta_tgt')


def test_homography_sampler(dev, golden):
    from stabstitch2_amd import ops
    from stabstitch2_amd.utils import torch_homo_transform
    g = golden('g2_homo')
    U, th = cases.g2_inputs()
    close(torch_homo_transform.transformer(U.to(dev), th.to(dev), (45, 60)), g['out'], 1e-4, 'homo nchw')
    close(torch_homo_transform.transformer(U.to(dev), th.to(dev), (23, 31)), g['out_small'], 1e-4, 'homo nchw small')
    out = ops.homo_warp_nhwc(ops.nchw_to_nhwc(U.to(dev)), th.to(dev), 45, 60)
    close(ops.nhwc_to_nchw(out), g['out'], 1e-4, 'homo nhwc')


def test_tps_points_and_tsmotion(dev, golden):
    from stabstitch2_amd import ops
    from stabstitch2_amd.utils import torch_tps_transform_point
    g = golden('g5_tps_points')
    nrigid, warped, query = [t.to(dev) for t in cases.g5_meshes()]
    close(torch_tps_transform_point.transformer(query, nrigid, warped), g['p_a'], 1e-5, 'tps points a')
    close(torch_tps_transform_point.transformer(query, warped, nrigid), g['p_b'], 1e-5, 'tps points b')
    # interpolation property: the spline maps its own control points onto the targets
    close(torch_tps_transform_point.transformer(warped, warped, nrigid), nrigid, 2e-5, 'tps interpolation')
    g8 = golden('g8_nets')
    smesh, tsm = ops.tsmotion(torch.from_numpy(g8['motion1']).to(dev), torch.from_numpy(g8['tmotion1']).to(dev))
    close(tsm, g8['tsmotion1'], 2e-3, 'tsmotion vs reference')


def test_tps_dense_warp_and_fusion(dev, golden):
    from stabstitch2_amd import ops, pipeline
    from stabstitch2_amd.utils import torch_tps_transform
    g = golden('g6_tps_warp')
    U, src, tgt, size, ident = cases.g6_inputs()
    Ud, sd, td = U.to(dev), src.to(dev), tgt.to(dev)
    wn = torch_tps_transform.transformer(Ud, sd, td, size, 'NORMAL')
    wf = torch_tps_transform.transformer(Ud, sd, td, size, 'FAST')
    close(wn[:, 3:5], g['normal'][:, 3:5], 2.5e-3, 'coords NORMAL (px)')          # observed 2.4e-4 px
    close(wn[:, 0:3], g['normal'][:, 0:3], 2e-3, 'intensity NORMAL')
    close_grad(wf, g['fast'], 5e-3, 2e-3, 'FAST')
    close(torch_tps_transform.transformer(Ud, ident.to(dev), td, (72, 96), 'NORMAL'), g['ident_normal'], 2e-3, 'id N')
    close_grad(torch_tps_transform.transformer(Ud, ident.to(dev), td, (72, 96), 'FAST'), g['ident_fast'], 5e-3, 2e-3,
               'id F')
    g7 = golden('g7_fusion')
    T = ops.tps_solve(sd, td)
    wm = ops.tps_warp(Ud[:, 0:3].contiguous(), sd, T, size[0], size[1], 'NORMAL', with_mask=True)
    close(wm, g7['warped_with_mask'], 2e-3, 'warp + ones mask')
    fused = ops.render_average([Ud[0, 0:3].contiguous(), Ud[1, 0:3].contiguous()], sd, T, size[0], size[1], 'NORMAL')
    close_grad(fused, g7['average'], 5e-3, 2e-3, 'fused AVERAGE render')
    gm = torch.from_numpy(g7['warped_with_mask']).to(dev)
    lin = pipeline.linear_blender(gm[0:1, 0:3], gm[1:2, 0:3], gm[0:1, 3:4], gm[1:2, 3:4])
    close(lin, g7['linear'], 1e-3, 'LINEAR fusion')
    mk = pipeline.linear_blender(gm[0:1, 0:3], gm[1:2, 0:3], gm[0:1, 3:4], gm[1:2, 3:4], mask=True)
    close(mk, g7['mask1'], 1e-5, 'LINEAR mask1')


# ------------------------------------------------------------------ nets / pipeline
@pytest.fixture(scope='module')
def hip_nets(dev):
    from stabstitch2_amd.spatial_network import SpatialNet
    from stabstitch2_amd.temporal_network import TemporalNet
    from stabstitch2_amd.smooth_network import SmoothNet
    nets = []
    for cls in (SpatialNet, TemporalNet, SmoothNet):
        m = cls()
        m.load_state_dict(synth.synthetic_state_dict(m), strict=True)
        nets.append(m.to(dev))
    return nets


@pytest.fixture(scope='module')
def clip16():
    return synth.make_clip(16, 360, 480, seed=0)


def test_nets_vs_reference(dev, golden, hip_nets, clip16):
    from stabstitch2_amd.spatial_network import build_SpatialNet
    from stabstitch2_amd.temporal_network import build_TemporalNet
    from stabstitch2_amd.smooth_network import build_SmoothNet
    assert [len(m.state_dict()) for m in hip_nets] == [130, 104, 14]
    g = golden('g8_nets')
    sp, tp, sm = hip_nets
    _, lr = clip16
    o1, o2r, o2t = sp(lr[0][0].to(dev), lr[1][0].to(dev))
    close(o1, g['offset_1'], 1e-4, 'offset_1')                 # gates = ~10x the observed deviations (4e-6 px)
    close(o2r, g['offset_2_ref'], 1e-4, 'offset_2_ref')
    close(o2t, g['offset_2_tgt'], 1e-4, 'offset_2_tgt')
    lr1 = torch.cat(lr[0], 0).to(dev)
    lr2 = torch.cat(lr[1], 0).to(dev)
    o = build_SpatialNet(sp, lr1, lr2)                      # whole clip as one batch
    close(o['motion1'], g['motion1'], 5e-3, 'motion1')         # observed 3-5e-4 px (the fp64 DLT vs the reference's fp32 inverse)
    close(o['motion2'], g['motion2'], 5e-3, 'motion2')
    tm = build_TemporalNet(tp, [f.to(dev) for f in lr[0]])['motion_list']
    close(torch.cat(tm, 0), g['tmotion1'], 1e-4, 'tmotion1')
    rigid = torch.from_numpy(cases.rigid(360, 480)).to(dev)
    ts1 = [torch.from_numpy(g['tsmotion1'][i:i + 1]).to(dev) for i in range(7)]
    ts2 = [torch.from_numpy(g['tsmotion2'][i:i + 1]).to(dev) for i in range(7)]
    ts1[0] = ts1[0] * 0
    ts2[0] = ts2[0] * 0
    sm1 = [rigid + torch.from_numpy(g['motion1'][i:i + 1]).to(dev) for i in range(7)]
    sm2 = [rigid + torch.from_numpy(g['motion2'][i:i + 1]).to(dev) for i in range(7)]
    w0 = build_SmoothNet(sm, ts1, ts2, sm1, sm2)
    assert sorted(w0) == sorted(k[3:] for k in g.files if k.startswith('w0_'))
    for k, v in w0.items():
        close(v, g['w0_' + k], 5e-4, 'window0 ' + k)             # observed 3e-5


def test_pipeline_vs_reference(dev, golden, hip_nets, clip16):
    from stabstitch2_amd import pipeline
    g = golden('g9_pipeline')
    hr, lr = clip16
    acc = pipeline.estimate_meshes(hip_nets, lr[0], lr[1])
    close(acc['smooth_mesh1'], g['smooth_mesh1'], 5e-3, 'smooth_mesh1')      # observed 4-5e-4 px
    close(acc['smooth_mesh2'], g['smooth_mesh2'], 5e-3, 'smooth_mesh2')
    close(acc['ori_path2'], g['ori_path2'], 1e-2, 'ori_path2')            # a 15-step cumulative sum: observed 1.1e-3
    close(acc['smooth_path2'], g['smooth_path2'], 1e-2, 'smooth_path2')
    m1 = torch.from_numpy(g['smooth_mesh1']).to(dev)
    m2 = torch.from_numpy(g['smooth_mesh2']).to(dev)
    for wm, fm in (('NORMAL', 'AVERAGE'), ('FAST', 'AVERAGE'), ('NORMAL', 'LINEAR')):
        tag = '%s_%s' % (wm.lower(), fm.lower())
        frames, ow, oh = pipeline.get_stable_sqe(hr[0], hr[1], m1, m2, wm, fm)
        assert [int(oh), int(ow)] == list(g['canvas_' + tag])
        got = np.stack([cases.box_down(f, 16) for f in frames])
        close_boxes(got, g['frames_' + tag], g['iqr_' + tag], 5e-2, 'frames ' + tag)
        if tag == 'normal_average':
            close(frames[0][150:214, 300:396], g['frame0_crop'], 5e-2, 'frame0 crop')
    # metric harness on device: LR warps with masks, fp64 PSNR / SSIM, stability, distortion vs the reference values
    from stabstitch2_amd import metrics
    w1 = metrics.warp_lr_planes(torch.cat(lr[0], 0).to(dev), m1)
    w2 = metrics.warp_lr_planes(torch.cat(lr[1], 0).to(dev), m2)
    ps, ss = metrics.alignment_psnr_ssim(w1, w2)
    assert float((ps.cpu() - torch.from_numpy(g['psnr'])).abs().max()) < 0.01, (ps.cpu(), g['psnr'])
    assert float((ss.cpu() - torch.from_numpy(g['ssim'])).abs().max()) < 1e-3, (ss.cpu(), g['ssim'])
    six = metrics.warp_lr_with_mask(torch.cat(lr[0], 0).to(dev), m1)
    p0, s0 = M.alignment_psnr_ssim(six[0].cpu().numpy(), metrics.warp_lr_with_mask(torch.cat(lr[1], 0).to(dev), m2)[0].cpu().numpy())
    assert abs(p0 - float(ps[0])) < 1e-6 and abs(s0 - float(ss[0])) < 1e-9      # device fp64 == oracle fp64 arithmetic
    assert abs(metrics.stability_score(torch.from_numpy(g['smooth_path2']).to(dev)) - float(g['stability'])) < 1e-4
    assert abs(metrics.distortion_score(m2) - float(g['distortion'])) < 1e-5
    ev = metrics.evaluate_clip(hip_nets, lr[0], lr[1])
    assert float((ev['psnr'].cpu() - torch.from_numpy(g['psnr'])).abs().max()) < 0.01
    assert float((ev['ssim'].cpu() - torch.from_numpy(g['ssim'])).abs().max()) < 1e-3
    assert abs(ev['stability'] - float(g['stability'])) < 1e-3 and abs(ev['distortion'] - float(g['distortion'])) < 1e-4
    # end to end with the pipeline's own meshes
    frames, hc, wc, sm1, sm2 = pipeline.run_two_view(hr[0], hr[1], lr[0], lr[1], hip_nets)
    assert [hc, wc] == list(g['canvas_normal_average'])
    got = np.stack([cases.box_down(f.permute(1, 2, 0).cpu().numpy(), 16) for f in frames])
    close_boxes(got, g['frames_normal_average'], g['iqr_normal_average'], 5e-2, 'end-to-end frames')


def test_three_view_vs_reference(dev, golden):
    from stabstitch2_amd import pipeline
    g = golden('g10_threeview')
    meshes = [m.to(dev) for m in cases.g10_meshes()]
    n = meshes[0].shape[1]
    hr, _ = synth.make_clip(n, 180, 320, seed=3, views=3)
    mesh1, mid, mesh3 = pipeline.three_view_compose(*meshes, 180, 320)
    close(mesh1, g['mesh1'], 5e-3, 'mesh1')                   # observed 3.4e-4 px
    close(mid, g['middle'], 1e-3, 'middle')
    close(mesh3, g['mesh3'], 5e-3, 'mesh3')
    gm = [torch.from_numpy(g[k]).to(dev) for k in ('mesh1', 'middle', 'mesh3')]
    for fm in ('AVERAGE', 'LINEAR'):
        frames, hc, wc = pipeline.three_view_render(hr[0], hr[1], hr[2], *gm, 'NORMAL', fm)
        assert [hc, wc] == list(g['canvas_' + fm.lower()])
        got = np.stack([cases.box_down(f.permute(1, 2, 0).cpu().numpy(), 4) for f in frames])
        # see tests/test_oracle_golden.py::test_g10_three_view for why AVERAGE is only loosely comparable
        # LINEAR: nonzero() centroids count the +-1e-3 out-of-range residues of the masks, so the blend weights move
        # by ~1e-3 between CPUs already (0.12 grey levels oracle-vs-golden across two x86 hosts)
        tol, cover = (3.0, 0.3) if fm == 'AVERAGE' else (0.1, 0.6)      # LINEAR observed 1.8e-2 (round 4: 0.5 -> 0.1)
        close_boxes(got, g['frames_' + fm.lower()], g['iqr_' + fm.lower()], tol, 'three-view ' + fm, k=4, cover=cover)
    # the fused 3-view kernel must equal the chained formula applied to the generic per-view warp, bit for bit
    from stabstitch2_amd import ops
    hc, wc, src, T = pipeline.render_plan(gm, 180, 320, prescaled=True)
    imgs = [hr[k][1].to(dev) for k in range(3)]
    w = ops.tps_warp(torch.cat(imgs, 0), src[1], T[1], hc, wc, 'NORMAL')
    f12 = w[0] * (w[0] / (w[0] + w[1] + 1e-6)) + w[1] * (w[1] / (w[0] + w[1] + 1e-6))
    f = f12 * (f12 / (f12 + w[2] + 1e-6)) + w[2] * (w[2] / (f12 + w[2] + 1e-6))
    fused = ops.render_average(imgs, src[1], T[1], hc, wc, 'NORMAL')
    assert torch.equal(fused, f), float((fused - f).abs().max())


def test_missing_device_fails_loudly():
    from stabstitch2_amd import ops, _hip
    with pytest.raises(_hip.HipError):
        ops.maxpool(torch.zeros(1, 4, 4, 4), 2, 2)


# ------------------------------------------------------------------ size-independent properties at full benchmark sizes
def test_full_size_properties_720p(dev):
    """720x1280 (BASELINE configs[2]) is too large for fixtures; check properties that hold at any size."""
    from stabstitch2_amd import ops, pipeline
    from stabstitch2_amd.spatial_network import get_rigid_mesh, get_norm_mesh
    h, w = 720, 1280
    hr, _ = synth.make_clip_device(2, h, w, seed=5, device=dev)
    img = hr[0, 0:1]                                                   # [1,3,720,1280]
    nrigid = get_norm_mesh(get_rigid_mesh(1, h, w, device=dev), h, w).contiguous()
    T = ops.tps_solve(nrigid, nrigid)
    # identity spline: affine part = identity, RBF weights ~ 0
    assert float((T[0, 0, :3] - torch.tensor([0., 1., 0.], device=dev)).abs().max()) < 1e-5
    assert float(T[0, :, 3:].abs().max()) < 1e-5
    # FAST + identity mesh = the image itself (align_corners=True); NORMAL = zoom by W/(W-1), last row/col exactly 0
    fast = ops.tps_warp(img, nrigid, T, h, w, 'FAST')
    assert float((fast - img).abs().max()) < 2e-2
    normal = ops.tps_warp(img, nrigid, T, h, w, 'NORMAL')
    assert float(normal[..., -1, :].abs().max()) < 1e-3 and float(normal[..., :, -1].abs().max()) < 1e-3
    # linearity of the sampler in the image: warp(a U1 + b U2) = a warp(U1) + b warp(U2)
    rs = np.random.RandomState(0)
    mesh = get_rigid_mesh(1, h, w, device=dev) + torch.from_numpy(rs.normal(0, 4, (1, 7, 9, 2)).astype(np.float32)).to(dev)
    src = get_norm_mesh(mesh, h, w).contiguous()
    T2 = ops.tps_solve(src, nrigid)
    u1, u2 = hr[0, 0:1], hr[1, 1:2]
    lhs = ops.tps_warp(0.25 * u1 + 0.5 * u2, src, T2, h, w, 'FAST')
    rhs = 0.25 * ops.tps_warp(u1, src, T2, h, w, 'FAST') + 0.5 * ops.tps_warp(u2, src, T2, h, w, 'FAST')
    assert float((lhs - rhs).abs().max()) < 1e-3
    # fused render == formula on the generic per-view warps, bit for bit, at the full canvas
    srcs = torch.cat((src, nrigid), 0)
    Ts = torch.cat((T2, T), 0)
    wv = ops.tps_warp(torch.cat((u1, u2), 0), srcs, Ts, h + 20, w + 500, 'NORMAL')
    f = wv[0] * (wv[0] / (wv[0] + wv[1] + 1e-6)) + wv[1] * (wv[1] / (wv[0] + wv[1] + 1e-6))
    assert torch.equal(ops.render_average([u1, u2], srcs, Ts, h + 20, w + 500, 'NORMAL'), f)
    # whole pipeline on a 720p clip: canvas bigger than the frame, every output finite, meshes reproducible
    from bench import build_nets
    nets, _ = build_nets(dev)
    hr8, lr8 = synth.make_clip_device(8, h, w, seed=2, device=dev)
    fr, hc, wc, m1, m2 = pipeline.run_two_view(hr8[0], hr8[1], lr8[0], lr8[1], nets)
    assert hc >= h and wc > w and fr.shape == (8, 3, hc, wc) and bool(torch.isfinite(fr).all())
    fr2, hc2, wc2, m1b, _ = pipeline.run_two_view(hr8[0], hr8[1], lr8[0], lr8[1], nets)
    assert (hc2, wc2) == (hc, wc) and torch.equal(m1, m1b) and torch.equal(fr, fr2)       # deterministic (fixed-order split-K)


def test_conv_properties_and_edges(dev):
    from stabstitch2_amd import ops
    torch.manual_seed(1)
    x1 = torch.randn(3, 11, 15, 128, device=dev)
    x2 = torch.randn(3, 11, 15, 128, device=dev)
    w = torch.randn(128, 1, 3, 3, 128, device=dev) * 0.03
    lin = ops.conv(2.0 * x1 - 3.0 * x2, w, None)
    ref = 2.0 * ops.conv(x1, w, None) - 3.0 * ops.conv(x2, w, None)
    assert float((lin - ref).abs().max()) < 2e-4                       # linearity (split-K path: M = 495)
    one = ops.conv(x1[:1, :1, :1].contiguous(), w, None)                # 1x1 image: only the centre tap is in bounds
    assert one.shape == (1, 1, 1, 128)
    exp = (x1[0, 0, 0] * w[:, 0, 1, 1, :]).sum(1)
    assert float((one.view(-1) - exp).abs().max()) < 1e-4
    with pytest.raises(Exception):
        ops.conv(torch.randn(1, 4, 4, 6, device=dev), torch.randn(8, 1, 3, 3, 6, device=dev))   # cin % 4 != 0
    with pytest.raises(Exception):
        ops.conv(torch.randn(1, 2, 2, 4, device=dev), torch.randn(8, 1, 5, 5, 4, device=dev), pad=(0, 0, 0))  # empty output


def test_short_clip_is_rejected(dev, hip_nets):
    from stabstitch2_amd import pipeline
    _, lr = synth.make_clip(6, 360, 480, seed=0)
    with pytest.raises(ValueError):
        pipeline.estimate_meshes(hip_nets, lr[0], lr[1])


def test_psnr_ssim_vs_skimage(dev, golden):
    """G11: scikit-image 0.18.3 values on two seeded images (mask plane = 1)."""
    from stabstitch2_amd import metrics
    g = golden('g11_metrics')
    a, b = cases.g11_images()
    def planes(x):
        t = torch.from_numpy(x).permute(2, 0, 1)
        return torch.cat((t, torch.ones(1, *t.shape[1:])), 0).unsqueeze(0).contiguous().to(dev)
    p, s = metrics.alignment_psnr_ssim(planes(a), planes(b))
    assert abs(float(p[0]) - float(g['psnr'])) < 1e-6 and abs(float(s[0]) - float(g['ssim'])) < 1e-6


def test_online_matches_offline(dev, hip_nets, clip16):
    """Streaming mode (ring buffer, cached temporal features, fixed canvas) reproduces the offline clip when it is
    given the offline canvas; with its own canvas it still yields one finite frame per pushed pair."""
    from stabstitch2_amd import pipeline, ops
    from stabstitch2_amd.online import OnlineStitcher
    hr, lr = clip16
    hrd = [[f.to(dev) for f in v] for v in hr]
    lrd = [[f.to(dev) for f in v] for v in lr]
    n = 12
    acc = pipeline.estimate_meshes(hip_nets, lrd[0][:n], lrd[1][:n])
    off, hc, wc = pipeline.render_frames([hrd[0][:n], hrd[1][:n]], [acc['smooth_mesh1'], acc['smooth_mesh2']])
    bbox = ops.mesh_bbox([acc['smooth_mesh1'], acc['smooth_mesh2']], 360, 480).cpu().tolist()
    st = OnlineStitcher(hip_nets, 360, 480, canvas=bbox)
    frames = []
    counts = []
    for t in range(n):
        got = st.push(hrd[0][t], hrd[1][t], lrd[0][t], lrd[1][t])
        counts.append(len(got))
        frames += got
    assert counts == [0] * 6 + [7] + [1] * (n - 7)
    assert (st.hc, st.wc) == (hc, wc)
    on = torch.stack(frames, 0)
    d = (on - off).abs()
    # same arithmetic per frame, batch-1 vs batched launches differ only in split-K summation order
    assert float(d.median()) < 1e-3 and float(torch.quantile(d.flatten()[::17], 0.999)) < 0.05, (float(d.median()), float(d.max()))
    st2 = OnlineStitcher(hip_nets, 360, 480)
    outs = []
    for t in range(9):
        outs += st2.push(hrd[0][t], hrd[1][t], lrd[0][t], lrd[1][t])
    assert len(outs) == 9 and all(bool(torch.isfinite(f).all()) for f in outs) and st2.hc >= hc and st2.wc >= wc


# ------------------------------------------------------------------ frame I/O
@pytest.mark.parametrize('h,w,lr_h,lr_w', [
    (720, 1280, 360, 480),      # the benchmark size: general fixed-point linear path
    (720, 960, 360, 480),       # exact 2x2 -> OpenCV routes INTER_LINEAR to the fast area average
    (360, 480, 360, 480),       # StabStitch-D native size: copy
    (101, 203, 360, 480),       # upscale, width not a multiple of 4 (scalar HR path, clamped taps both ends)
    (1080, 1920, 360, 480),     # 3x / 4x decimation
    (37, 52, 20, 31),
])
def test_ingest_u8_bit_exact(dev, h, w, lr_h, lr_w):
    """ss_ingest_u8 == the numpy restatement of cv2.imread->float / cv2.resize->/127.5-1 (test_online_tra.py:252-264),
    bit for bit, including extreme-valued and constant frames."""
    from oracle import frame_io as FIO
    from stabstitch2_amd import ops
    rng = np.random.RandomState(h * 7 + w)
    frames = rng.randint(0, 256, (3, h, w, 3)).astype(np.uint8)
    frames[1] = 255
    frames[2, ::2] = 0
    hr, lr = ops.ingest_u8(torch.from_numpy(frames).to(dev), lr_h, lr_w)
    for i in range(3):
        rh, rl = FIO.load_frame(frames[i], lr_h, lr_w)
        assert np.array_equal(hr[i].cpu().numpy(), rh), 'hr planes differ'
        got = lr[i].cpu().numpy()
        assert np.array_equal(got, rl), ('lr differs', float(np.abs(got - rl).max()) * 127.5)
    _, lr_only = ops.ingest_u8(torch.from_numpy(frames).to(dev), lr_h, lr_w, want_hr=False)
    assert torch.equal(lr_only, lr)


def test_canvas_to_u8_bit_exact(dev):
    """ss_canvas_to_u8 == `.astype(np.uint8)` of the HWC canvas (test_online_tra.py:413), aligned and ragged sizes."""
    from oracle import frame_io as FIO
    from stabstitch2_amd import ops
    rng = np.random.RandomState(3)
    for (h, w) in ((64, 128), (37, 53), (1, 1), (730, 1414)):
        x = (rng.rand(2, 3, h, w) * 256).astype(np.float32)
        x[0, :, 0, 0] = [255.99998, 0.0, 254.99998]
        if h > 1:
            x[1, :, -1, -1] = [-0.5, 256.0, -3.7]          # outside 0..255: low byte of the truncated int32
        got = ops.canvas_to_u8(torch.from_numpy(x).to(dev)).cpu().numpy()
        for i in range(2):
            assert np.array_equal(got[i], FIO.to_video_frame(x[i])), (h, w)


def test_u8_pipeline_matches_float_pipeline(dev, hip_nets):
    """uint8 in -> uint8 out equals the fp32 path fed with the oracle-loaded frames, truncated: the front-end and
    the sink add no arithmetic of their own."""
    from oracle import frame_io as FIO
    from stabstitch2_amd import pipeline
    clip = synth.make_clip(9, 540, 720, seed=5)
    u8 = [np.stack([np.clip(np.rint(f[0].numpy().transpose(1, 2, 0)), 0, 255).astype(np.uint8) for f in v])
          for v in clip[0]]
    out, hc, wc, m1, m2 = pipeline.run_two_view_u8(u8[0], u8[1], hip_nets, device=dev)
    loaded = [[FIO.load_frame(f) for f in v] for v in u8]
    hr = [torch.from_numpy(np.stack([a for a, _ in v])).to(dev) for v in loaded]
    lr = [torch.from_numpy(np.stack([b for _, b in v])).to(dev) for v in loaded]
    ref, hc2, wc2, r1, r2 = pipeline.run_two_view(hr[0], hr[1], lr[0], lr[1], hip_nets)
    assert (hc, wc) == (hc2, wc2) and torch.equal(m1, r1) and torch.equal(m2, r2)
    want = np.stack([FIO.to_video_frame(f) for f in ref.cpu().numpy()])
    assert out.dtype == torch.uint8 and tuple(out.shape) == (9, hc, wc, 3)
    assert np.array_equal(out.cpu().numpy(), want)
    # the render that samples the uint8 frames and writes the uint8 video frame itself (default) against the three-step
    # route ingest -> fp32 render -> ss_canvas_to_u8: equal bytes, both warp modes, with and without footprints
    assert pipeline.U8_FUSED
    for wm in ('NORMAL', 'FAST'):
        for skip in (True, False):
            old_skip, pipeline.SKIP_OUTSIDE = pipeline.SKIP_OUTSIDE, skip
            try:
                fused = pipeline.run_two_view_u8(u8[0], u8[1], hip_nets, warp_mode=wm, device=dev)[0]
                pipeline.U8_FUSED = False
                steps = pipeline.run_two_view_u8(u8[0], u8[1], hip_nets, warp_mode=wm, device=dev)[0]
            finally:
                pipeline.U8_FUSED = True
                pipeline.SKIP_OUTSIDE = old_skip
            assert torch.equal(fused, steps), (wm, skip)
    # three views through the same kernel
    from stabstitch2_amd import ops
    f3 = [torch.from_numpy(np.ascontiguousarray(v)).to(dev) for v in u8] + [torch.from_numpy(np.ascontiguousarray(u8[0][:, :, ::-1])).to(dev)]
    src = torch.stack([pipeline.get_norm_mesh(pipeline.get_rigid_mesh(1, 540, 720, device=dev) + 3.0 * k, 560, 760)[0] for k in range(3)])
    tgt = pipeline.get_norm_mesh(pipeline.get_rigid_mesh(1, 540, 720, device=dev), 540, 720).expand(3, -1, -1).contiguous()
    T = ops.tps_solve(src.contiguous(), tgt)
    planes = [ops.ingest_u8(f[:1])[0][0] for f in f3]
    want3 = ops.canvas_to_u8(ops.render_average(planes, src, T, 560, 760).unsqueeze(0))[0]
    got3 = ops.render_average_u8([f[0] for f in f3], src, T, 560, 760)
    assert torch.equal(got3, want3)


def test_host_clip_runner_overlapped_copies(dev, hip_nets):
    """The three-stream host pipeline (upload / compute / download overlapped) returns, clip for clip, exactly what
    the synchronous uint8 path returns."""
    from stabstitch2_amd import pipeline
    clips = []
    for seed in (1, 2, 3):
        c = synth.make_clip(8, 360, 480, seed=seed)
        clips.append(tuple(np.stack([np.clip(np.rint(f[0].numpy().transpose(1, 2, 0)), 0, 255).astype(np.uint8)
                                     for f in v]) for v in c[0]))
    want = [pipeline.run_two_view_u8(a, b, hip_nets, device=dev, to_host=True) for a, b in clips]
    runner = pipeline.HostClipRunner(hip_nets, dev)
    got = []
    for video, hc, wc in runner.run((torch.from_numpy(a).pin_memory(), torch.from_numpy(b).pin_memory())
                                    for a, b in clips):
        got.append((video.numpy().copy(), hc, wc))
    assert len(got) == 3
    for (v, hc, wc), (w, hc2, wc2, _, _) in zip(got, want):
        assert (hc, wc) == (hc2, wc2) and np.array_equal(v, w)
    assert list(runner.run(iter(()))) == []


def test_conv_random_shapes_and_address_modes(dev, request):
    """Seeded sweep over ragged conv geometries (every padding / stride / kernel size class the tap table and the
    tap-validity masks distinguish: 1..49 taps -> 32-bit masks, 64-bit masks, > 64 taps -> arithmetic path) against
    F.conv2d / F.conv3d on the CPU; the LDS-table and the arithmetic address paths must agree bit for bit."""
    from stabstitch2_amd import ops, _hip
    # the address-mode switch is a tuning knob: it exists only in the tools/ build of the library (-DSS_TUNING), which
    # this test loads for its own launches; the comparison against the CPU convolution also runs on the product library
    import importlib.util
    spec = importlib.util.spec_from_file_location('_tuning', os.path.join(os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), 'tools', '_tuning.py'))
    tuning = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tuning)
    product = _hip.lib()
    request.addfinalizer(lambda: setattr(_hip, '_lib', product))
    lib = tuning.lib()
    rs = np.random.RandomState(2024)
    cases = []
    for _ in range(36):
        k = int(rs.choice([1, 2, 3, 5, 7, 8]))
        cases.append((int(rs.randint(1, 4)), int(rs.choice([1, 3, 4, 8, 20, 64, 68])), int(rs.randint(1, 140)),
                      int(rs.randint(1, 24)), int(rs.randint(1, 24)), k, int(rs.choice([1, 2, 3])), int(rs.randint(0, k // 2 + 2))))
    cases += [(1, 4, 64, 12, 12, 8, 1, 4), (2, 8, 3, 9, 31, 8, 2, 0)]          # 64 taps: last bit of the 64-bit mask
    for (n, cin, cout, h, w, k, s, p) in cases:
        if (h + 2 * p - k) // s + 1 <= 0 or (w + 2 * p - k) // s + 1 <= 0:
            continue
        x = torch.from_numpy(rs.normal(0, 1, (n, cin, h, w)).astype(np.float32))
        wt = torch.from_numpy((rs.normal(0, 1, (cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32))
        b = torch.from_numpy(rs.normal(0, 1, cout).astype(np.float32))
        ref = F.conv2d(x, wt, b, stride=s, padding=p)
        cp = (cin + 3) // 4 * 4
        xd, wd, bd = ops.nchw_to_nhwc(x.to(dev), cp), _pack(wt, cp).to(dev), b.to(dev)
        outs = []
        for mode in (0, 1):
            lib.ss_debug_set(3, mode)
            outs.append(ops.conv(xd, wd, bd, None, stride=s, pad=(0, p, p), relu=False))
        lib.ss_debug_set(3, 0)
        assert torch.equal(outs[0], outs[1]), ('table vs arithmetic addressing differ', (n, cin, cout, h, w, k, s, p))
        _hip._lib = product
        prod = ops.conv(xd, wd, bd, None, stride=s, pad=(0, p, p), relu=False)
        _hip._lib = lib
        assert torch.equal(prod, outs[0]), 'product and tuning builds differ'
        close(ops.nhwc_to_nchw(outs[0]), ref, 3e-5 * max(1.0, float(ref.abs().max())), 'conv %s' % ((n, cin, cout, h, w, k, s, p),))
    # 3-D: 5x3x3 = 45 taps (64-bit masks) and 5x5x5 = 125 taps (arithmetic fallback), temporal padding included
    for (kt, kk, pt, pp) in ((5, 3, 2, 1), (3, 3, 0, 1), (5, 5, 2, 2)):
        x = torch.from_numpy(rs.normal(0, 1, (2, 8, 6, 7, 9)).astype(np.float32))
        wt = torch.from_numpy((rs.normal(0, 1, (10, 8, kt, kk, kk)) / np.sqrt(8 * kt * kk * kk)).astype(np.float32))
        b = torch.from_numpy(rs.normal(0, 1, 10).astype(np.float32))
        ref = F.conv3d(x, wt, b, padding=(pt, pp, pp))
        out = ops.conv(x.permute(0, 2, 3, 4, 1).contiguous().to(dev), _pack(wt, 8).to(dev), b.to(dev), None, stride=1,
                       pad=(pt, pp, pp), relu=False)
        close(out.permute(0, 4, 1, 2, 3), ref, 3e-5 * max(1.0, float(ref.abs().max())), 'conv3d %s' % ((kt, kk, pt, pp),))


def test_online_graph_replay_equals_eager(dev, hip_nets, clip16):
    """The captured steady state (HIP graph over static ring buffers) produces exactly the frames of the eager
    steady state, push after push, and both continue the warm-up (list-based) state without a seam."""
    from stabstitch2_amd.online import OnlineStitcher
    hr, lr = clip16
    hrd = [[f.to(dev) for f in v] for v in hr]
    lrd = [[f.to(dev) for f in v] for v in lr]
    outs = {}
    for use_graph in (False, True):
        st = OnlineStitcher(hip_nets, 360, 480, use_graph=use_graph)
        frames = []
        for t in range(14):
            frames += st.push(hrd[0][t], hrd[1][t], lrd[0][t], lrd[1][t])
        assert len(frames) == 14 and (st.graph is not None) == use_graph
        outs[use_graph] = torch.stack(frames, 0)
    assert torch.equal(outs[False], outs[True])


def test_degenerate_inputs_stay_finite_and_match_oracle(dev, hip_nets):
    """Constant frames (all black / all white: zero-variance features, degenerate correlation) go through the whole
    path without NaN/Inf and agree with the CPU oracle like ordinary frames do."""
    from stabstitch2_amd import pipeline
    onets = (N.SpatialNet().eval(), N.TemporalNet().eval(), N.SmoothNet().eval())
    for m in onets:
        m.load_state_dict(synth.synthetic_state_dict(m), strict=True)
    n = 8
    for val in (0.0, 255.0):
        hr = torch.full((n, 3, 360, 480), val)
        lr = hr / 127.5 - 1.0
        frames, hc, wc, m1, m2 = pipeline.run_two_view(hr.to(dev), hr.to(dev), lr.to(dev), lr.to(dev), hip_nets)
        assert bool(torch.isfinite(frames).all()) and bool(torch.isfinite(m1).all()) and bool(torch.isfinite(m2).all())
        ref = P.run_two_view([hr[i:i + 1] for i in range(n)], [hr[i:i + 1] for i in range(n)],
                             [lr[i:i + 1] for i in range(n)], [lr[i:i + 1] for i in range(n)], onets)
        assert (hc, wc) == (ref[1], ref[2])
        close(m1.cpu(), ref[3], 5e-3, 'degenerate smooth_mesh1 (val %g)' % val)
        close(m2.cpu(), ref[4], 5e-3, 'degenerate smooth_mesh2 (val %g)' % val)


# ------------------------------------------------------------------ round 2: three-view full path, 720p oracle parity, host API
def _oracle_nets():
    onets = (N.SpatialNet().eval(), N.TemporalNet().eval(), N.SmoothNet().eval())
    for m in onets:
        m.load_state_dict(synth.synthetic_state_dict(m), strict=True)
    return onets


def test_cost_volume_norm_true(dev, golden):
    """cost_volume(norm=True) -- the reference signature's default (spatial_network.py:335-337), unused by inference."""
    from stabstitch2_amd.spatial_network import SpatialNet
    from stabstitch2_amd.temporal_network import TemporalNet
    g = golden('g3_costvol')
    a, b = cases.g3_inputs(False)
    close(SpatialNet.cost_volume(a.to(dev), b.to(dev), 5, norm=True), g['cv5n'], 1e-6, 'cv5 norm=True')
    close(TemporalNet.cost_volume(a.to(dev), b.to(dev), 3, norm=True), g['cv3n'], 1e-6, 'cv3 norm=True')
    close(SpatialNet.cost_volume(a.to(dev), b.to(dev), 5, norm=False), g['cv5'], 1e-5, 'cv5 norm=False')


def _three_view_unshared(nets, lr):
    """Two independent 2-view passes with none of run_three_view's sharing (no reused TemporalNet motions, no reused
    SpatialNet trunk features, no shared stem)."""
    from stabstitch2_amd import pipeline
    outs = []
    for a, b in ((lr[0], lr[1]), (lr[1], lr[2])):
        s1, s2 = pipeline.spatial_stage(nets[0], a, b)
        t1, t2 = pipeline.temporal_stage(nets[1], a), pipeline.temporal_stage(nets[1], b)
        outs.append((s1, s2, t1, t2))
    return outs


def test_run_three_view_sharing_is_exact(dev, hip_nets):
    """configs[4]: run_three_view computes the middle view's TemporalNet motions and SpatialNet trunk features once and
    the (1,2) pair's stems jointly; every shared quantity must equal the unshared computation up to split-K /
    batch-composition summation order (<= 1e-4 px), the canvas exactly."""
    from stabstitch2_amd import pipeline
    n = 9
    hr, lr = synth.make_clip_device(n, 180, 320, seed=6, views=3, device=dev)
    a12 = pipeline.estimate_meshes(hip_nets, lr[0], lr[1], keep_spatial_cache2=True)
    a23 = pipeline.estimate_meshes(hip_nets, lr[1], lr[2], tmotion1=a12['tmotion2'],
                                   spatial_cache1=a12.get('spatial_cache2'))
    u12, u23 = _three_view_unshared(hip_nets, lr)
    for name, got, ref in (('s1_12', a12['smotion1'], u12[0]), ('s2_12', a12['smotion2'], u12[1]),
                           ('t1_12', a12['tmotion1'], u12[2]), ('t2_12', a12['tmotion2'], u12[3]),
                           ('s1_23', a23['smotion1'], u23[0]), ('s2_23', a23['smotion2'], u23[1]),
                           ('t1_23', a23['tmotion1'], u23[2]), ('t2_23', a23['tmotion2'], u23[3])):
        close(got, ref, 1e-4, 'shared vs unshared ' + name)
    # whole path: shared run_three_view vs the same clip through plain (unshared, separate-stem) 2-view passes
    import stabstitch2_amd.pipeline as PL
    fr, hc, wc, m1, mid, m3 = pipeline.run_three_view(hr[0], hr[1], hr[2], lr[0], lr[1], lr[2], hip_nets)
    old = PL.SHARED_STEM
    PL.SHARED_STEM = False
    try:
        b12 = pipeline.estimate_meshes(hip_nets, lr[0], lr[1])
        b23 = pipeline.estimate_meshes(hip_nets, lr[1], lr[2])
    finally:
        PL.SHARED_STEM = old
    r1, rmid, r3 = pipeline.three_view_compose(b12['smooth_mesh1'], b12['smooth_mesh2'], b23['smooth_mesh1'],
                                               b23['smooth_mesh2'], 180, 320)
    close(m1, r1, 1e-3, 'three-view mesh1 shared vs unshared')
    close(mid, rmid, 1e-3, 'three-view middle shared vs unshared')
    close(m3, r3, 1e-3, 'three-view mesh3 shared vs unshared')
    fr2, hc2, wc2 = pipeline.three_view_render(hr[0], hr[1], hr[2], r1, rmid, r3)
    assert (hc, wc) == (hc2, wc2) and fr.shape == (n, 3, hc, wc) and bool(torch.isfinite(fr).all())


@pytest.mark.parametrize('h,w,n', [(180, 320, 8), (720, 1280, 7)])
def test_run_three_view_vs_oracle(dev, hip_nets, h, w, n):
    """configs[4] against oracle/pipeline.py run_three_view (two full 2-view passes + composition + 3-image render) on
    the same clip: re-projected meshes, canvas, box-median frames."""
    from stabstitch2_amd import pipeline
    hr, lr = synth.make_clip(n, h, w, seed=7, views=3)
    hrd = [torch.cat(v, 0).to(dev) for v in hr]
    lrd = [torch.cat(v, 0).to(dev) for v in lr]
    fr, hc, wc, m1, mid, m3 = pipeline.run_three_view(hrd[0], hrd[1], hrd[2], lrd[0], lrd[1], lrd[2], hip_nets)
    ofr, ohc, owc, om1, omid, om3 = P.run_three_view(hr[0], hr[1], hr[2], lr[0], lr[1], lr[2], _oracle_nets())
    scale = h / 360.0                      # meshes are HR canvas pixels: LR-px error times the HR scale
    close(m1, om1, 5e-3 * scale, 'mesh1 vs oracle')
    close(mid, omid, 5e-3 * scale, 'middle vs oracle')
    close(m3, om3, 5e-3 * scale, 'mesh3 vs oracle')
    assert (hc, wc) == (ohc, owc)
    k = 4 if h < 360 else 16
    got = np.stack([cases.box_down(f.permute(1, 2, 0).cpu().numpy(), k) for f in fr])
    ref = np.stack([cases.box_down(f.numpy().transpose(1, 2, 0), k) for f in ofr])
    rng = np.stack([cases.box_iqr(f.numpy().transpose(1, 2, 0), k) for f in ofr])
    # chained AVERAGE is chaotic where only view 3 is valid (avg(residue, residue) hits its 1e-6
    # denominator, the reference's own output there differs between machines by O(10) grey levels): compare the boxes
    # left of the middle view's right border sharply, the rest only through the median deviation
    xlim = int(float(omid[..., 0].max()) // k) - 1
    ok = cases.smooth_boxes(rng, k)
    ok[:, :, xlim:] = False
    assert ok.mean() > 0.3, ok.mean()
    dd = np.abs(got - ref)[ok]
    if os.environ.get('SS_VERBOSE'):
        print('  three-view AVERAGE (views 1-2 region): p99 %.3e  p99.9 %.3e  max %.3e' % (
            np.quantile(dd, 0.99), np.quantile(dd, 0.999), dd.max()))
    # isolated boxes on the outer border (neither of views 1, 2 valid) stay chaotic: quantiles, with the old loose bound as max
    assert np.quantile(dd, 0.99) < 0.05 and dd.max() < 3.0, (float(np.quantile(dd, 0.99)), float(dd.max()))
    clean = cases.smooth_boxes(rng, k)
    assert np.median(np.abs(got - ref)[clean]) < 0.02, float(np.median(np.abs(got - ref)[clean]))
    # LINEAR fusion has no singular denominator: whole canvas
    frl = pipeline.run_three_view(hrd[0], hrd[1], hrd[2], lrd[0], lrd[1], lrd[2], hip_nets, 'NORMAL', 'LINEAR')[0]
    ofl = P.three_view_render(hr[0], hr[1], hr[2], om1, omid, om3, 'NORMAL', 'LINEAR')[0]
    gl = np.stack([cases.box_down(f.permute(1, 2, 0).cpu().numpy(), k) for f in frl])
    rl = np.stack([cases.box_down(f.numpy().transpose(1, 2, 0), k) for f in ofl])
    il = np.stack([cases.box_iqr(f.numpy().transpose(1, 2, 0), k) for f in ofl])
    # (round 4: the blender's masks / centroids / mask1 are pinned on the reference's own values,
    # test_three_view_linear_blender_internals_vs_reference: mask1 moves by <= 4.4e-3 -> 0.5 tightened to 0.3, observed 0.19)
    close_boxes(gl, rl, il, 0.3, 'three-view LINEAR frames vs oracle', k=k, cover=0.5)


def test_three_view_full_path_vs_reference(dev, golden, hip_nets):
    """G12: test_online_tra_threeview.py:154-505 end to end, produced by the reference itself (two 2-view passes through
    its networks, then its composition / render block verbatim)."""
    from stabstitch2_amd import pipeline
    g = golden('g12_threeview_full')
    n = g['mesh1'].shape[1]
    hr, lr = synth.make_clip(n, 180, 320, seed=4, views=3)
    hrd = [torch.cat(v, 0).to(dev) for v in hr]
    lrd = [torch.cat(v, 0).to(dev) for v in lr]
    a12 = pipeline.estimate_meshes(hip_nets, lrd[0], lrd[1], keep_spatial_cache2=True)
    a23 = pipeline.estimate_meshes(hip_nets, lrd[1], lrd[2], tmotion1=a12['tmotion2'],
                                   spatial_cache1=a12.get('spatial_cache2'))
    close(a12['smooth_mesh1'], g['w12_m1'], 5e-3, 'pass (1,2) smooth_mesh1 vs reference')
    close(a12['smooth_mesh2'], g['w12_m2'], 5e-3, 'pass (1,2) smooth_mesh2 vs reference')
    close(a23['smooth_mesh1'], g['w23_m1'], 5e-3, 'pass (2,3) smooth_mesh1 vs reference')
    close(a23['smooth_mesh2'], g['w23_m2'], 5e-3, 'pass (2,3) smooth_mesh2 vs reference')
    for fm in ('AVERAGE', 'LINEAR'):
        fr, hc, wc, m1, mid, m3 = pipeline.run_three_view(hrd[0], hrd[1], hrd[2], lrd[0], lrd[1], lrd[2], hip_nets,
                                                          'NORMAL', fm)
        close(m1, g['mesh1'], 5e-3, 'mesh1 vs reference')
        close(mid, g['middle'], 5e-3, 'middle vs reference')
        close(m3, g['mesh3'], 5e-3, 'mesh3 vs reference')
        assert [hc, wc] == list(g['canvas_' + fm.lower()])
        got = np.stack([cases.box_down(f.permute(1, 2, 0).cpu().numpy(), 4) for f in fr])
        tol, cover = (3.0, 0.3) if fm == 'AVERAGE' else (0.1, 0.6)      # LINEAR observed 1.8e-2 (round 4: 0.5 -> 0.1)
        close_boxes(got, g['frames_' + fm.lower()], g['iqr_' + fm.lower()], tol, 'G12 frames ' + fm, k=4, cover=cover)


def test_two_view_720p_vs_oracle(dev, hip_nets):
    """configs[2] as a real parity test (what bench.py checks at benchmark time): 8-frame 720x1280 clip, HIP path vs
    the CPU oracle -- meshes, canvas, frames (median / p99.9), alignment PSNR / SSIM within 0.01 dB / 1e-3."""
    from stabstitch2_amd import pipeline, metrics
    n = 8
    hr, lr = synth.make_clip_device(n, 720, 1280, seed=0, device='cpu')
    fr, hc, wc, m1, m2 = pipeline.run_two_view(hr[0].to(dev), hr[1].to(dev), lr[0].to(dev), lr[1].to(dev), hip_nets)
    sl = lambda t: [t[i:i + 1] for i in range(n)]
    ofr, ohc, owc, om1, om2 = P.run_two_view(sl(hr[0]), sl(hr[1]), sl(lr[0]), sl(lr[1]), _oracle_nets())
    close(m1, om1, 5e-3, '720p smooth_mesh1 vs oracle')
    close(m2, om2, 5e-3, '720p smooth_mesh2 vs oracle')
    assert (hc, wc) == (ohc, owc)
    for i in range(n):                 # every frame of the clip
        d = np.abs(fr[i].permute(1, 2, 0).cpu().numpy() - ofr[i])
        assert np.median(d) < 5e-3 and np.quantile(d, 0.999) < 0.1, (i, float(np.median(d)), float(np.quantile(d, 0.999)))
    k = 3
    c1 = M.warp_lr_with_mask(sl(lr[0])[:k], om1[:, :k])
    c2 = M.warp_lr_with_mask(sl(lr[1])[:k], om2[:, :k])
    cps = [M.alignment_psnr_ssim(a, b) for a, b in zip(c1, c2)]
    gp, gs = metrics.alignment_psnr_ssim(metrics.warp_lr_planes(lr[0][:k].to(dev), m1[:, :k]),
                                         metrics.warp_lr_planes(lr[1][:k].to(dev), m2[:, :k]))
    for i in range(k):
        assert abs(float(gp[i]) - cps[i][0]) < 0.01 and abs(float(gs[i]) - cps[i][1]) < 1e-3, (i, float(gp[i]), cps[i])


def test_load_nets_and_checkpoint_dir(dev, tmp_path, hip_nets):
    """pipeline.load_nets mirrors test_online_tra.py:173-194: exactly three *.pth, torch.load(p)['model'], strict."""
    from stabstitch2_amd import pipeline
    from stabstitch2_amd.spatial_network import SpatialNet
    from stabstitch2_amd.temporal_network import TemporalNet
    from stabstitch2_amd.smooth_network import SmoothNet
    d = str(tmp_path / 'Full_model_inference' / 'full_model_tra')
    synth.write_synthetic_checkpoints(d, SpatialNet(), TemporalNet(), SmoothNet())
    assert pipeline.find_model_dir(str(tmp_path)) == d
    nets = pipeline.load_nets(d, dev)
    _, lr = synth.make_clip_device(7, 360, 480, seed=1, device=dev)
    a = pipeline.estimate_meshes(nets, lr[0], lr[1])
    b = pipeline.estimate_meshes(hip_nets, lr[0], lr[1])
    assert torch.equal(a['smooth_mesh1'], b['smooth_mesh1']) and torch.equal(a['smooth_mesh2'], b['smooth_mesh2'])
    os.remove(os.path.join(d, 'smooth_warp.pth'))
    with pytest.raises(FileNotFoundError):
        pipeline.load_nets(d, dev)


def test_chunked_stages_equal_unchunked(dev, hip_nets):
    """Long clips: regressors / SmoothNet windows run in batch chunks (a conv launch addresses < 2 GiB); chunking must
    not change a bit of the per-item results, and a one-frame list behaves like the reference."""
    from stabstitch2_amd import layers as L, smooth_network as SN
    from stabstitch2_amd.temporal_network import build_TemporalNet
    sp, tp, sm = hip_nets
    x = torch.randn(7, 45, 60, 52, device=dev) * 0.1
    x[..., 49:] = 0
    full = L.run_regressor(x, tp._prepared()['r2'])
    part = L.run_regressor(x, tp._prepared()['r2'], chunk=3)
    close(part, full, 2e-5, 'chunked regressor')              # small batches take the split-K path: summation order
    cv = torch.randn(2, 5, 45, 60, 124, device=dev) * 0.1
    cv[..., 121:] = 0
    fa, fb = L.run_regressor_pair(cv, sp._prepared()['r2_pair'])
    pa, pb = L.run_regressor_pair(cv, sp._prepared()['r2_pair'], chunk=2)
    close(pa, fa, 2e-5, 'chunked twin regressor a')
    close(pb, fb, 2e-5, 'chunked twin regressor b')
    rs = np.random.RandomState(5)
    rigid = torch.from_numpy(cases.rigid(360, 480)).to(dev)
    mesh = [rigid + torch.from_numpy(rs.normal(0, 2, (20, 7, 9, 2)).astype(np.float32)).to(dev) for _ in range(2)]
    ts = [torch.from_numpy(rs.normal(0, 1, (20, 7, 9, 2)).astype(np.float32)).to(dev) for _ in range(2)]
    o_full, d_full = sm.run_windows(mesh[0], mesh[1], ts[0], ts[1], 14, 7, 1, 1)
    old = SN.WINDOW_CHUNK
    SN.WINDOW_CHUNK = 5
    try:
        o_part, d_part = sm.run_windows(mesh[0], mesh[1], ts[0], ts[1], 14, 7, 1, 1)
    finally:
        SN.WINDOW_CHUNK = old
    close(d_part, d_full, 2e-5, 'chunked SmoothNet windows')
    for k in o_full:
        close(o_part[k], o_full[k], 2e-4, 'chunked windows ' + k)
    one = build_TemporalNet(tp, [torch.zeros(1, 3, 360, 480, device=dev)])['motion_list']
    assert len(one) == 1 and tuple(one[0].shape) == (1, 7, 9, 2) and float(one[0].abs().max()) == 0.0


def test_reloaded_weights_invalidate_derived_caches(dev):
    """Shared-stem filters and the streaming twin trunk derive from two nets' weights; reloading one net must rebuild
    them (weights_version), not reuse the stale ones."""
    from stabstitch2_amd import pipeline
    from stabstitch2_amd.spatial_network import SpatialNet
    from stabstitch2_amd.temporal_network import TemporalNet
    from stabstitch2_amd.smooth_network import SmoothNet
    nets = []
    for cls in (SpatialNet, TemporalNet, SmoothNet):
        m = cls()
        m.load_state_dict(synth.synthetic_state_dict(m), strict=True)
        nets.append(m.to(dev))
    _, lr = synth.make_clip_device(7, 360, 480, seed=2, device=dev)
    a = pipeline.estimate_meshes(nets, lr[0], lr[1])
    sd = synth.synthetic_state_dict(nets[1])
    sd['feature_extractor_stage1.0.weight'] = sd['feature_extractor_stage1.0.weight'] * 1.5
    v0 = nets[1].weights_version
    nets[1].load_state_dict(sd, strict=True)
    assert nets[1].weights_version != v0
    b = pipeline.estimate_meshes(nets, lr[0], lr[1])              # shared stem path with the NEW temporal conv1
    old = pipeline.SHARED_STEM
    pipeline.SHARED_STEM = False
    try:
        c = pipeline.estimate_meshes(nets, lr[0], lr[1])          # separate stems: cannot be stale
    finally:
        pipeline.SHARED_STEM = old
    assert float((a['tmotion2'] - b['tmotion2']).abs().max()) > 1e-4
    close(b['tmotion2'], c['tmotion2'], 1e-4, 'shared stem after reload')


@pytest.mark.skipif(not torch.cuda.is_available() or torch.cuda.device_count() < 2, reason='needs two GPUs')
def test_nets_on_second_gpu_while_first_is_current():
    """Launches go to the GPU (and that GPU's current stream) that owns the tensors, not to the current device."""
    from stabstitch2_amd import pipeline
    from bench import build_nets
    torch.cuda.set_device(0)
    d1 = torch.device('cuda:1')
    nets1, _ = build_nets(d1)
    nets0, _ = build_nets(torch.device('cuda:0'))
    hr, lr = synth.make_clip_device(8, 360, 480, seed=3, device='cpu')
    r1 = pipeline.run_two_view(hr[0].to(d1), hr[1].to(d1), lr[0].to(d1), lr[1].to(d1), nets1)
    assert torch.cuda.current_device() == 0 and r1[0].device == d1
    r0 = pipeline.run_two_view(hr[0].cuda(0), hr[1].cuda(0), lr[0].cuda(0), lr[1].cuda(0), nets0)
    assert (r0[1], r0[2]) == (r1[1], r1[2]) and torch.equal(r0[3].cpu(), r1[3].cpu()) and torch.equal(r0[0].cpu(), r1[0].cpu())


@pytest.mark.parametrize('n,cin,cout,h,w,bias,res,relu', [
    (3, 64, 64, 90, 120, True, True, True),        # layer1 body (8x4 tile blocks, ragged bottom)
    (2, 128, 128, 45, 60, True, False, True),      # layer2 body: odd height (half tiles), 2 cout blocks
    (2, 256, 256, 23, 30, True, True, True),       # layer3 body: 4x8 tile blocks, odd height
    (2, 121, 64, 45, 60, False, False, True),      # cost-volume regressor conv: 121 -> 124 channels (channel tail)
    (1, 49, 64, 45, 60, False, False, True),       # 49 -> 52 channels
    (2, 64, 128, 37, 51, True, True, False),       # odd x odd map, no ReLU
    (5, 128, 64, 11, 15, False, False, True),      # small map (below the dispatch threshold, kernel must still be right)
    (1, 32, 64, 2, 2, True, False, False),         # a single tile
])
def test_conv_winograd(dev, n, cin, cout, h, w, bias, res, relu):
    """Fused Winograd F(2x2,3x3) kernel vs F.conv2d (fp32 CPU) and vs the implicit-GEMM kernel."""
    from stabstitch2_amd import ops
    rs = np.random.RandomState(n * 977 + cin + cout + h)
    x = torch.from_numpy(rs.normal(0, 1, (n, cin, h, w)).astype(np.float32))
    wt = torch.from_numpy((rs.normal(0, 1, (cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32))
    b = torch.from_numpy(rs.normal(0, 1, cout).astype(np.float32)) if bias else None
    ref = F.conv2d(x, wt, b, stride=1, padding=1)
    r = torch.from_numpy(rs.normal(0, 1, tuple(ref.shape)).astype(np.float32)) if res else None
    if res:
        ref = ref + r
    if relu:
        ref = F.relu(ref)
    cp = (cin + 3) // 4 * 4
    xd = ops.nchw_to_nhwc(x.to(dev), cp)
    rd = ops.nchw_to_nhwc(r.to(dev)) if res else None
    wd = _pack(wt, cp).to(dev)
    out = ops.conv_winograd(xd, wd, b.to(dev) if bias else None, rd, relu=relu)
    scale = max(1.0, float(ref.abs().max()))
    close(ops.nhwc_to_nchw(out), ref, 4e-5 * scale, 'winograd conv vs F.conv2d')
    old = ops.WINOGRAD
    ops.WINOGRAD = False
    try:
        direct = ops.conv(xd, wd, b.to(dev) if bias else None, rd, stride=1, pad=(0, 1, 1), relu=relu)
    finally:
        ops.WINOGRAD = old
    close(out, direct, 4e-5 * scale, 'winograd vs implicit GEMM')
    # out_cs > cout (channel-padded destination) leaves the padding untouched
    wide = torch.full((n, h, w, cout + 4), 7.0, device=dev)
    ops.conv_winograd(xd, wd, b.to(dev) if bias else None, None, relu=False, out=wide)
    assert float((wide[..., cout:] - 7.0).abs().max()) == 0.0


def test_conv_winograd_grouped_and_dispatch(dev):
    """Grouped launches (twin regressors / twin trunks) and the dispatch rule: ops.conv picks Winograd exactly where
    ss_conv_uses_winograd says so, and both engines agree on the pipeline's layer shapes."""
    from stabstitch2_amd import ops, _hip
    rs = np.random.RandomState(11)
    x = torch.from_numpy(rs.normal(0, 1, (2, 6, 45, 60, 64)).astype(np.float32)).to(dev)
    w = torch.from_numpy((rs.normal(0, 1, (2, 128, 1, 3, 3, 64)) / 24.0).astype(np.float32)).to(dev)
    b = torch.from_numpy(rs.normal(0, 1, (2, 128)).astype(np.float32)).to(dev)
    g = ops.conv_winograd(x, w, b, None, relu=True)
    for k in range(2):
        one = ops.conv_winograd(x[k], w[k].contiguous(), b[k].contiguous(), None, relu=True)
        assert torch.equal(g[k], one)
    shared = ops.conv_winograd(x[0], w, b, None, relu=True)          # one input read by both groups
    assert torch.equal(shared[0], g[0])
    lib = _hip.lib()
    assert lib.ss_conv_uses_winograd(1, 3, 3, 1, 64, 64, 90, 120, 64) == 1
    assert lib.ss_conv_uses_winograd(1, 3, 3, 2, 64, 128, 45, 60, 64) == 0       # strided
    assert lib.ss_conv_uses_winograd(1, 7, 7, 2, 4, 64, 180, 240, 64) == 0       # conv1
    assert lib.ss_conv_uses_winograd(5, 3, 3, 1, 128, 128, 7, 9, 26) == 0        # Conv3d
    assert lib.ss_conv_uses_winograd(1, 3, 3, 1, 4, 64, 23, 30, 32) == 0         # 2-channel flow input
    xl = torch.from_numpy(rs.normal(0, 1, (64, 45, 60, 128)).astype(np.float32)).to(dev)
    wl = torch.from_numpy((rs.normal(0, 1, (128, 1, 3, 3, 128)) / 34.0).astype(np.float32)).to(dev)
    a = ops.conv(xl, wl, None, None, relu=True)                        # large launch on a 60-wide map: F(4x4,3x3) (round 4)
    assert ops.last_conv_path == 'wino43'
    assert torch.equal(a, ops.conv_winograd43(xl, wl, None, None, relu=True))
    f23 = ops.conv_winograd(xl, wl, None, None, relu=True)
    close(a, f23, 1e-4 * max(1.0, float(f23.abs().max())), 'dispatch: F(4x4,3x3) vs F(2x2,3x3)')
    if ops.WINO43 == 'auto':
        a6 = ops.conv(xl[:6], wl, None, None, relu=True)               # a shallow launch of the same layer stays on F(2x2,3x3)
        assert ops.last_conv_path == 'wino' and torch.equal(a6, f23[:6])
    old = ops.WINOGRAD
    ops.WINOGRAD = False
    try:
        d = ops.conv(xl, wl, None, None, relu=True)
    finally:
        ops.WINOGRAD = old
    close(a, d, 1e-4 * max(1.0, float(d.abs().max())), 'dispatch: winograd vs implicit GEMM')


def test_conv_winograd_schedules_bit_identical(dev, request):
    """The dispatched stream schedule of the Winograd K loop against the phase-alternating schedule it replaced (kept in
    the tools/ build, ss_debug_set(7, 1)): every accumulator receives its products in the same order, so the outputs
    must be EQUAL -- ragged maps, channel tails, residual / ReLU, grouped launches; and the product build must match."""
    from stabstitch2_amd import ops, _hip
    import importlib.util
    spec = importlib.util.spec_from_file_location('_tuning', os.path.join(os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), 'tools', '_tuning.py'))
    tuning = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tuning)
    product = _hip.lib()
    request.addfinalizer(lambda: setattr(_hip, '_lib', product))
    lib = tuning.lib()
    request.addfinalizer(lambda: lib.ss_debug_set(7, 0))
    rs = np.random.RandomState(77)
    for (n, cin, cout, h, w, g) in ((64, 64, 64, 90, 120, 1), (16, 128, 128, 45, 60, 1), (8, 256, 256, 23, 30, 1),
                                    (3, 36, 64, 37, 53, 1), (1, 32, 128, 8, 8, 1), (5, 160, 64, 11, 15, 2), (2, 20, 64, 3, 5, 1),
                                    (2, 16, 64, 9, 7, 1), (1, 4, 64, 5, 5, 1)):        # one chunk: the peeled last chunk is the first
        shape = (g, n, h, w, cin) if g > 1 else (n, h, w, cin)
        x = torch.from_numpy(rs.normal(0, 1, shape).astype(np.float32)).to(dev)
        wt = torch.from_numpy((rs.normal(0, 1, ((g,) if g > 1 else ()) + (cout, 1, 3, 3, cin)) / np.sqrt(9 * cin)).astype(np.float32)).to(dev)
        b = torch.from_numpy(rs.normal(0, 1, ((g,) if g > 1 else ()) + (cout,)).astype(np.float32)).to(dev)
        r = torch.from_numpy(rs.normal(0, 1, shape[:-1] + (cout,)).astype(np.float32)).to(dev)
        for (res, relu) in ((None, False), (r, True)):
            outs = []
            for variant in (0, 1):
                lib.ss_debug_set(7, variant)
                outs.append(ops.conv_winograd(x, wt, b, res, relu=relu))
            lib.ss_debug_set(7, 0)
            assert torch.equal(outs[0], outs[1]), ('stream vs alternating schedule', (n, cin, cout, h, w, g, relu))
            _hip._lib = product
            prod = ops.conv_winograd(x, wt, b, res, relu=relu)
            _hip._lib = lib
            assert torch.equal(prod, outs[0]), 'product and tuning builds differ'


def test_wino_pack3_slices_sum_exactly(dev):
    """ss_wino_pack3 against ss_wino_pack: the three bf16 slices of every transformed filter value add up to EXACTLY the
    fp32 value the fp32 kernel multiplies with (fp32 adds of the slices are exact: 8 + 8 + 8 significand bits), for a
    channel tail (cin = 36 -> 3 chunks, zero padding) and a grouped pack."""
    from stabstitch2_amd import ops
    rs = np.random.RandomState(9)
    for (g, cout, cin) in ((1, 64, 36), (2, 128, 64), (1, 64, 256)):
        wt = torch.from_numpy(rs.normal(0, 1, ((g,) if g > 1 else ()) + (cout, 1, 3, 3, cin)).astype(np.float32)).to(dev)
        if g == 1:
            wt_ = wt
        else:
            wt_ = wt
        p1 = ops.wino_packed(wt_, g, sliced=False)
        p3 = ops.wino_packed(wt_, g, sliced=True)
        nchunk = (cin + 15) // 16
        a = p1.view(g, cout // 32, nchunk, 16, 2, 64, 4).permute(0, 1, 2, 3, 5, 4, 6).reshape(g, cout // 32, nchunk, 16, 64, 8)
        b = p3.view(torch.bfloat16).view(g, cout // 32, nchunk, 16, 3, 64, 8).float()
        s12 = b[:, :, :, :, 0] + b[:, :, :, :, 1]
        tot = s12 + b[:, :, :, :, 2]
        assert torch.equal(tot, a), (g, cout, cin, float((tot - a).abs().max()))
        assert float(b[:, :, :, :, 1].abs().max()) <= float(b[:, :, :, :, 0].abs().max()) * 2.0 ** -7      # the slices descend


def test_conv_winograd_bf16x9_products(dev, golden, hip_nets, clip16, request):
    """The opt-in arithmetic of the Winograd GEMMs (ops.WINO_MATH = 'bf16x9': every fp32 operand as three exact bf16
    slices, all nine slice products on the bf16 matrix pipe, fp32 accumulation): against an fp64 convolution it must be
    no worse than the fp32-MFMA kernel (it forms every product exactly; only the accumulation order differs), and the
    reference's meshes (G9) must hold at the SAME gates as the default arithmetic."""
    from stabstitch2_amd import ops, pipeline
    old = ops.WINO_MATH
    request.addfinalizer(lambda: setattr(ops, 'WINO_MATH', old))
    rs = np.random.RandomState(41)
    for (n, cin, cout, h, w, g) in ((4, 64, 64, 90, 120, 1), (3, 36, 64, 37, 53, 1), (2, 256, 256, 23, 30, 1), (5, 160, 64, 11, 15, 2),
                                    (2, 16, 64, 9, 7, 1)):
        shape = (g, n, h, w, cin) if g > 1 else (n, h, w, cin)
        x = torch.from_numpy(np.maximum(rs.normal(0, 1, shape), 0).astype(np.float32)).to(dev)
        wt = torch.from_numpy((rs.normal(0, 1, ((g,) if g > 1 else ()) + (cout, 1, 3, 3, cin)) * np.sqrt(2.0 / (9 * cin))).astype(np.float32)).to(dev)
        b = torch.from_numpy(rs.normal(0, 1, ((g,) if g > 1 else ()) + (cout,)).astype(np.float32)).to(dev)
        r = torch.from_numpy(rs.normal(0, 1, shape[:-1] + (cout,)).astype(np.float32)).to(dev)
        xs, ws, bs, rs_ = (x, wt, b, r) if g == 1 else (x[1], wt[1], b[1], r[1])
        ref = F.conv2d(xs.permute(0, 3, 1, 2).double().cpu(), ws[:, 0].permute(0, 3, 1, 2).double().cpu(), bs.double().cpu(),
                       padding=1).permute(0, 2, 3, 1) + rs_.double().cpu()
        err = {}
        for math in ('f32', 'bf16x9'):
            ops.WINO_MATH = math
            o = ops.conv_winograd(x, wt, b, r, relu=False)
            o = o if g == 1 else o[1]
            err[math] = float((o.double().cpu() - ref).abs().max())
        ops.WINO_MATH = old
        scale = max(1.0, float(ref.abs().max()))
        assert err['bf16x9'] <= 1e-6 * scale, (err, scale)          # observed 2-3e-7 x scale for both
        assert err['bf16x9'] <= 2.0 * err['f32'] + 1e-7 * scale, err
    ops.WINO_MATH = 'bf16x9'
    gg = golden('g9_pipeline')
    hr, lr = clip16
    acc = pipeline.estimate_meshes(hip_nets, lr[0], lr[1])
    close(acc['smooth_mesh1'], gg['smooth_mesh1'], 5e-3, 'smooth_mesh1 (bf16x9 products)')
    close(acc['smooth_mesh2'], gg['smooth_mesh2'], 5e-3, 'smooth_mesh2 (bf16x9 products)')
    close(acc['smooth_path2'], gg['smooth_path2'], 1e-2, 'smooth_path2 (bf16x9 products)')


def test_ingest_u8_vs_handworked_cv2_vectors(dev):
    """The HIP front-end against the scalar hand derivation of OpenCV's uint8 INTER_LINEAR (tests/golden/
    cv2_resize_handworked.json, independent of oracle/frame_io.py): lr = resize / 127.5 - 1, bit for bit."""
    import json
    from stabstitch2_amd import ops
    d = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'cv2_resize_handworked.json')))
    for name, c in d['cases'].items():
        src, dst = np.array(c['src'], np.uint8), np.array(c['dst'], np.uint8)
        dw, dh = c['dsize']
        hr, lr = ops.ingest_u8(torch.from_numpy(src[None]).to(dev), dh, dw)
        want = (dst.astype(np.float32).transpose(2, 0, 1) / np.float32(127.5)) - np.float32(1.0)
        assert np.array_equal(lr[0].cpu().numpy(), want.astype(np.float32)), name
        assert np.array_equal(hr[0].cpu().numpy(), src.astype(np.float32).transpose(2, 0, 1)), name


def test_tsmotion_cached_rigid_inverse(dev, golden):
    """tsmotion always solves its TPS systems from the RIGID mesh: with the cached fp64 W^-1 (ss_tps_inverse, the
    reference's own torch.inverse + matmul formulation) the result equals the per-frame elimination and the reference."""
    from stabstitch2_amd import ops
    g8 = golden('g8_nets')
    sm, tm = torch.from_numpy(g8['motion1']).to(dev), torch.from_numpy(g8['tmotion1']).to(dev)
    a_mesh, a = ops.tsmotion(sm, tm)
    old = ops.RIGID_INVERSE_CACHE
    ops.RIGID_INVERSE_CACHE = False
    try:
        b_mesh, b = ops.tsmotion(sm, tm)
    finally:
        ops.RIGID_INVERSE_CACHE = old
    assert torch.equal(a_mesh, b_mesh)
    close(a, b, 2e-5, 'tsmotion: cached inverse vs per-frame elimination')
    close(a, g8['tsmotion1'], 2e-3, 'tsmotion (cached inverse) vs reference')
    # W^-1 W = I for the rigid control points (fp64)
    winv = ops.rigid_winv(360, 480, dev)
    assert winv.dtype == torch.float64 and tuple(winv.shape) == (66, 66)
    nr = torch.from_numpy(cases.norm(cases.rigid(360, 480), 360, 480)[0]).double()
    d2 = ((nr[:, None, :] - nr[None, :, :]) ** 2).sum(2).float()
    r = (d2 * torch.log(d2 + 1e-6)).double()
    p = torch.cat((torch.ones(63, 1, dtype=torch.float64), nr), 1)
    W = torch.cat((torch.cat((p, r), 1), torch.cat((torch.zeros(3, 3, dtype=torch.float64), p.t()), 1)), 0)
    # (W here uses torch's fp32 log, the kernel a correctly rounded one: the entries differ in the last bit, ~1e-6 through W^-1)
    assert float((winv.cpu() @ W - torch.eye(66, dtype=torch.float64)).abs().max()) < 1e-4


@pytest.mark.parametrize('n,h,w,cout', [(2, 72, 96, 64), (3, 45, 61, 128), (1, 360, 480, 64), (1, 7, 9, 64)])
def test_conv_stem_row_packed(dev, n, h, w, cout):
    """ss_conv_stem3 (7x7 / 2 / pad 3 on the row-packed 3-channel layout, K = 7 x 24) vs F.conv2d + folded BN + ReLU, and
    vs the 4-channel NHWC form of the same layer; grouped launch = two stems on the same frames."""
    from stabstitch2_amd import ops, layers as L
    rs = np.random.RandomState(h * 31 + w)
    x = torch.from_numpy(rs.normal(0, 1, (n, 3, h, w)).astype(np.float32))
    conv = torch.nn.Conv2d(3, cout, 7, 2, 3, bias=False)
    bn = torch.nn.BatchNorm2d(cout).eval()
    with torch.no_grad():
        conv.weight.copy_(torch.from_numpy((rs.normal(0, 1, (cout, 3, 7, 7)) / 12.0).astype(np.float32)))
        bn.running_mean.normal_(0, 0.1); bn.running_var.uniform_(0.5, 2); bn.weight.uniform_(0.8, 1.2); bn.bias.normal_(0, 0.1)
    ref = F.relu(bn(conv(x)))
    wgt, bias = L.pack_stem3(conv, bn)
    assert tuple(wgt.shape) == (cout, 7, 24) and float(wgt[:, :, 21:].abs().max()) == 0.0
    buf = ops.stem_input([x[:1].to(dev), x[1:].to(dev)] if n > 1 else x.to(dev))
    assert tuple(buf.shape) == (n, h, w + 8, 3) and float(buf[:, :, :3].abs().max()) == 0.0 and float(buf[:, :, w + 3:].abs().max()) == 0.0
    out = ops.conv_stem(buf, wgt.to(dev), bias.to(dev), relu=True)
    close(ops.nhwc_to_nchw(out), ref, 2e-5 * max(1.0, float(ref.abs().max())), 'stem conv vs F.conv2d')
    w4, b4 = L.pack_conv2d(conv, bn)
    old = ops.conv(ops.nchw_to_nhwc(x.to(dev), 4), w4.to(dev), b4.to(dev), stride=2, pad=(0, 3, 3), relu=True)
    close(out, old, 2e-5 * max(1.0, float(ref.abs().max())), 'stem conv vs 4-channel NHWC form')
    w2 = torch.stack((wgt, wgt.flip(0)), 0).contiguous().to(dev)
    b2 = torch.stack((bias, bias.flip(0)), 0).contiguous().to(dev)
    g = ops.conv_stem(buf, w2, b2, relu=True)
    assert torch.equal(g[0], out) and torch.equal(g[1], out.flip(-1))


@pytest.mark.parametrize('views', [2, 3])
def test_render_footprint_skipping(dev, hip_nets, views):
    """AVERAGE render with footprints (a view's spline is skipped on tiles it provably cannot reach, its contribution
    taken as exactly 0) against the full evaluation: bit-identical wherever no view was skipped, within the rounding
    residue of the clamped sampler (<= 5e-3 grey levels) wherever some view is valid, and a skipped view is really
    outside: the full evaluation's validity mask is ~0 on every pixel of a skipped tile."""
    from stabstitch2_amd import ops, pipeline
    n, h, w = 7, 720, 1280
    hr, lr = synth.make_clip_device(n, h, w, seed=3, views=views, device=dev)
    if views == 2:
        acc = pipeline.estimate_meshes(hip_nets, lr[0], lr[1])
        meshes, pres = [acc['smooth_mesh1'], acc['smooth_mesh2']], False
    else:
        a12 = pipeline.estimate_meshes(hip_nets, lr[0], lr[1])
        a23 = pipeline.estimate_meshes(hip_nets, lr[1], lr[2])
        meshes = list(pipeline.three_view_compose(a12['smooth_mesh1'], a12['smooth_mesh2'], a23['smooth_mesh1'],
                                                  a23['smooth_mesh2'], h, w))
        pres = True
    hc, wc, src, T = pipeline.render_plan(meshes, h, w, pres)
    fp = ops.render_footprints(src, T, h, w, hc, wc)
    # lattice: rows every 8 px, columns every 32 px (tile corners + long-edge midpoints); nbx tiles per row
    nbx = (wc + 63) // 64
    ny, nx = (hc + 7) // 8 + 1, 2 * nbx + 1
    assert fp.shape == (n, views * ny * nx * 2 + views * 4 + 4 + 4 * (ny - 1) * nbx)         # lattice, hulls, 4 class counters + lists
    assert bool(torch.isfinite(fp[:, :views * ny * nx * 2 + views * 4]).all())
    base = views * ny * nx * 2 + views * 4
    nt = (ny - 1) * nbx
    for i in range(n):                                     # the tile order: every tile in exactly one class list, classes by view count
        words = fp[i, base:].view(torch.int32).cpu().numpy().astype(np.int64) & 0xFFFFFFFF
        cnt, lists = words[:4], words[4:].reshape(4, nt)
        assert int(cnt.sum()) == nt
        seen = np.concatenate([lists[k, :cnt[k]] for k in range(4)])
        assert len(np.unique((seen & 0xFFF) + ((seen >> 12) & 0xFFF) * nbx)) == nt
        for k in range(4):
            assert all(bin(int(e >> 24)).count('1') == 3 - k for e in lists[k, :cnt[k]][:50])
    skipped_frac = []
    for i in (0, n - 1):
        imgs = [hr[k][i] for k in range(views)]
        full = ops.render_average(imgs, src[i], T[i], hc, wc, 'NORMAL')
        skip = ops.render_average(imgs, src[i], T[i], hc, wc, 'NORMAL', footprint=fp[i])
        wm = ops.tps_warp(torch.stack(imgs, 0), src[i], T[i], hc, wc, 'NORMAL', with_mask=True)[:, 3]      # [V,hc,wc]
        # re-derive the tile classification on the host from the footprint block
        lat = fp[i, :views * ny * nx * 2].view(views, ny, nx, 2)
        hull = fp[i, views * ny * nx * 2:views * ny * nx * 2 + 4 * views].view(views, 4)
        need = torch.ones((views, ny - 1, nbx), dtype=torch.bool, device=dev)
        bxs = torch.arange(nbx, device=dev, dtype=torch.float32)
        bys = torch.arange(ny - 1, device=dev, dtype=torch.float32)
        tx0 = -1 + 2.0 / (wc - 1) * (64 * bxs - 8); tx1 = -1 + 2.0 / (wc - 1) * (64 * bxs + 71)
        ty0 = -1 + 2.0 / (hc - 1) * (8 * bys - 8); ty1 = -1 + 2.0 / (hc - 1) * (8 * bys + 15)
        for v in range(views):
            off_hull = ((tx1 < hull[v, 0]) | (tx0 > hull[v, 1]))[None, :] | ((ty1 < hull[v, 2]) | (ty0 > hull[v, 3]))[:, None]
            off_img = torch.zeros_like(off_hull)
            for c, m in ((0, 1.0 + 16.0 / w), (1, 1.0 + 16.0 / h)):
                q = torch.stack([lat[v, r:ny - 1 + r, k:k + 2 * nbx:2, c] for r in (0, 1) for k in (0, 1, 2)], 0)
                off_img |= (q.min(0).values > m) | (q.max(0).values < -m)
            need[v] = ~(off_hull & off_img)
        tile = need.repeat_interleave(8, 1).repeat_interleave(64, 2)[:, :hc, :wc]                           # per pixel
        skipped_frac.append(1.0 - float(tile.float().mean()))
        # (1) a skipped view is outside: its validity mask is ~0 on all those pixels
        assert float(wm[~tile].abs().max()) < 1e-2, float(wm[~tile].abs().max())
        # (2) nothing skipped -> bit identical
        allv = tile.all(0)
        assert torch.equal(skip[:, allv], full[:, allv])
        # (3) some view valid -> equal up to the residue the skipped views would have contributed
        # (three views: where only view 3 is valid the reference's chained AVERAGE is chaotic -- avg(residue, residue) feeds
        #  the second fusion, DESIGN.md 4 -- so the sharp comparison covers the pixels views 1 or 2 reach, the rest the median)
        anyvalid = (wm[:2] > 0.5).any(0)
        d = (skip - full).abs()[:, anyvalid]
        assert float(d.max()) < 5e-2, float(d.max())
        rest = (wm > 0.5).any(0) & ~anyvalid
        if bool(rest.any()):
            assert float((skip - full).abs()[:, rest].median()) < 2e-2
        # (4) no view reaches the tile -> exactly 0
        none = ~tile.any(0)
        assert float(skip[:, none].abs().max()) == 0.0 if bool(none.any()) else True
    assert min(skipped_frac) > 0.25, skipped_frac             # a quarter of the (pixel, view) pairs at least, on this geometry
    if os.environ.get('SS_VERBOSE'):
        print('  skipped (pixel, view) fraction: %s' % skipped_frac)
