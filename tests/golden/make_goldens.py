"""Generate the golden fixtures by RUNNING THE REFERENCE (CPU) in the build container.

    cd /root/repo && python tests/golden/make_goldens.py

Imports /root/reference/Full_model_inference/Codes under tests/golden/ref_shim.py, feeds it the
seeded inputs of tests/golden/cases.py and the synthetic checkpoints of stabstitch2_amd/synth.py,
and writes small .npz files next to this script.  Only the .npz outputs travel; the reference
itself never does.  G11 (PSNR/SSIM) is produced by scikit-image 0.18.3 in /opt/conda/bin/python3.9.
"""
import json
import os
import subprocess
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_shim  # noqa: E402
ref_shim.install()

import cases  # noqa: E402
import manifest  # noqa: E402
from stabstitch2_amd import synth  # noqa: E402

import spatial_network as RS  # noqa: E402   (reference)
import temporal_network as RT  # noqa: E402
import smooth_network as RM  # noqa: E402
import test_online_tra as RP  # noqa: E402
import test_metric_ssd as RMET  # noqa: E402
import utils.torch_DLT as R_DLT  # noqa: E402
import utils.torch_homo_transform as R_HOMO  # noqa: E402
import utils.torch_tps_transform as R_TPS  # noqa: E402
import utils.torch_tps_transform_point as R_TPSP  # noqa: E402

torch.set_grad_enabled(False)


def save(name, **arrs):
    assert sorted(arrs) == sorted(manifest.KEYS[name]), \
        '%s: keys differ from tests/golden/manifest.py: %s' % (name, sorted(set(arrs) ^ set(manifest.KEYS[name])))
    out = {}
    for k, v in arrs.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print('%-28s %8.1f KB' % (name + '.npz', os.path.getsize(path) / 1024))


def ref_nets(profile='default'):
    sp, tp, sm = RS.SpatialNet().eval(), RT.TemporalNet().eval(), RM.SmoothNet().eval()
    for m in (sp, tp, sm):
        m.load_state_dict(synth.synthetic_state_dict(m, profile=profile), strict=True)
    return sp, tp, sm


# ------------------------------------------------------------------ G1 DLT + decomposition
def g1():
    off = cases.g1_offsets()
    b = off.shape[0]
    outs = {}
    for tag, scale in (('full', 1.0), ('feat', 8.0)):
        src = torch.tensor([[0., 0.], [480., 0.], [0., 360.], [480., 360.]]).unsqueeze(0).expand(b, -1, -1)
        m = off.reshape(b, 4, 2)
        H = R_DLT.tensor_DLT(src / scale, (src + m) / scale)
        H_tgt = R_DLT.tensor_DLT(src / scale, (src + m / 2.) / scale)
        H_ref = torch.matmul(torch.inverse(H), H_tgt)
        outs.update({'H_' + tag: H, 'H_tgt_' + tag: H_tgt, 'H_ref_' + tag: H_ref})
    rigid = RS.get_rigid_mesh(b, 360, 480)
    outs['mesh_ref'] = RS.H2Mesh(outs['H_ref_full'], rigid)
    outs['mesh_tgt'] = RS.H2Mesh(outs['H_tgt_full'], rigid)
    outs['rigid'] = rigid
    outs['norm_rigid'] = RS.get_norm_mesh(rigid, 360, 480)
    save('g1_dlt', **outs)


# ------------------------------------------------------------------ G2 homography sampler
def g2():
    U, thetas = cases.g2_inputs()
    out = R_HOMO.transformer(U, thetas, (45, 60))
    out_small = R_HOMO.transformer(U, thetas, (23, 31))
    save('g2_homo', out=out, out_small=out_small)


# ------------------------------------------------------------------ G3 cost volume
def g3():
    a, b = cases.g3_inputs(False)
    cv5 = RS.SpatialNet.cost_volume(a, b, search_range=5, norm=False)
    cv3 = RT.TemporalNet.cost_volume(a, b, search_range=3, norm=False)
    fa, fb = cases.g3_inputs(True)
    full5 = RS.SpatialNet.cost_volume(fa, fb, search_range=5, norm=False)
    full3 = RT.TemporalNet.cost_volume(fa, fb, search_range=3, norm=False)
    # norm=True (the signature's default; spatial_network.py:335-337) -- never used by the inference path, pinned anyway
    cv5n = RS.SpatialNet.cost_volume(a, b, search_range=5, norm=True)
    cv3n = RT.TemporalNet.cost_volume(a, b, search_range=3, norm=True)
    save('g3_costvol', cv5=cv5, cv3=cv3, full5_chsum=full5.sum(dim=(2, 3)), full3_chsum=full3.sum(dim=(2, 3)),
         full5_rows=full5[0, :, 22, :], full3_rows=full3[0, :, 0, :], cv5n=cv5n, cv3n=cv3n)


# ------------------------------------------------------------------ G4 CCL
def g4(sp):
    a, b = cases.g4_inputs(False)
    fa, fb = cases.g4_inputs(True)
    save('g4_ccl', flow=sp.CCL(a, b), flow_full=sp.CCL(fa, fb))


# ------------------------------------------------------------------ G5 TPS solve / points
def g5():
    nrigid, warped, query = cases.g5_meshes()
    # (a) tsmotion convention: source = rigid, target = warped, evaluated at query
    p_a = R_TPSP.transformer(query, nrigid, warped)
    # (b) render convention: source = warped, target = rigid
    p_b = R_TPSP.transformer(query, warped, nrigid)
    # T is not returned by the reference API; recover it exactly from the affine+RBF basis by
    # evaluating the spline at the control points is not possible, so pin T through outputs only.
    save('g5_tps_points', p_a=p_a, p_b=p_b)


# ------------------------------------------------------------------ G6 / G7 TPS dense warp, fusion
def g6_g7():
    U, src, tgt, size, ident = cases.g6_inputs()
    wn = R_TPS.transformer(U, src, tgt, size, 'NORMAL')
    wf = R_TPS.transformer(U, src, tgt, size, 'FAST')
    idn = R_TPS.transformer(U, ident, tgt, (72, 96), 'NORMAL')
    idf = R_TPS.transformer(U, ident, tgt, (72, 96), 'FAST')
    save('g6_tps_warp', normal=wn, fast=wf, ident_normal=idn, ident_fast=idf)
    # fusion on the NORMAL warps: AVERAGE on colour channels, LINEAR with a warped ones-mask
    avg = wn[0, 0:3] * (wn[0, 0:3] / (wn[0, 0:3] + wn[1, 0:3] + 1e-6)) + \
        wn[1, 0:3] * (wn[1, 0:3] / (wn[0, 0:3] + wn[1, 0:3] + 1e-6))
    one = torch.ones_like(U[:, 0:1])
    wm = R_TPS.transformer(torch.cat((U[:, 0:3], one), 1), src, tgt, size, 'NORMAL')
    lin = RP.linear_blender(wm[0:1, 0:3], wm[1:2, 0:3], wm[0:1, 3:4], wm[1:2, 3:4])
    mask1 = RP.linear_blender(wm[0:1, 0:3], wm[1:2, 0:3], wm[0:1, 3:4], wm[1:2, 3:4], mask=True)
    save('g7_fusion', warped_with_mask=wm, average=avg, linear=lin, mask1=mask1)


# ------------------------------------------------------------------ G8 nets end to end
def run_motion_stages(nets, lr1, lr2):
    """Tensor-level replay of test() in test_online_tra.py:284-392 using the reference's functions."""
    sp, tp, sm = nets
    n = len(lr1)
    s1, s2 = [], []
    for k in range(n):
        o = RS.build_SpatialNet(sp, lr1[k], lr2[k])
        s1.append(o['motion1'])
        s2.append(o['motion2'])
    t1 = RT.build_TemporalNet(tp, lr1)['motion_list']
    t2 = RT.build_TemporalNet(tp, lr2)['motion_list']
    rigid = RP.get_rigid_mesh(1, 360, 480)
    nrigid = RP.get_norm_mesh(rigid, 360, 480)

    def prep(s, t):
        smesh, tsm = [], []
        for k in range(n):
            sm_k = rigid + s[k]
            if k == 0:
                ts = s[k].clone() * 0
            else:
                prev = RP.get_norm_mesh(rigid + s[k - 1], 360, 480)
                tm = RP.get_norm_mesh(rigid + t[k], 360, 480)
                ts = RP.recover_mesh(R_TPSP.transformer(tm, nrigid, prev), 360, 480) - sm_k
            smesh.append(sm_k)
            tsm.append(ts)
        return smesh, tsm
    sm1, ts1 = prep(s1, t1)
    sm2, ts2 = prep(s2, t2)

    acc = None
    first = None
    for k in range(n - 6):
        a = ts1[k:k + 7]
        a[0] = a[0] * 0
        b = ts2[k:k + 7]
        b[0] = b[0] * 0
        o = RM.build_SmoothNet(sm, a, b, sm1[k:k + 7], sm2[k:k + 7])
        if k == 0:
            first = o
            acc = {key: o[key] for key in ('ori_mesh1', 'smooth_mesh1', 'ori_mesh2', 'smooth_mesh2',
                                           'ori_path2', 'smooth_path2')}
        else:
            for key in ('ori_mesh1', 'smooth_mesh1', 'ori_mesh2', 'smooth_mesh2'):
                acc[key] = torch.cat((acc[key], o[key][:, -1, ...].unsqueeze(1)), 1)
            new_ori = acc['ori_path2'][:, -1, ...] + (o['ori_path2'][:, -1, ...] - o['ori_path2'][:, -2, ...])
            acc['ori_path2'] = torch.cat((acc['ori_path2'], new_ori.unsqueeze(1)), 1)
            new_sm = acc['ori_path2'][:, -1, ...] + (o['smooth_path2'][:, -1, ...] - o['ori_path2'][:, -1, ...])
            acc['smooth_path2'] = torch.cat((acc['smooth_path2'], new_sm.unsqueeze(1)), 1)
    return dict(s1=s1, s2=s2, t1=t1, t2=t2, ts1=ts1, ts2=ts2, first=first, acc=acc)


def g8_g9(nets):
    sp, tp, sm = nets
    hr, lr = synth.make_clip(16, 360, 480, seed=0)
    o1, o2r, o2t = sp(lr[0][0], lr[1][0])
    # batch-of-2 forward as well (eval mode is batch invariant)
    st = run_motion_stages(nets, lr[0], lr[1])
    save('g8_nets', offset_1=o1, offset_2_ref=o2r, offset_2_tgt=o2t,
         motion1=torch.cat(st['s1'], 0), motion2=torch.cat(st['s2'], 0),
         tmotion1=torch.cat(st['t1'], 0), tmotion2=torch.cat(st['t2'], 0),
         tsmotion1=torch.cat(st['ts1'], 0), tsmotion2=torch.cat(st['ts2'], 0),
         **{'w0_' + k: v for k, v in st['first'].items()})

    acc = st['acc']
    g9 = dict(smooth_mesh1=acc['smooth_mesh1'], smooth_mesh2=acc['smooth_mesh2'],
              ori_mesh2=acc['ori_mesh2'], ori_path2=acc['ori_path2'], smooth_path2=acc['smooth_path2'])
    for wm, fm in (('NORMAL', 'AVERAGE'), ('FAST', 'AVERAGE'), ('NORMAL', 'LINEAR')):
        frames, ow, oh = RP.get_stable_sqe(hr[0], hr[1], acc['smooth_mesh1'], acc['smooth_mesh2'], wm, fm)
        tag = '%s_%s' % (wm.lower(), fm.lower())
        g9['canvas_' + tag] = np.array([int(oh), int(ow)])
        g9['frames_' + tag] = np.stack([cases.box_down(f, 16) for f in frames])
        g9['iqr_' + tag] = np.stack([cases.box_iqr(f, 16) for f in frames])
        if tag == 'normal_average':
            g9['frame0_crop'] = frames[0][150:214, 300:396].copy()   # full-resolution crop across the seam
    # metric harness: LR warps with masks -> PSNR/SSIM by scikit-image 0.18.3
    l1, l2 = RMET.get_stable_sqe(lr[0], lr[1], acc['smooth_mesh1'], acc['smooth_mesh2'])
    ps = skimage_metrics([(a[..., 0:3] * (a[..., 3:6] * b[..., 3:6]), b[..., 0:3] * (a[..., 3:6] * b[..., 3:6]))
                          for a, b in zip(l1, l2)])
    g9['psnr'] = np.array([p for p, _ in ps])
    g9['ssim'] = np.array([s for _, s in ps])
    g9['lr_warp1_frame3'] = cases.box_down(l1[3], 4)
    # stability / distortion exactly as test_metric_ssd.py:444-482
    sp2 = acc['smooth_path2']
    L = RMET.l_num_loss
    mid = sp2[:, 3:-3]
    stab = (L(sp2[:, :-6], mid, 2) + L(sp2[:, 6:], mid, 2)) * 0.1
    stab += (L(sp2[:, 1:-5], mid, 2) + L(sp2[:, 5:-1], mid, 2)) * 0.3
    stab += (L(sp2[:, 2:-4], mid, 2) + L(sp2[:, 4:-2], mid, 2)) * 0.9
    dist = max((1 * RMET.inter_grid_loss(acc['smooth_mesh2'][:, k, ...].unsqueeze(1))
                + 1 * RMET.intra_grid_loss(acc['smooth_mesh2'][:, k, ...].unsqueeze(1))).item()
               for k in range(acc['smooth_mesh2'].shape[1]))
    g9['stability'] = np.array(stab.item())
    g9['distortion'] = np.array(dist)
    save('g9_pipeline', **g9)
    return acc


def skimage_metrics(pairs):
    """PSNR/SSIM by skimage 0.18.3 (metrics.peak_signal_noise_ratio / structural_similarity are the
    renamed 0.15 compare_psnr / compare_ssim with the same defaults) in the conda interpreter."""
    with tempfile.TemporaryDirectory() as td:
        np.savez(os.path.join(td, 'in.npz'), **{'a%d' % i: p[0] for i, p in enumerate(pairs)},
                 **{'b%d' % i: p[1] for i, p in enumerate(pairs)})
        code = (
            "import numpy as np, json, sys\n"
            "from skimage.metrics import peak_signal_noise_ratio as P, structural_similarity as S\n"
            "d = np.load(sys.argv[1]); n = len(d.files)//2; out = []\n"
            "for i in range(n):\n"
            "    a, b = d['a%d'%i], d['b%d'%i]\n"
            "    out.append([float(P(a, b, data_range=255)), float(S(a, b, data_range=255, multichannel=True))])\n"
            "print(json.dumps(out))\n")
        r = subprocess.run(['/opt/conda/bin/python3.9', '-c', code, os.path.join(td, 'in.npz')],
                           capture_output=True, text=True, check=True)
        return json.loads(r.stdout.strip().splitlines()[-1])


# ------------------------------------------------------------------ G10 three-view
def g10():
    import test_online_tra_threeview as R3
    src_lines = open(os.path.join(ref_shim.REF, 'test_online_tra_threeview.py')).read().split('\n')
    # lines 345..505 of test(): from "# resize the mesh to the original resolution" through
    # stable_list.append(...) -- executed verbatim with prepared locals
    body = '\n'.join(l[4:] if l.startswith('    ') else l for l in src_lines[344:505])
    m12_1, m12_2, m23_1, m23_2 = cases.g10_meshes()
    n = m12_1.shape[1]
    hr, _ = synth.make_clip(n, 180, 320, seed=3, views=3)
    res = {}
    for wm, fm in (('NORMAL', 'AVERAGE'), ('NORMAL', 'LINEAR')):
        ns = dict(vars(R3))
        ns.update(warp12_mesh1=m12_1.clone(), warp12_mesh2=m12_2.clone(), warp23_mesh1=m23_1.clone(),
                  warp23_mesh2=m23_2.clone(), img1_list=hr[0], img2_list=hr[1], img3_list=hr[2],
                  args=types.SimpleNamespace(warp_mode=wm, fusion_mode=fm))
        if fm == 'LINEAR':
            # record what the reference's blender sees and decides, per call (two chained calls per frame, threeview:498-501):
            # the warped masks, the nonzero counts and centroids its weights hang on, and its mask1
            calls = []
            ref_blender = ns['linear_blender']

            def spy(ref, tgt, ref_m, tgt_m, mask=False, _f=ref_blender, _calls=calls):
                r1, c1 = torch.nonzero(ref_m[0, 0], as_tuple=True)
                r2, c2 = torch.nonzero(tgt_m[0, 0], as_tuple=True)
                _calls.append(dict(count=np.array([r1.numel(), r2.numel()]),
                                   center=np.array([float(r1.float().mean()), float(c1.float().mean()),
                                                    float(r2.float().mean()), float(c2.float().mean())]),
                                   ref_m=cases.box_down(ref_m[0, 0].numpy()[..., None], 4)[..., 0],
                                   tgt_m=cases.box_down(tgt_m[0, 0].numpy()[..., None], 4)[..., 0],
                                   mask1=cases.box_down(_f(ref, tgt, ref_m, tgt_m, True)[0, 0].numpy()[..., None], 4)[..., 0]))
                return _f(ref, tgt, ref_m, tgt_m, mask)
            ns['linear_blender'] = spy
        exec(compile(body, 'threeview_345_505', 'exec'), ns)
        tag = fm.lower()
        if fm == 'LINEAR':
            assert len(calls) == 2 * n
            for key in ('count', 'center', 'ref_m', 'tgt_m', 'mask1'):
                res['lin_' + key] = np.stack([np.stack([calls[2 * i][key], calls[2 * i + 1][key]]) for i in range(n)])
        res['canvas_' + tag] = np.array([int(ns['out_height'].int()), int(ns['out_width'].int())])
        res['frames_' + tag] = np.stack([cases.box_down(f.numpy().transpose(1, 2, 0), 4)
                                         for f in ns['stable_list']])
        res['iqr_' + tag] = np.stack([cases.box_iqr(f.numpy().transpose(1, 2, 0), 4) for f in ns['stable_list']])
        if fm == 'AVERAGE':
            res['mesh1'] = ns['warp12_mesh1']
            res['middle'] = ns['middle_mesh']
            res['mesh3'] = ns['warp23_mesh2']
    save('g10_threeview', **res)


# ------------------------------------------------------------------ G12 three-view, full path (nets -> compose -> render)
def three_view_body():
    """Lines 345..505 of test() in test_online_tra_threeview.py, read at generation time (never stored)."""
    src_lines = open(os.path.join(ref_shim.REF, 'test_online_tra_threeview.py')).read().split('\n')
    return '\n'.join(l[4:] if l.startswith('    ') else l for l in src_lines[344:505])


def g12(nets):
    """test_online_tra_threeview.py:154-505 end to end: two 2-view passes (v1,v2), (v2,v3) through the reference's
    networks (the frame loop of :225-343 replayed by run_motion_stages with the reference's own functions), then the
    composition / render block executed verbatim.  8-frame 3-view clip, HR 180x320 (LR inputs are always 360x480)."""
    import test_online_tra_threeview as R3
    n = 8
    hr, lr = synth.make_clip(n, 180, 320, seed=4, views=3)
    a12 = run_motion_stages(nets, lr[0], lr[1])['acc']
    a23 = run_motion_stages(nets, lr[1], lr[2])['acc']
    res = dict(w12_m1=a12['smooth_mesh1'], w12_m2=a12['smooth_mesh2'], w23_m1=a23['smooth_mesh1'],
               w23_m2=a23['smooth_mesh2'])
    body = three_view_body()
    for wm, fm in (('NORMAL', 'AVERAGE'), ('NORMAL', 'LINEAR')):
        ns = dict(vars(R3))
        ns.update(warp12_mesh1=a12['smooth_mesh1'].clone(), warp12_mesh2=a12['smooth_mesh2'].clone(),
                  warp23_mesh1=a23['smooth_mesh1'].clone(), warp23_mesh2=a23['smooth_mesh2'].clone(),
                  img1_list=hr[0], img2_list=hr[1], img3_list=hr[2],
                  args=types.SimpleNamespace(warp_mode=wm, fusion_mode=fm))
        if fm == 'LINEAR':
            # record what the reference's blender sees and decides, per call (two chained calls per frame, threeview:498-501):
            # the warped masks, the nonzero counts and centroids its weights hang on, and its mask1
            calls = []
            ref_blender = ns['linear_blender']

            def spy(ref, tgt, ref_m, tgt_m, mask=False, _f=ref_blender, _calls=calls):
                r1, c1 = torch.nonzero(ref_m[0, 0], as_tuple=True)
                r2, c2 = torch.nonzero(tgt_m[0, 0], as_tuple=True)
                _calls.append(dict(count=np.array([r1.numel(), r2.numel()]),
                                   center=np.array([float(r1.float().mean()), float(c1.float().mean()),
                                                    float(r2.float().mean()), float(c2.float().mean())]),
                                   ref_m=cases.box_down(ref_m[0, 0].numpy()[..., None], 4)[..., 0],
                                   tgt_m=cases.box_down(tgt_m[0, 0].numpy()[..., None], 4)[..., 0],
                                   mask1=cases.box_down(_f(ref, tgt, ref_m, tgt_m, True)[0, 0].numpy()[..., None], 4)[..., 0]))
                return _f(ref, tgt, ref_m, tgt_m, mask)
            ns['linear_blender'] = spy
        exec(compile(body, 'threeview_345_505', 'exec'), ns)
        tag = fm.lower()
        if fm == 'LINEAR':
            assert len(calls) == 2 * n
            for key in ('count', 'center', 'ref_m', 'tgt_m', 'mask1'):
                res['lin_' + key] = np.stack([np.stack([calls[2 * i][key], calls[2 * i + 1][key]]) for i in range(n)])
        res['canvas_' + tag] = np.array([int(ns['out_height'].int()), int(ns['out_width'].int())])
        res['frames_' + tag] = np.stack([cases.box_down(f.numpy().transpose(1, 2, 0), 4) for f in ns['stable_list']])
        res['iqr_' + tag] = np.stack([cases.box_iqr(f.numpy().transpose(1, 2, 0), 4) for f in ns['stable_list']])
        if fm == 'AVERAGE':
            res['mesh1'] = ns['warp12_mesh1']
            res['middle'] = ns['middle_mesh']
            res['mesh3'] = ns['warp23_mesh2']
    save('g12_threeview_full', **res)


# ------------------------------------------------------------------ G11 PSNR / SSIM
def g13(nets):
    """uint8 in, uint8 out, as the reference's script runs on decoded video frames (test_online_tra.py:252-278, 409-417):
    the G9 clip quantised to uint8 (what cv2.imread hands over), HR = those bytes as fp32, LR = cv2.resize(frame,
    (480, 360)) / 127.5 - 1 -- at 360x480 the resize is the identity, so no resize arithmetic enters -- through the
    reference's networks and get_stable_sqe (NORMAL / AVERAGE), then `stable_list[k].astype(np.uint8)` (:413).
    Also the fp32 values of frame 0 on canvas columns where ONE view lies outside its image (tiles the product's
    footprint skipping drops): the reference's clamped-sampler residue fused with the other view's content."""
    hr, _ = synth.make_clip(16, 360, 480, seed=0)
    hr = [[f.round().clamp(0, 255) for f in v] for v in hr]               # uint8-valued fp32 frames
    lr = [[f / 127.5 - 1.0 for f in v] for v in hr]
    acc = run_motion_stages(nets, lr[0], lr[1])['acc']
    frames, ow, oh = RP.get_stable_sqe(hr[0], hr[1], acc['smooth_mesh1'], acc['smooth_mesh2'], 'NORMAL', 'AVERAGE')
    save('g13_frames_u8', canvas=np.array([int(oh), int(ow)]), smooth_mesh1=acc['smooth_mesh1'], smooth_mesh2=acc['smooth_mesh2'],
         frames_u8=np.stack([frames[0].astype(np.uint8), frames[-1].astype(np.uint8)]), frame_idx=np.array([0, len(frames) - 1]),
         left_f32=frames[0][:, 96:160].copy(), right_f32=frames[0][:, 544:608].copy())


def g14():
    """G8 + G9 again under the harsh checkpoint (`synth.synthetic_state_dict(profile='trained_like')`: BN-folded channel scales over
    four decades, Student-t taps, per-channel log-normal gains) on a 24-frame clip -- long enough that the product's launch-size rule
    picks F(4x4,3x3) for layer1 AND layer2 by itself (48 images x 12 tile blocks x 2 cout blocks = 576 >= 512 workgroups).
    Pins: the networks' raw outputs, the motions, the smoothed meshes, the NORMAL / AVERAGE frames and the alignment PSNR / SSIM."""
    nets = ref_nets('trained_like')
    sp = nets[0]
    n = 24
    hr, lr = synth.make_clip(n, 360, 480, seed=5)
    o1, o2r, o2t = sp(lr[0][0], lr[1][0])
    st = run_motion_stages(nets, lr[0], lr[1])
    acc = st['acc']
    res = dict(offset_1=o1, offset_2_ref=o2r, offset_2_tgt=o2t,
               motion1=torch.cat(st['s1'], 0), motion2=torch.cat(st['s2'], 0),
               tmotion1=torch.cat(st['t1'], 0), tmotion2=torch.cat(st['t2'], 0),
               tsmotion1=torch.cat(st['ts1'], 0), tsmotion2=torch.cat(st['ts2'], 0),
               smooth_mesh1=acc['smooth_mesh1'], smooth_mesh2=acc['smooth_mesh2'], ori_mesh2=acc['ori_mesh2'],
               ori_path2=acc['ori_path2'], smooth_path2=acc['smooth_path2'],
               **{'w0_' + k: v for k, v in st['first'].items()})
    frames, ow, oh = RP.get_stable_sqe(hr[0], hr[1], acc['smooth_mesh1'], acc['smooth_mesh2'], 'NORMAL', 'AVERAGE')
    res['canvas_normal_average'] = np.array([int(oh), int(ow)])
    res['frames_normal_average'] = np.stack([cases.box_down(f, 16) for f in frames])
    res['iqr_normal_average'] = np.stack([cases.box_iqr(f, 16) for f in frames])
    l1, l2 = RMET.get_stable_sqe(lr[0], lr[1], acc['smooth_mesh1'], acc['smooth_mesh2'])
    ps = skimage_metrics([(a[..., 0:3] * (a[..., 3:6] * b[..., 3:6]), b[..., 0:3] * (a[..., 3:6] * b[..., 3:6]))
                          for a, b in zip(l1, l2)])
    res['psnr'] = np.array([p for p, _ in ps])
    res['ssim'] = np.array([s for _, s in ps])
    save('g14_trained_like', **res)


def g11():
    a, b = cases.g11_images()
    (p, s), = skimage_metrics([(a, b)])
    save('g11_metrics', psnr=np.array(p), ssim=np.array(s))


if __name__ == '__main__':
    only = set(sys.argv[1:])            # e.g. `make_goldens.py g3 g12` regenerates just those

    def want(tag):
        return not only or tag in only
    nets = ref_nets()
    if want('g1'):
        g1()
    if want('g2'):
        g2()
    if want('g3'):
        g3()
    if want('g4'):
        g4(nets[0])
    if want('g5'):
        g5()
    if want('g6') or want('g7'):
        g6_g7()
    if want('g8') or want('g9'):
        g8_g9(nets)
    if want('g13'):
        g13(nets)
    if want('g10'):
        g10()
    if want('g11'):
        g11()
    if want('g12'):
        g12(nets)
    if want('g14'):
        g14()
