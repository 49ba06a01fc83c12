"""Key manifest of the golden fixtures: the ONE list both sides are held to.

`make_goldens.save()` refuses to write a fixture whose keys differ from this table, and `test_fixture_keys_match_manifest`
(CPU suite) refuses a committed .npz whose keys differ from it -- so extending a generator function forces an edit here, and an
edit here forces the fixture to be regenerated: a fixture can no longer drift from its recipe (round 4: g10 had).
"""

KEYS = {
    'g1_dlt': [
        'H_feat', 'H_full', 'H_ref_feat', 'H_ref_full', 'H_tgt_feat', 'H_tgt_full', 'mesh_ref', 'mesh_tgt', 'norm_rigid',
        'rigid',
    ],
    'g2_homo': [
        'out', 'out_small',
    ],
    'g3_costvol': [
        'cv3', 'cv3n', 'cv5', 'cv5n', 'full3_chsum', 'full3_rows', 'full5_chsum', 'full5_rows',
    ],
    'g4_ccl': [
        'flow', 'flow_full',
    ],
    'g5_tps_points': [
        'p_a', 'p_b',
    ],
    'g6_tps_warp': [
        'fast', 'ident_fast', 'ident_normal', 'normal',
    ],
    'g7_fusion': [
        'average', 'linear', 'mask1', 'warped_with_mask',
    ],
    'g8_nets': [
        'motion1', 'motion2', 'offset_1', 'offset_2_ref', 'offset_2_tgt', 'tmotion1', 'tmotion2', 'tsmotion1', 'tsmotion2',
        'w0_ori_mesh1', 'w0_ori_mesh2', 'w0_ori_path1', 'w0_ori_path2', 'w0_smooth_mesh1', 'w0_smooth_mesh2',
        'w0_smooth_path1', 'w0_smooth_path2',
    ],
    'g9_pipeline': [
        'canvas_fast_average', 'canvas_normal_average', 'canvas_normal_linear', 'distortion', 'frame0_crop',
        'frames_fast_average', 'frames_normal_average', 'frames_normal_linear', 'iqr_fast_average', 'iqr_normal_average',
        'iqr_normal_linear', 'lr_warp1_frame3', 'ori_mesh2', 'ori_path2', 'psnr', 'smooth_mesh1', 'smooth_mesh2',
        'smooth_path2', 'ssim', 'stability',
    ],
    'g10_threeview': [
        'canvas_average', 'canvas_linear', 'frames_average', 'frames_linear', 'iqr_average', 'iqr_linear', 'lin_center',
        'lin_count', 'lin_mask1', 'lin_ref_m', 'lin_tgt_m', 'mesh1', 'mesh3', 'middle',
    ],
    'g11_metrics': [
        'psnr', 'ssim',
    ],
    'g12_threeview_full': [
        'canvas_average', 'canvas_linear', 'frames_average', 'frames_linear', 'iqr_average', 'iqr_linear', 'lin_center',
        'lin_count', 'lin_mask1', 'lin_ref_m', 'lin_tgt_m', 'mesh1', 'mesh3', 'middle', 'w12_m1', 'w12_m2', 'w23_m1', 'w23_m2',
    ],
    'g13_frames_u8': [
        'canvas', 'frame_idx', 'frames_u8', 'left_f32', 'right_f32', 'smooth_mesh1', 'smooth_mesh2',
    ],
    'g14_trained_like': [
        'canvas_normal_average', 'frames_normal_average', 'iqr_normal_average', 'motion1', 'motion2', 'offset_1',
        'offset_2_ref', 'offset_2_tgt', 'ori_mesh2', 'ori_path2', 'psnr', 'smooth_mesh1', 'smooth_mesh2', 'smooth_path2',
        'ssim', 'tmotion1', 'tmotion2', 'tsmotion1', 'tsmotion2', 'w0_ori_mesh1', 'w0_ori_mesh2', 'w0_ori_path1',
        'w0_ori_path2', 'w0_smooth_mesh1', 'w0_smooth_mesh2', 'w0_smooth_path1', 'w0_smooth_path2',
    ],
}
