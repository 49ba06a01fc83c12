/* libstabstitch_hip.so -- C ABI of the MI355X (gfx950) StabStitch++ inference hot path.
 *
 * The reference (nie-lang/StabStitch2) has no FFI: its hot path is a chain of stock ATen ops
 * behind a Python module API.  The Python modules of `stabstitch2_amd/` keep that API
 *  and bind the entry points below through ctypes; each entry point names the
 * reference code it replaces (paths relative to Full_model_inference/Codes of the reference).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 data unless noted; the caller owns all buffers
 *     (the library never allocates, never synchronises, never touches the default stream);
 *   - `stream` is a hipStream_t passed as void*; every call is asynchronous on it;
 *   - return value: SS_OK or a negative SS_ERR_* code (no exceptions cross the boundary);
 *   - "nhwc" tensors are [N][T][H][W][C] with C fastest (T = 1 for 2-D); channel counts of conv
 *     inputs must be multiples of 4 (pad with zero channels; weights carry matching zero taps);
 *   - meshes are [.., 7, 9, 2] = 63 control points, (x, y) interleaved, as in the reference.
 */
#ifndef STABSTITCH_HIP_H
#define STABSTITCH_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* the library is built with -fvisibility=hidden: exactly the entry points declared here are exported */
#if defined(__GNUC__) || defined(__clang__)
#define SS_API __attribute__((visibility("default")))
#else
#define SS_API
#endif

#define SS_OK 0
#define SS_ERR_ARG (-1)          /* bad dimension / null pointer / unsupported combination */
#define SS_ERR_LAUNCH (-2)       /* hipGetLastError() != hipSuccess after the launch */
#define SS_ERR_UNSUPPORTED (-3)  /* this launch's sizes are outside what the kernel addresses; nothing was launched */
#define SS_ERR_DEVICE (-4)       /* the DEVICE cannot run the kernel (e.g. no 144 KB of LDS per workgroup); nothing was launched */

#define SS_WARP_NORMAL 0         /* the reference's clamped-index bilinear (utils/torch_tps_transform.py:30-106) */
#define SS_WARP_FAST 1           /* F.grid_sample(bilinear, zeros, align_corners=True) (:158-162) */
#define SS_WARP_EPS_FOLD 16      /* OR-ed into `mode` of the fused AVERAGE renders (ss_render_average*): the radial term as
                                  * a log(a), a = dx^2 + (dy^2 + 1e-6), instead of the reference's d2 log(d2 + 1e-6)
                                  * (utils/torch_tps_transform.py:108-137): ~1e-3 px at 720p, ~7 % of the kernel.  Opt-in. */

SS_API int ss_version(void);
SS_API const char* ss_error_string(int code);

/* ---- layout plumbing (replaces the implicit NCHW tensors of the reference modules) ---------- */
/* [n][c][h][w] -> [n][h][w][c_pad], channels c..c_pad-1 written as 0 */
SS_API int ss_nchw_to_nhwc(const float* in, float* out, int n, int c, int h, int w, int c_pad, void* stream);
/* [n][h][w][c_stride] (first c channels) -> [n][c][h][w] */
SS_API int ss_nhwc_to_nchw(const float* in, float* out, int n, int c, int h, int w, int c_stride, void* stream);

/* ---- K1/K2/K4/K11: convolution as fp32-MFMA implicit GEMM ------------------------------------
 * Replaces nn.Conv2d(+BatchNorm2d eval)(+residual)(+ReLU) of the ResNet-18 trunk
 * (spatial_network.py:123-139), the regressor convs (spatial_network.py:147-209,
 * temporal_network.py:65-93) and nn.Conv3d of SmoothNet (smooth_network.py:123-131).
 *   in   [n][t][h][w][cin]            cin % 4 == 0
 *   wgt  [cout][kt][kh][kw][cin]      (BN folded by the caller)
 *   bias [cout] ([groups][cout] for grouped launches) or NULL, res (same shape as out) or NULL
 *   out  [n][to][ho][wo][out_cs]      first cout channels written (out_cs >= cout)
 * stride applies to h and w (temporal stride is 1); kernel extents <= 8 per axis.  `groups` > 1 runs
 * `groups` independent problems with element strides in_gs / w_gs / out_gs between them (used by the
 * CCL Gram).  ws / ws_floats: caller workspace for the split-K partial sums of small problems;
 * ss_conv_workspace_need(...) returns the floats THIS launch wants (0 for most: no split); a NULL or
 * smaller workspace disables splitting (same result up to summation order, fewer workgroups). */
SS_API long long ss_conv_workspace_need(int n, int t, int h, int w, int cin, int cout, int kt, int kh, int kw,
                                 int stride, int pad_t, int pad_h, int pad_w, int groups);
SS_API int ss_conv_nhwc(const float* in, const float* wgt, const float* bias, const float* res, float* out,
                 int n, int t, int h, int w, int cin, int cout, int kt, int kh, int kw, int stride,
                 int pad_t, int pad_h, int pad_w, int relu, int out_cs,
                 int groups, long long in_gs, long long w_gs, long long out_gs,
                 float* ws, long long ws_floats, void* stream);
/* Conv2d + bias + ReLU + MaxPool2d(2, 2) in the launches that run split-K (ss_conv_workspace_need(...) > 0, ws of that size):
 * the pool rides in the split-K reduction (the regressors' conv, ReLU, MaxPool2d(2, 2) on small maps, spatial_network.py:147-259).
 * out [groups][n][Ho/2][Wo/2][out_cs].  Launches that would not split return SS_ERR_UNSUPPORTED and launch nothing (the caller
 * runs ss_conv_nhwc + ss_maxpool_nhwc).  Bit-identical to those two. */
SS_API int ss_conv_pool2_nhwc(const float* in, const float* wgt, const float* bias, float* out, int n, int h, int w, int cin,
                       int cout, int kh, int kw, int stride, int pad_h, int pad_w, int relu, int out_cs, int groups,
                       long long in_gs, long long w_gs, long long out_gs, float* ws, long long ws_floats, void* stream);

/* ---- the network stem: nn.Conv2d(3, 64, 7, stride 2, pad 3)(+BN)(+ReLU) of the ResNet-18 trunk (spatial_network.py:127-129,
 * temporal_network.py:47-49) on a 3-channel layout whose filter ROWS are contiguous: K = 7 x 24 = 168 instead of
 * 49 x 4 = 196 of the 4-channel NHWC form (147 real products).
 *   ss_nchw_to_nhwc3_padded   frames [n][3][h][w] -> [n][h][w + 8][3], 3 zero pixels left / 5 right
 *   ss_conv_stem3             in_padded as above, wgt [cout][7][24] (wgt[co][dh][3 dw + c] = w[co][c][dh][dw], rest 0),
 *                             out [n][ho][wo][out_cs]; bias / relu / out_cs / groups as ss_conv_nhwc (in_gs = 0 shares the
 *                             input between groups) */
SS_API int ss_nchw_to_nhwc3_padded(const float* in, float* out, int n, int h, int w, void* stream);
SS_API int ss_conv_stem3(const float* in_padded, const float* wgt, const float* bias, float* out, int n, int h, int w,
                         int cout, int relu, int out_cs, int groups, long long in_gs, long long w_gs, long long out_gs,
                         void* stream);
/* the whole stem in ONE kernel: Conv2d(3, 64, 7, 2, 3) + folded BN + ReLU + MaxPool2d(3, 2, 1) (spatial_network.py:127-130,
 * temporal_network.py:47-50) for `groups` 64-filter banks reading the same frames; the un-pooled 64-channel map is never
 * written (csrc/stem.hip).
 *   ss_stem_pool_pack    wgt [groups][64][7][24] (the ss_conv_stem3 layout) -> packed, ss_stem_pool_packed_floats(groups) floats
 *   ss_stem_pool         in_padded [n][h][w + 8][3] (ss_nchw_to_nhwc3_padded); bias [groups][64] or NULL;
 *                        out [groups][n][hp][wp][64], group stride out_gs floats, hp = ((h-1)/2)/2 + 1, wp likewise */
SS_API long long ss_stem_pool_packed_floats(int groups);
SS_API int ss_stem_pool_pack(const float* wgt, float* packed, int groups, void* stream);
SS_API int ss_stem_pool(const float* in_padded, const float* packed, const float* bias, float* out, int n, int h, int w,
                 int groups, long long out_gs, void* stream);

/* ---- the same convolution for 3x3 / stride 1 / pad 1 layers as fused Winograd F(2x2,3x3) on the fp32 matrix cores
 * (2.25x fewer MFMA flops; input and output transforms inside the GEMM kernel, nothing extra through HBM).  Replaces
 * the stride-1 3x3 nn.Conv2d(+BN)(+residual)(+ReLU) of the trunk bodies and regressors (spatial_network.py:132-136,
 * 147-209, temporal_network.py:65-93).  Filters are transformed (G g G^T in fp64, rounded to fp32) and laid out in
 * the kernel's B-operand register order ONCE per checkpoint load:
 *   ss_wino_packed_floats(cout, cin)         floats of the packed buffer (per group); cout % 16 == 0
 *   ss_wino_pack(wgt, packed, ...)           wgt [groups][cout][1][3][3][cin]  ->  packed [groups][...]
 *   ss_conv3x3_wino_nhwc(...)                in [n][h][w][cin], out [n][h][w][out_cs]; cin % 4 == 0, cout % 64 == 0;
 *                                            bias / res / relu / out_cs / groups as ss_conv_nhwc (u_gs = packed floats per group)
 *   ss_conv_uses_winograd(...)               the engine's own dispatch rule for a layer geometry (1 = Winograd pays:
 *                                            3x3 s1, cin >= 32, cout % 64 == 0, >= 70 % of the tile slots used, >= 96
 *                                            workgroups); callers may apply any rule, results agree to fp32 rounding */
SS_API long long ss_wino_packed_floats(int cout, int cin);
/* Opt-in variant of the same convolution: every fp32 x fp32 product is formed EXACTLY from three bf16 slices per operand
 * (all nine slice products) on the bf16 matrix pipe and accumulated in fp32 -- results agree with ss_conv3x3_wino_nhwc to
 * fp32 rounding (different accumulation order), not bit for bit.  ss_wino_pack3 slices the transformed filters once
 * (1.5x the bytes of ss_wino_pack); same argument meaning as the fp32 entry points. */
SS_API long long ss_wino_packed3_floats(int cout, int cin);
SS_API int ss_wino_pack3(const float* wgt, float* packed3, int cout, int cin, int groups, void* stream);
SS_API int ss_conv3x3_wino3_nhwc(const float* in, const float* packed3, const float* bias, const float* res, float* out,
                          int n, int h, int w, int cin, int cout, int relu, int out_cs, int groups,
                          long long in_gs, long long u_gs, long long out_gs, void* stream);
SS_API int ss_wino_pack(const float* wgt, float* packed, int cout, int cin, int groups, void* stream);
SS_API int ss_conv_uses_winograd(int kt, int kh, int kw, int stride, int cin, int cout, int ho, int wo, int images);
SS_API int ss_conv3x3_wino_nhwc(const float* in, const float* packed, const float* bias, const float* res, float* out,
                                int n, int h, int w, int cin, int cout, int relu, int out_cs, int groups,
                                long long in_gs, long long u_gs, long long out_gs, void* stream);

/* The same convolution followed by the regressors' MaxPool2d(2, 2) (spatial_network.py / temporal_network.py: conv3x3, ReLU,
 * conv3x3, ReLU, MaxPool2d) in ONE kernel: an F(2x2,3x3) output tile is a pooling window, the epilogue takes the maximum of its
 * four pixels before bias and ReLU (both monotone: bit-identical to ss_conv3x3_wino_nhwc + ss_maxpool_nhwc), the un-pooled map
 * is never written.  out [groups][n][h/2][w/2][out_cs] (floor, like MaxPool2d); out_gs = its group stride; no residual. */
SS_API int ss_conv3x3_wino_pool2_nhwc(const float* in, const float* packed, const float* bias, float* out, int n, int h, int w,
                               int cin, int cout, int relu, int out_cs, int groups, long long in_gs, long long u_gs,
                               long long out_gs, void* stream);

/* The same stride-1 3x3 convolution as fused Winograd F(4x4,3x3): 36 instead of 64 (F(2x2,3x3)) or 144 (direct) products per
 * 4x4 outputs and (cin, cout) pair -- 1.78x fewer MFMA flops than ss_conv3x3_wino_nhwc for a costlier input transform, run once
 * per workgroup (csrc/wino43.hip).  Same layers of the reference (the ResNet-18 bodies: spatial_network.py:132-136,
 * temporal_network.py:65-93); results agree with the other two forms to fp32 rounding of the larger transform constants
 * (per layer ~2e-5 relative; end to end inside the gates of tests/: tools/sim_wino43.py).  cin % 16 == 0, cout % 64 == 0, else
 * SS_ERR_UNSUPPORTED.  ss_wino43_pack: wgt [groups][cout][1][3][3][cin] -> packed [groups][ss_wino43_packed_floats].  All other
 * arguments as ss_conv3x3_wino_nhwc. */
SS_API long long ss_wino43_packed_floats(int cout, int cin);
/* The engine's dispatch rule for the F(4x4,3x3) kernel, for callers that want the library's choice (the Python host applies it:
 * ops._uses_wino43; the counterpart of ss_conv_uses_winograd): 1 when the geometry is one the kernel takes (1x3x3, stride 1,
 * cin % 16 == 0, cout % 64 == 0, 32-bit buffer offsets) AND the launch pays -- one 8 x 60-pixel x 64-channel tile per workgroup,
 * one workgroup per CU: >= min_wgs workgroups (<= 0: 512, two rounds of the chip), >= min_fill_pct % of the tile slots on real
 * pixels (<= 0: 85 for the 8 x 60 geometry, 60 for the 16 x 32 one that serves maps <= 31 columns wide), cin >= min_cin (<= 0: 64); all three at 1 = wherever the kernel runs at all.  images = images per group,
 * groups = launch groups.  Results of the three 3x3 kernels agree to fp32
 * rounding, so the kernel choice (hence the launch size) shows in the last digits: pin it with min_wgs for reproducible runs. */
SS_API int ss_conv_uses_wino43(int kt, int kh, int kw, int stride, int cin, int cout, int ho, int wo, int images, int groups,
                               int min_wgs, int min_cin, int min_fill_pct);
SS_API int ss_wino43_pack(const float* wgt, float* packed, int cout, int cin, int groups, void* stream);
SS_API int ss_conv3x3_wino43_nhwc(const float* in, const float* packed, const float* bias, const float* res, float* out,
                                  int n, int h, int w, int cin, int cout, int relu, int out_cs, int groups,
                                  long long in_gs, long long u_gs, long long out_gs, void* stream);
/* Process-wide A/B knob of ss_conv3x3_wino43_nhwc (output-neutral: bit-identical results either way): 1 (default) = PERSISTENT
 * workgroups, one per CU, each walking its share of the launch's tile blocks and requesting the next block's rows in front of
 * the current block's epilogue; 0 = the same kernel launched with one workgroup per tile block (the schedule of rounds 4-5). */
SS_API int ss_wino43_set_persistent(int on);

/* nn.MaxPool2d(k, stride, pad) on nhwc (floor mode; spatial_network.py:130,152; -inf padding) */
SS_API int ss_maxpool_nhwc(const float* in, float* out, int n, int h, int w, int c, int k, int stride, int pad,
                    void* stream);
/* same pooling, channels [0,c/2) -> out0 and [c/2,c) -> out1 (both [n][ho][wo][c/2]; c % 8 == 0): lets the SpatialNet and
 * TemporalNet stems (identical 7x7 s2 conv + pool on the same frames) share one conv1 launch with 2x64 filters */
SS_API int ss_maxpool_nhwc_split(const float* in, float* out0, float* out1, int n, int h, int w, int c, int k, int stride,
                          int pad, void* stream);

/* K5: nn.Linear (+ReLU): y[m][nout] = x[m][k] . w[nout][k] + b  (spatial_network.py:170-178, 211-219) */
SS_API int ss_linear(const float* x, const float* w, const float* b, float* y, int m, int k, int nout, int relu,
              void* stream);
/* regressNet2_part2_ref / _tgt (spatial_network.py:233-259) and TemporalNet's regressNet2_part2 on two views
 * (temporal_network.py:87-105) share their launches: `groups` <= 8 fully connected layers of identical shape in one launch: x [groups][m][k] (group stride x_group_stride floats),
 * w [groups][nout][k], b [groups][nout] or NULL; group g's [m][nout] result goes to y_groups[g] (HOST array of device pointers:
 * each regressor head's output lands where its consumer reads it).  k % 4 == 0.  Row results equal ss_linear's bit for bit. */
SS_API int ss_linear_grouped(const float* x, long long x_group_stride, const float* w, const float* b,
                      float* const* y_groups, int groups, int m, int k, int nout, int relu, void* stream);

/* ---- K3: contextual correlation layer (spatial_network.py:369-425) ---------------------------
 * f1, f2 nhwc [n][h][w][c]; flow out NCHW [n][2][h][w] (ch0 = dx, ch1 = dy).
 * ws: caller workspace of ss_ccl_workspace_floats(n,h,w,c) floats. */
SS_API long long ss_ccl_workspace_floats(int n, int h, int w, int c);
SS_API int ss_ccl(const float* f1, const float* f2, float* flow_nchw, float* flow_nhwc4, int n, int h, int w, int c,
           float softmax_scale, float* ws, void* stream);

/* F.normalize(x, p=2, dim=channels) on nhwc: out[p][:] = in[p][:] / max(||in[p][:]||_2, 1e-12)
 * (first step of CCL, spatial_network.py:372-373, and of cost_volume(norm=True), spatial_network.py:335-337) */
SS_API int ss_l2norm_nhwc(const float* in, float* out, long long n_pixels, int c, void* stream);

/* ---- K8: cost volume (spatial_network.py:333-358, temporal_network.py:149-174) ---------------
 * x1, x2 nhwc [n][h][w][c]; out nhwc [n][h][w][out_cs], channel j*(2r+1)+i, channels >= (2r+1)^2
 * written as 0.  out[.,y,x,d] = leaky_relu_0.1(mean_c x1[y,x,c] * x2[y+j-r, x+i-r, c]). */
SS_API int ss_cost_volume(const float* x1, const float* x2, float* out, int n, int h, int w, int c, int r,
                   int out_cs, void* stream);
/* both directions of SpatialNet's stage 2 (spatial_network.py:318, 325) in ONE launch: out [2][n][h][w][out_cs] =
 * cost_volume(x1, x2), cost_volume(x2, x1) */
SS_API int ss_cost_volume_bidir(const float* x1, const float* x2, float* out, int n, int h, int w, int c, int r,
                         int out_cs, void* stream);
/* n volumes over inputs that hold every image ONCE: volume b = ss_cost_volume of image b + (b >= split ? shift : 0) of x1 and x2.
 * A chain of S pairs (view s, view s + 1) stores its S + 1 views' features once and gets the 2 S volumes [first views | second
 * views] with split = S, shift = 1 - S (temporal_network.py:120-147 per view). */
SS_API int ss_cost_volume_shifted(const float* x1, const float* x2, float* out, int n, int h, int w, int c, int r, int out_cs,
                                  int split, int shift, void* stream);
/* tile height of the cost-volume kernel (spatial_network.py:333-358; process-wide A/B knob): 0 = the library's choice, 4 or 8 output rows x 16 columns per
 * workgroup; identical results. */
SS_API int ss_cost_volume_set_tile(int ty);

/* ---- K6: 4-point DLT, bidirectional decomposition, H -> mesh (fp64 on device) -----------------
 * ss_tensor_dlt: utils/torch_DLT.py:17-45; src, dst [n][4][2] -> H [n][3][3]. */
SS_API int ss_tensor_dlt(const float* src, const float* dst, float* H, int n, void* stream);
/* spatial_network.py:291-312: offset_1 [n][8] at image size (img_h, img_w), feature scale 8 ->
 * normalised homographies theta_ref = M^-1 H_ref M, theta_tgt = M^-1 H_tgt M, each [n][3][3]. */
SS_API int ss_spatial_decompose(const float* offset8, float* theta_ref, float* theta_tgt, int n, float img_h,
                         float img_w, void* stream);
/* spatial_network.py:63-118 (build_SpatialNet tail): offset_1 [n][8], offset_2_ref/tgt [n][126] ->
 * motion1, motion2 [n][7][9][2] (mesh - rigid). */
SS_API int ss_spatial_meshes(const float* offset8, const float* off_ref, const float* off_tgt, float* motion1,
                      float* motion2, int n, float img_h, float img_w, void* stream);

/* ---- K7: homography sampler (utils/torch_homo_transform.py:6-184) ----------------------------
 * theta [n][3][3]; nhwc variant for the 1/8 feature maps, nchw variant = the reference API. */
SS_API int ss_homo_warp_nhwc(const float* in, const float* theta, float* out, int n, int h, int w, int c,
                      int out_h, int out_w, void* stream);
/* (warp(in1, theta[0:n]), warp(in2, theta[n:2n])) -> out [2n][out_h][out_w][c] as ONE launch.  in1 and in2 are images of ONE
 * NHWC tensor a whole number of images apart and may overlap: a chain of pairs (view 1, view 2), (view 2, view 3) reads views
 * [0:n] and [1:n+1] of its trunk output (spatial_network.py:302-318 per pair).  SS_ERR_ARG when in2 - in1 is not that. */
SS_API int ss_homo_warp_pair_nhwc(const float* in1, const float* in2, const float* theta, float* out, int n, int h, int w, int c,
                                  int out_h, int out_w, void* stream);
SS_API int ss_homo_warp_nchw(const float* in, const float* theta, float* out, int n, int c, int h, int w,
                      int out_h, int out_w, void* stream);

/* ---- K9/K10: thin-plate spline (utils/torch_tps_transform.py:168-226,
 *      utils/torch_tps_transform_point.py:21-125) -------------------------------------------- */
/* source, target [n][63][2] (normalised) -> T [n][2][66]; fp32 kernel matrix, fp64 solve. */
SS_API int ss_tps_solve(const float* source, const float* target, float* T, int n, void* stream);
/* n control-point sets source [n][63][2] against ONE shared target [63][2] (the render's splines: every frame's warped mesh
 * maps onto the same rigid mesh, test_online_tra.py:129-137) -> T [n][2][66]; no broadcast copy of the target. */
SS_API int ss_tps_solve_shared_target(const float* source, const float* target, float* T, int n, void* stream);
/* point [n][q][2] evaluated through (source, T) -> out [n][q][2] */
SS_API int ss_tps_points(const float* point, const float* source, const float* T, float* out, int n, int q,
                  void* stream);
/* W^-1 (fp64, [66][66]) of the TPS system of ONE control-point set: `torch.inverse(W.double())` of
 * utils/torch_tps_transform_point.py:113.  For callers that solve from a constant source mesh (see ss_tsmotion). */
SS_API int ss_tps_inverse(const float* source, double* winv, void* stream);
/* test_online_tra.py:309-347 for one view, all frames at once: smotion, tmotion [n][63][2] (LR px)
 * -> smesh [n][63][2] = rigid + smotion, tsmotion [n][63][2] (frame 0 = 0).
 * Every system of this composition has the RIGID mesh as its source, so T = W^-1 [target; 0] with one constant W^-1:
 * rigid_winv = ss_tps_inverse(normalised rigid mesh of (img_h, img_w)) computed once by the caller, or NULL (then each
 * frame runs its own elimination).  ws: ss_tsmotion_workspace_floats(n) floats. */
SS_API long long ss_tsmotion_workspace_floats(int n);
SS_API int ss_tsmotion(const float* smotion, const float* tmotion, float* smesh, float* tsmotion, int n,
                       float img_h, float img_w, const double* rigid_winv, float* ws, void* stream);
/* the same with frame k pairing with frame k - lag (test_online_tra.py:320-340 for S streams interleaved as frame = time * S +
 * stream: lag = S; the first `lag` frames get tsmotion 0).  ss_tsmotion = lag 1. */
SS_API int ss_tsmotion_lag(const float* smotion, const float* tmotion, float* smesh, float* tsmotion, int n, int lag,
                    float img_h, float img_w, const double* rigid_winv, float* ws, void* stream);

/* ---- K12/K13: dense TPS warp and fusion (utils/torch_tps_transform.py:108-165,
 *      test_online_tra.py:34-58, 138-150) ----------------------------------------------------- */
/* generic: U [b][c][h][w] NCHW, source [b][63][2], T [b][2][66] -> out [b][c][hc][wc] */
SS_API int ss_tps_warp_nchw(const float* U, const float* source, const float* T, float* out, int b, int c, int h,
                     int w, int hc, int wc, int mode, void* stream);
/* same, plus one extra output channel = warp of an all-ones plane (the validity mask of
 * test_online_tra.py:144-147): out [b][c+1][hc][wc] */
SS_API int ss_tps_warp_mask_nchw(const float* U, const float* source, const float* T, float* out, int b, int c,
                          int h, int w, int hc, int wc, int mode, void* stream);
/* all views of one frame in one launch, per-view image pointers (host array of `views` <= 3 device pointers to
 * [3][h][w]); out [views][4][hc][wc] = 3 colour planes + ones-mask plane (test_online_tra.py:144-147) */
SS_API int ss_tps_warp_views(const float* const* imgs, const float* source, const float* T, float* out, int views,
                      int h, int w, int hc, int wc, int mode, void* stream);
/* fused render of one stitched frame, AVERAGE fusion, 2 or 3 views (chained (1+2)+3):
 * imgs: array of `views` device pointers (host array) to [3][h][w]; source [views][63][2];
 * T [views][2][66]; out [3][hc][wc].
 * footprint: NULL = every view is evaluated at every canvas pixel (the reference's arithmetic everywhere, including the
 * rounding residue its clamped sampler returns outside a view's image); or this frame's block of ss_render_footprints:
 * 64 x 8-pixel tiles outside a view's mesh hull skip its 63-term spline and take its contribution as exactly 0
 * (differs from the reference there by that residue, <~ 1e-2 grey levels of machine-dependent noise).
 * footprint_floats: size of that block; must equal ss_render_footprint_floats(views, hc, wc) (a block built for another
 * canvas or view count is SS_ERR_ARG, not an out-of-bounds read); ignored when footprint is NULL. */
SS_API int ss_render_average(const float* const* imgs, const float* source, const float* T, const float* footprint,
                             long long footprint_floats, float* out, int views, int h, int w, int hc, int wc, int mode,
                             void* stream);
/* the same render from decoded uint8 frames [h][w][3] (cv2.imread's layout, test_online_tra.py:252-258) straight to the
 * uint8 video frame [hc][wc][3] (`.astype(np.uint8)` of the fused values, :413): bit-identical to ss_ingest_u8 ->
 * ss_render_average -> ss_canvas_to_u8 without the fp32 frame planes and the fp32 canvas ever being written */
SS_API int ss_render_average_u8(const unsigned char* const* frames, const float* source, const float* T,
                         const float* footprint, long long footprint_floats, unsigned char* out, int views, int h, int w,
                         int hc, int wc, int mode, void* stream);
/* a whole clip in ONE launch (the frame loop of get_stable_sqe, test_online_tra.py:127-152): views_base = host array of
 * `views` device pointers to [frames][3][h][w] fp32; source [frames][views][63][2]; T [frames][views][2][66]; footprint
 * [frames][footprint_floats] or NULL; out [frames][3][hc][wc].  Bit-identical to `frames` calls of ss_render_average. */
SS_API int ss_render_average_clip(const float* const* views_base, const float* source, const float* T,
                           const float* footprint, long long footprint_floats, float* out, int frames, int views, int h,
                           int w, int hc, int wc, int mode, void* stream);
/* the same from decoded uint8 clips [frames][h][w][3] per view to uint8 video frames [frames][hc][wc][3] */
SS_API int ss_render_average_clip_u8(const unsigned char* const* views_base, const float* source, const float* T,
                              const float* footprint, long long footprint_floats, unsigned char* out, int frames,
                              int views, int h, int w, int hc, int wc, int mode, void* stream);
/* footprints of frames x views splines (source [frames][views][63][2], T [frames][views][2][66]) on an hc x wc canvas, one
 * launch: per frame ss_render_footprint_floats(views, hc, wc) floats = exactly evaluated sampling coordinates on the
 * lattice of tile corners and long-edge midpoints, the bounding box of each view's control points (its mesh hull) and the
 * frame's tile order (four counters + four lists: the tiles by the number of views that reach them, most expensive class
 * first; the order inside a class is not specified); a tile is skipped for a
 * view when it lies outside the hull AND its four corners and two long-edge midpoints all sample more than 8 pixels beyond
 * the same side of the (h x w) image (csrc/render.hip; a test, not a proof -- pass footprint = NULL for the reference's
 * arithmetic at every pixel). */
SS_API long long ss_render_footprint_floats(int views, int hc, int wc);
SS_API int ss_render_footprints(const float* source, const float* T, float* fp, int frames, int views, int h, int w,
                                int hc, int wc, void* stream);
/* ss_render_footprints + the streaming canvas' overflow watcher inside the same launches: ss_canvas_watch's update of
 * watch_i / watch_f [frames][4] from `source` (frame = stream), for pushes that normalise their control points elsewhere
 * (ss_three_view_splines).  One graph node less on a batch-1 push. */
SS_API int ss_render_footprints_watch(const float* source, const float* T, float* fp, int frames, int views, int h, int w,
                                      int hc, int wc, float guard, int* watch_i, float* watch_f, void* stream);
/* LINEAR fusion (linear_blender): ref, tgt [3][hc][wc]; ref_m, tgt_m [hc][wc]; out [3][hc][wc];
 * mask1_out optional [hc][wc]; ws: ss_linear_blend_workspace_floats(hc, wc) floats. */
SS_API long long ss_linear_blend_workspace_floats(int hc, int wc);
SS_API int ss_linear_blend(const float* ref, const float* tgt, const float* ref_m, const float* tgt_m, float* out,
                    float* mask1_out, int hc, int wc, float* ws, void* stream);
/* LINEAR fusion of a whole clip -- the frame loop of get_stable_sqe with fusion_mode == 'LINEAR' (test_online_tra.py:127-152
 * around linear_blender :34-58; three views chained ((1 (+) 2) (+) 3) with mask12 = m1 + m2 - m1 m2 as
 * test_online_tra_threeview.py:489-502) in three launches (a second blend pass with three views): warp of every view with the
 * blender's statistics gathered on the way, one reduction per (frame, pass), one fused X / blur / blend kernel.
 * views_base = host array of `views` device pointers to [frames][3][h][w] fp32 (the _u8 form: decoded uint8 [frames][h][w][3]);
 * source [frames][views][63][2]; T [frames][views][2][66]; out [frames][3][hc][wc] fp32 (the _u8 form: the video frames
 * [frames][hc][wc][3] uint8, `.astype(np.uint8)` of the blend); mask1_out optional [frames][views-1][hc][wc] (mask1 of every
 * blend pass); ws: ss_linear_clip_workspace_floats(frames, views, hc, wc) floats.  Bit-identical, frame by frame, to
 * ss_tps_warp_views + ss_linear_blend (+ ss_mask_union + ss_linear_blend). */
SS_API long long ss_linear_clip_workspace_floats(int frames, int views, int hc, int wc);
/* form of the clip blend kernel (the blur + blend of linear_blender, test_online_tra.py:52-58; process-wide, for A/B runs and the equivalence test): rows = 0 the default (rolling pass,
 * strips of 64 columns x 96 rows per wave), > 0 that many rows per strip, < 0 the 64 x 64-tile kernel; identical output. */
SS_API int ss_linear_clip_set_rows(int rows);
SS_API int ss_render_linear_clip(const float* const* views_base, const float* source, const float* T, float* out,
                          float* mask1_out, int frames, int views, int h, int w, int hc, int wc, int mode, float* ws,
                          void* stream);
SS_API int ss_render_linear_clip_u8(const unsigned char* const* views_base, const float* source, const float* T,
                             unsigned char* out, float* mask1_out, int frames, int views, int h, int w, int hc, int wc,
                             int mode, float* ws, void* stream);

/* ---- K14: canvas bounding box and mesh normalisation (test_online_tra.py:103-136) ------------
 * mesh: n_points (x,y) pairs at LR scale (480x360); each is scaled to the HR frame as the reference
 * does (x*img_w/480, y*img_h/360) before the min/max; img_w <= 0 / img_h <= 0 means the mesh is
 * already in canvas pixels (three-view second canvas).  bbox (device) [4] = wmin, wmax, hmin, hmax;
 * accumulate != 0 folds the existing bbox contents in (second view, more clips). */
SS_API int ss_mesh_bbox(const float* mesh, int n_points, float img_h, float img_w, float* bbox, int accumulate,
                 void* stream);
/* out = norm(scale(mesh) - (wmin,hmin); Hc_f, Wc_f) with the float canvas size read from bbox on device */
SS_API int ss_mesh_normalize(const float* mesh, const float* bbox, float* out, int n_points, float img_h,
                      float img_w, void* stream);
/* H2Mesh (spatial_network.py:20-36): out = persp_divide(H^-1 [x y 1]^T) over mesh [n][n_points][2]; H [n][3][3]; the 3 x 3
 * inverse and the products in fp64 */
SS_API int ss_h2mesh(const float* H, const float* mesh, float* out, int n, int n_points, void* stream);
/* three-view mesh alignment (test_online_tra_threeview.py:345-420), meshes [frames][63][2] at LR scale:
 *   ss_three_view_align   scale the four meshes to HR, add the per-frame mean offset of (w12_m2 - w23_m1) to pair (2,3),
 *                         middle = (w12_m2 + w23_m1) / 2 -> a1, a2, b1, b2, mid in HR pixels (before the canvas translation)
 *   ss_three_view_finish  bbox = first canvas (ss_mesh_bbox over a1, a2, b1, b2): mesh1 / mesh3 = the re-projected outer
 *                         meshes n1 / n3 (normalised, from ss_tps_points) back in canvas pixels, middle = mid - (wmin, hmin) */
SS_API int ss_three_view_align(const float* w12_m1, const float* w12_m2, const float* w23_m1, const float* w23_m2,
                        float* a1, float* a2, float* b1, float* b2, float* mid, int frames, float img_h, float img_w,
                        void* stream);
/* the composition's five meshes (HR pixels, as ss_three_view_align writes them) normalised on the first canvas `bbox` in ONE launch
 * (= five ss_mesh_normalize calls with img_h = img_w = 0, bit for bit), laid out for one batched solve + one point evaluation of
 * both re-projections (threeview:381-420): out [6][n_points][2] = {a1, b2 | a2, b1 | mid, mid}: points = out[0:2], sources =
 * out[2:4], targets = out[4:6] */
SS_API int ss_three_view_normalize(const float* a1, const float* a2, const float* b1, const float* b2, const float* mid,
                            const float* bbox, float* out, long long n_points, void* stream);
SS_API int ss_three_view_finish(const float* n1, const float* n3, const float* mid, const float* bbox, float* mesh1,
                         float* middle, float* mesh3, long long n_points, void* stream);
/* The streaming three-view push between the pair chains' smoothed meshes and the render (test_online_tra_threeview.py:345-420 +
 * the splines of :421-505) as ONE launch of 3 workgroups per frame: ss_three_view_align -> ss_three_view_normalize ->
 * ss_tps_solve -> ss_tps_points -> ss_three_view_finish on the FIRST canvas `first_box`, then every view's final mesh normalised
 * on the OUTPUT canvas `out_box` (ss_stream_normalize_watch's arithmetic) and ss_tps_solve_shared_target onto `nrigid` [63][2].
 * Same device functions in the same order: bit-identical to those seven launches.  w*_m*: [frames][63][2] at LR scale, frame f
 * at + f * mesh_frame_stride floats; boxes: device (wmin, wmax, hmin, hmax); -> mesh1 / middle / mesh3 [frames][63][2] in
 * first-canvas pixels, src [frames][3][63][2], T [frames][3][2][66].  The watcher is NOT updated here: ss_render_footprints_watch
 * or ss_canvas_watch on `src`. */
SS_API int ss_three_view_splines(const float* w12_m1, const float* w12_m2, const float* w23_m1, const float* w23_m2,
                                 long long mesh_frame_stride, const float* first_box, const float* out_box, const float* nrigid,
                                 float* mesh1, float* middle, float* mesh3, float* src, float* T, int frames, float img_h,
                                 float img_w, void* stream);
/* A streaming push's render splines as ONE launch (views x streams workgroups): ss_stream_normalize_watch's normalisation of
 * every view's newest mesh on its stream's canvas (same arguments) + ss_tps_solve_shared_target onto `nrigid`; -> src
 * [streams][views][63][2], T [streams][views][2][66].  Bit-identical to the two launches; the watcher is NOT updated here
 * (ss_render_footprints_watch or ss_canvas_watch on `src`). */
SS_API int ss_stream_splines(const float* const* meshes, int views, long long mesh_frame_stride, const float* bboxes,
                             int bbox_frame_stride, const float* nrigid, float* src, float* T, int streams, float img_h,
                             float img_w, void* stream);
/* Streaming mode (stabstitch2_amd/online.py): the reference sizes the canvas from ALL frames of the clip (test_online_tra.py:
 * 106-120); a live stream fixes it after its first window, so a mesh that drifts past it later would be cropped silently.  This
 * launch (one wave per stream, capturable) looks at the push's control points src [streams][views][63][2], already normalised to
 * each stream's canvas ([-1, 1] = inside), and updates device-resident state that is read only when asked for:
 *   watch_i [streams][4] int32 = {frames seen, frames with a point outside the canvas, index of the first such frame (-1 = none),
 *                                 frames with a point within `guard` (normalised units) of an edge or outside}
 *   watch_f [streams][4] fp32  = running {xmin, xmax, ymin, ymax} of the normalised coordinates (what a grown canvas must cover)
 * Initialise to {0, 0, -1, 0} / {+inf, -inf, +inf, -inf}. */
SS_API int ss_canvas_watch(const float* src, int streams, int views, float guard, int* watch_i, float* watch_f, void* stream);
/* One streaming push's control points of ALL views + the watcher above in ONE launch (one wave per stream; a batch-1 push is
 * bound by its launch count): meshes[v] = view v's newest LR-scale meshes, stream s at meshes[v] + s * mesh_frame_stride floats;
 * bboxes [4] (bbox_frame_stride 0: one canvas) or [streams][4] (4: a canvas per stream); out [streams][views][63][2] = what `views`
 * calls of ss_mesh_normalize_views(_boxes) write, bit for bit; watch_i / watch_f as ss_canvas_watch (both NULL: no watcher).
 * A NaN control point counts as outside the canvas; a guard below the 2.5e-4 rounding slack makes `near` coincide with `outside`. */
SS_API int ss_stream_normalize_watch(const float* const* meshes, int views, long long mesh_frame_stride, const float* bboxes,
                              int bbox_frame_stride, float* out, int streams, float img_h, float img_w, float guard,
                              int* watch_i, float* watch_f, void* stream);
/* the same for view `view` of `views` of a clip, mesh [frames][63][2], written into the render's source layout
 * out [frames][views][63][2] (one call per view assembles it; test_online_tra.py:129-136) */
SS_API int ss_mesh_normalize_views(const float* mesh, const float* bbox, float* out, int frames, int view, int views,
                            float img_h, float img_w, void* stream);
/* the same (test_online_tra.py:129-136) with ONE CANVAS BOX PER FRAME (bboxes [frames][4]) and frame f's mesh at mesh + f * mesh_frame_stride floats: S live
 * streams, each with its own fixed canvas, normalised in one launch per view (batch-of-streams streaming mode) */
SS_API int ss_mesh_normalize_views_boxes(const float* mesh, long long mesh_frame_stride, const float* bboxes, float* out,
                                  int frames, int view, int views, float img_h, float img_w, void* stream);
/* p[0..n) = value (the zero motion of frame 0, temporal_network.py:31-33, written in place) */
SS_API int ss_fill_f32(float* p, float value, long long n, void* stream);

/* ---- K11: SmoothNet glue (smooth_network.py:64-157) ------------------------------------------
 * smesh1/2, tsmotion1/2 [frames][63][2] (LR px).  Window wi covers frames wi*wstride .. +t-1
 * (wstride = 1: the sliding window of test_online_tra.py:359-392 without materialising it;
 * wstride = t: plain batch of windows).  tsflow = running sum of tsmotion inside the window; with
 * zero_first != 0 the first tsmotion of every window counts as 0 (test_online_tra.py:362-366).
 * embed: hidden nhwc [nw][t][7][9][128] =
 *        relu(E1 smesh1) | relu(E3 tsflow1) | relu(E1 smesh2) | relu(E3 tsflow2);  e1w,e3w [32][2]; e1b,e3b [32]. */
SS_API int ss_smooth_embed(const float* smesh1, const float* smesh2, const float* ts1, const float* ts2,
                    const float* e1w, const float* e1b, const float* e3w, const float* e3b, float* hidden,
                    int nw, int t, int wstride, int zero_first, void* stream);
/* finalize: delta [nw][t][63][4] (decoder output) -> the 8 tensors of build_SmoothNet, each
 * [nw][t][63][2]; any output pointer may be NULL. */
SS_API int ss_smooth_finalize(const float* smesh1, const float* smesh2, const float* ts1, const float* ts2,
                       const float* delta, float* ori_mesh1, float* ori_mesh2, float* ori_path1,
                       float* ori_path2, float* smooth_mesh1, float* smooth_mesh2, float* smooth_path1,
                       float* smooth_path2, int nw, int t, int wstride, int zero_first, void* stream);
/* the clip's tensors straight from the sliding windows (wstride 1, first tsmotion of every window zeroed): window 0
 * contributes its t frames, every later window its last frame (test_online_tra.py:377-392); the metric harness's paths
 * are chained across windows sequentially as test_metric_ssd.py:427-436 does.  smesh*, ts* [n][63][2] with
 * n = nw + t - 1; delta [nw][t][63][4] -> ori_mesh1/2, smooth_mesh1/2 [n][63][2]; ori_path2, smooth_path2 [n][63][2]
 * (both or neither may be NULL). */
SS_API int ss_smooth_stitch(const float* smesh1, const float* smesh2, const float* ts1, const float* ts2,
                     const float* delta, float* ori_mesh1, float* ori_mesh2, float* smooth_mesh1, float* smooth_mesh2,
                     float* ori_path2, float* smooth_path2, int nw, int t, void* stream);

/* Streaming mode (one frame pair per call; the reference's per-frame loop, test_online_tra.py:284-392, keeps Python lists):
 * the sliding buffers have fixed addresses, a new frame shifts them.  ring [rings][window][elems]: every ring drops its
 * oldest slot and takes the `elems` floats at src + src_off[r] (src_off: HOST array of `rings` <= 8 element offsets) as its
 * newest; in the same launch `blocks` blocks of `block` floats are moved inside `state`: block b (at b * stride) <- the
 * floats `delta` (>= block) further.  (window - 1) * elems <= 2048. */
SS_API int ss_window_push(float* ring, const float* src, const long long* src_off, int rings, int window, int elems,
                   float* state, int blocks, int block, long long stride, long long delta, void* stream);
/* S streams advancing together (batch-of-streams streaming mode; per stream the window bookkeeping of test_online_tra.py:359-392): groups x per rings, ring g * per + j takes the row at
 * src + src_off[g] + j * elems (src_off: HOST array of `groups` <= 8 offsets, one per ring KIND; per = S).  With more than one
 * state block, stride >= delta + block (the blocks move without ordering between them). */
SS_API int ss_window_push_groups(float* ring, const float* src, const long long* src_off, int groups, int per, int window,
                          int elems, float* state, int blocks, int block, long long stride, long long delta, void* stream);

/* canvas-sized elementwise helpers of the harnesses: out = (in + add) * mul  ((img+1)*127.5,
 * test_metric_ssd.py:166);  out = a + b - a*b  (three-view mask union, test_online_tra_threeview.py:501) */
SS_API int ss_add_mul(const float* in, float* out, float add, float mul, long long n, void* stream);
SS_API int ss_mask_union(const float* a, const float* b, float* out, long long n, void* stream);

/* ---- frame I/O either side of the path ----------------------------------
 * ss_ingest_u8 replaces the per-frame host code of test_online_tra.py:252-278: `frames` is DEVICE uint8
 * [n][h][w][3] (decoded frames as cv2.imread returns them, channel order preserved);
 *   hr  [n][3][h][w]        = float(frame)                     (NULL to skip)
 *   lr  [n][3][lr_h][lr_w]  = cv2.resize(frame, (lr_w, lr_h)) / 127.5 - 1.0
 * with OpenCV 4.5.1's (environment.yml:343) uint8 INTER_LINEAR arithmetic restated bit for bit: 11-bit fixed-point
 * taps, ((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2)>>2, the exact-2x2 case routed to INTER_AREA ((a+b+c+d+2)>>2),
 * equal sizes copied.  Parity of this entry point is UNPINNED against cv2 itself (absent from the image); see
 * oracle/frame_io.py.
 * ss_canvas_to_u8 replaces `stable_list[k].astype(np.uint8)` (:413): [n][3][h][w] fp32 -> uint8 [n][h][w][3]
 * (truncation toward zero, low 8 bits of the int32 value outside 0..255). */
SS_API int ss_ingest_u8(const unsigned char* frames, float* hr, float* lr, int n, int h, int w, int lr_h, int lr_w,
                 void* stream);
SS_API int ss_canvas_to_u8(const float* canvas, unsigned char* out, int n, int h, int w, void* stream);

/* ---- metric harness (test_metric_ssd.py:444-482, 513-527) -------------------------------------
 * w1, w2: [frames][4][h][w] = 3 colour planes (0..255) + validity-mask plane, as ss_tps_warp_mask_nchw
 * writes them.  out (device, fp64) [frames][2] = alignment PSNR (dB), SSIM of (w1*ov, w2*ov), ov = m1*m2,
 * computed in fp64 with scikit-image 0.15 compare_psnr / compare_ssim semantics (data_range 255, 7x7
 * uniform window, multichannel).  ws: 2*frames doubles. */
SS_API int ss_alignment_psnr_ssim(const float* w1, const float* w2, double* out, double* ws, int frames, int h,
                           int w, void* stream);
/* path [t][63][2] (stitched smooth path of view 2) -> out[0] = stability score (test_metric_ssd.py:459-468) */
SS_API int ss_stability_score(const float* path, float* out, int t, void* stream);
/* mesh [t][7][9][2] (LR px) -> out[0] = max over frames of inter + intra grid loss (:38-87, 473-482);
 * ws: t floats */
SS_API int ss_distortion_score(const float* mesh, float* out, float* ws, int t, void* stream);

#ifdef __cplusplus
}
#endif
#endif
