// Device-side math shared by geom.hip and render.hip (fp32, reference operation order).
#pragma once
#include "common.h"

// torch.linspace(0, size, steps)[i] in fp32 (two-sided formula of ATen's linspace)
__device__ __forceinline__ float linspace_at(float start, float end, int steps, int i) {
    float step = (end - start) / (float)(steps - 1);
    return i < steps / 2 ? start + step * (float)i : end - step * (float)(steps - 1 - i);
}

// bilinear core shared by the homography and TPS samplers (utils/torch_homo_transform.py:50-125):
// indices clamped to the image, the CLAMPED values enter the weights.
struct SsTaps {
    int x0, x1, y0, y1;
    float wa, wb, wc, wd;
};
__device__ __forceinline__ SsTaps taps_normal(float xn, float yn, int W, int H) {
    float x = __fmul_rn(__fadd_rn(xn, 1.0f), (float)W) / 2.0f;
    float y = __fmul_rn(__fadd_rn(yn, 1.0f), (float)H) / 2.0f;
    // keep the float->int conversion defined for wild coordinates (they are fully clamped anyway)
    float xf = fminf(fmaxf(floorf(x), -4.0f), (float)W + 4.0f);
    float yf = fminf(fmaxf(floorf(y), -4.0f), (float)H + 4.0f);
    int x0 = (int)xf, y0 = (int)yf;
    int x1 = x0 + 1, y1 = y0 + 1;
    SsTaps t;
    t.x0 = min(max(x0, 0), W - 1);
    t.x1 = min(max(x1, 0), W - 1);
    t.y0 = min(max(y0, 0), H - 1);
    t.y1 = min(max(y1, 0), H - 1);
    float x0f = (float)t.x0, x1f = (float)t.x1, y0f = (float)t.y0, y1f = (float)t.y1;
    t.wa = __fmul_rn(__fsub_rn(x1f, x), __fsub_rn(y1f, y));
    t.wb = __fmul_rn(__fsub_rn(x1f, x), __fsub_rn(y, y0f));
    t.wc = __fmul_rn(__fsub_rn(x, x0f), __fsub_rn(y1f, y));
    t.wd = __fmul_rn(__fsub_rn(x, x0f), __fsub_rn(y, y0f));
    return t;
}
__device__ __forceinline__ float blend4(const SsTaps& t, float ia, float ib, float ic, float id) {
    return __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(t.wa, ia), __fmul_rn(t.wb, ib)), __fmul_rn(t.wc, ic)),
                     __fmul_rn(t.wd, id));
}

__device__ __forceinline__ float tps_rbf(float dx, float dy) {
    float d2 = __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));
    return __fmul_rn(d2, logf(__fadd_rn(d2, 1e-6f)));
}

// spline value at (x, y): T . [1, x, y, r_1..r_63], sequential fp32 accumulation
__device__ __forceinline__ void tps_eval(const float* __restrict__ sx, const float* __restrict__ sy,
                                         const float* __restrict__ Tx, const float* __restrict__ Ty, float x, float y,
                                         float& ox, float& oy) {
    float ax = fmaf(Tx[2], y, fmaf(Tx[1], x, Tx[0]));
    float ay = fmaf(Ty[2], y, fmaf(Ty[1], x, Ty[0]));
#pragma unroll 9
    for (int k = 0; k < SS_NV; ++k) {
        float r = tps_rbf(__fsub_rn(x, sx[k]), __fsub_rn(y, sy[k]));
        ax = fmaf(Tx[3 + k], r, ax);
        ay = fmaf(Ty[3 + k], r, ay);
    }
    ox = ax;
    oy = ay;
}

// same spline, control points interleaved (x,y) in global memory; every index is wave-uniform so the
// compiler turns the loads into scalar (SGPR) loads -- no LDS traffic in the dense warp.
__device__ __forceinline__ void tps_eval_interleaved(const float* __restrict__ src, const float* __restrict__ Tx,
                                                     const float* __restrict__ Ty, float x, float y, float& ox,
                                                     float& oy) {
    float ax = fmaf(Tx[2], y, fmaf(Tx[1], x, Tx[0]));
    float ay = fmaf(Ty[2], y, fmaf(Ty[1], x, Ty[0]));
#pragma unroll 9
    for (int k = 0; k < SS_NV; ++k) {
        float r = tps_rbf(__fsub_rn(x, src[2 * k]), __fsub_rn(y, src[2 * k + 1]));
        ax = fmaf(Tx[3 + k], r, ax);
        ay = fmaf(Ty[3 + k], r, ay);
    }
    ox = ax;
    oy = ay;
}

// The dense warp's spline evaluation.  One arithmetic, two shapes (a scalar one for the generic per-view warp and a
// packed one for the fused render; same operations in the same order -> bit-identical coordinates, so the fused kernel
// can be checked against the chained one exactly):
//     per control point k:  dx = x - sx_k;  d2 = fma(dx, dx, dy2_k)  with dy2_k = fl((y - sy_k)^2);
//                           r  = d2 * log2(d2 + 1e-6);   Sx = fma(tx_k, r, Sx);  Sy = fma(ty_k, r, Sy)
//     result:               fma(Sx, ln 2, T0 + T1 x + T2 y)      (the spline's radial sum is formed in log2 units)
// log2 is the hardware v_log_f32 (1 ulp).  The argument lies in [1e-6, ~16], so no denormal / range handling is
// needed; the result differs from an accurate logf by <= ~1.5 ulp, the same class as the reference's own vectorised
// logf, and two orders below the reference's fp32-vs-fp64 coordinate noise.
typedef float ss_f2 __attribute__((ext_vector_type(2)));
#define SS_LN2 0.6931471805599453f

__device__ __forceinline__ void tps_eval_fast(const float* __restrict__ src, const float* __restrict__ Tx,
                                              const float* __restrict__ Ty, float x, float y, float& ox, float& oy) {
    float sx = 0.f, sy = 0.f;
#pragma unroll 9
    for (int k = 0; k < SS_NV; ++k) {
        const float dx = __fsub_rn(x, src[2 * k]), dy = __fsub_rn(y, src[2 * k + 1]);
        const float d2 = fmaf(dx, dx, __fmul_rn(dy, dy));
        const float r = __fmul_rn(d2, __builtin_amdgcn_logf(__fadd_rn(d2, 1e-6f)));
        sx = fmaf(Tx[3 + k], r, sx);
        sy = fmaf(Ty[3 + k], r, sy);
    }
    ox = fmaf(sx, SS_LN2, fmaf(Tx[2], y, fmaf(Tx[1], x, Tx[0])));
    oy = fmaf(sy, SS_LN2, fmaf(Ty[2], y, fmaf(Ty[1], x, Ty[0])));
}

// ONE spline at the same column x of TWO canvas rows (y0, y1), packed over the rows.  Everything that depends on the
// row only -- dy2_k of both rows -- is wave-uniform in the fused render (a wave = 64 columns of the same two rows): the
// wave computes the 63 pairs once (lane k: control point k, tps_rows_table) and every lane reads them back as LDS
// broadcasts; dx is shared by the two rows.  Per control point and lane: 6 VALU instructions + 2 v_log_f32 for two
// points (the version that evaluated dx, dy and their squares per lane: 9 + 2).
//
// FOLD (opt-in, SS_WARP_EPS_FOLD): the reference's `+ 1e-6` is folded into the row table and the radial term
// becomes a log2(a) with a = fma(dx, dx, dy2 + 1e-6) -- one packed instruction of five less per control point beside the two
// quarter-rate logs -- instead of d2 log2(d2 + 1e-6): the term changes by 1e-6 log2(a) (<= 2e-5 of a term of order one), the
// sampling coordinate by ~1e-3 px at 720p.  NOT the reference's arithmetic: never the default, never the headline.
template <bool FOLD = false>
__device__ __forceinline__ void tps_rows_table(const float* __restrict__ src, float y0, float y1, int lane,
                                               ss_f2* __restrict__ tab) {
    if (lane < SS_NV) {
        const float sy = src[2 * lane + 1];
        const float d0 = __fsub_rn(y0, sy), d1 = __fsub_rn(y1, sy);
        tab[lane] = FOLD ? (ss_f2){__fadd_rn(__fmul_rn(d0, d0), 1e-6f), __fadd_rn(__fmul_rn(d1, d1), 1e-6f)}
                         : (ss_f2){__fmul_rn(d0, d0), __fmul_rn(d1, d1)};
    }
}
template <bool FOLD = false>
__device__ __forceinline__ void tps_eval_rows(const float* __restrict__ src, const float* __restrict__ T,
                                              const ss_f2* __restrict__ tab, float x, float y0, float y1, ss_f2& ox,
                                              ss_f2& oy) {
    const float* Tx = T;
    const float* Ty = T + SS_NT;
    ss_f2 sx = {0.f, 0.f}, sy = {0.f, 0.f};
    const ss_f2 eps = {1e-6f, 1e-6f};
#pragma unroll 9
    for (int k = 0; k < SS_NV; ++k) {
        const float dx = __fsub_rn(x, src[2 * k]);
        const ss_f2 dxx = {dx, dx};
        const ss_f2 d2 = __builtin_elementwise_fma(dxx, dxx, tab[k]);
        const ss_f2 a = FOLD ? d2 : d2 + eps;
        const ss_f2 lg = {__builtin_amdgcn_logf(a.x), __builtin_amdgcn_logf(a.y)};
        const ss_f2 r = d2 * lg;
        const ss_f2 tx = {Tx[3 + k], Tx[3 + k]};
        const ss_f2 ty = {Ty[3 + k], Ty[3 + k]};
        sx = __builtin_elementwise_fma(tx, r, sx);
        sy = __builtin_elementwise_fma(ty, r, sy);
    }
    const ss_f2 ax = {fmaf(Tx[2], y0, fmaf(Tx[1], x, Tx[0])), fmaf(Tx[2], y1, fmaf(Tx[1], x, Tx[0]))};
    const ss_f2 ay = {fmaf(Ty[2], y0, fmaf(Ty[1], x, Ty[0])), fmaf(Ty[2], y1, fmaf(Ty[1], x, Ty[0]))};
    const ss_f2 ln2 = {SS_LN2, SS_LN2};
    ox = __builtin_elementwise_fma(sx, ln2, ax);
    oy = __builtin_elementwise_fma(sy, ln2, ay);
}

__device__ __forceinline__ float norm1(float v, float size) { return __fsub_rn(__fmul_rn(v, 2.0f) / size, 1.0f); }
__device__ __forceinline__ float recover1(float v, float size) { return __fmul_rn(__fadd_rn(v, 1.0f), size) / 2.0f; }

