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

__device__ __forceinline__ float norm1(float v, float size) { return __fsub_rn(__fmul_rn(v, 2.0f) / size, 1.0f); }
__device__ __forceinline__ float recover1(float v, float size) { return __fmul_rn(__fadd_rn(v, 1.0f), size) / 2.0f; }

