// Frame I/O either side of the hot path: uint8 ingest with OpenCV-exact LR generation,
// and the fp32 -> uint8 video-frame cast.  HBM-bound byte work: one pass over the uint8 frame per output.
#include "common.h"

// ---------------------------------------------------------------------------------------------------------------
// HR planes: uint8 [n][h][w][3] -> fp32 [n][3][h][w] (test_online_tra.py:254-256).  One thread = 4 pixels
// (12 bytes in, one float4 per plane out) when w % 4 == 0, else one pixel.
__global__ __launch_bounds__(256) void ingest_hr4_kernel(const uint32_t* __restrict__ in, float* __restrict__ out,
                                                         long long groups, int plane4) {
    long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    if (g >= groups) return;
    uint32_t a = in[3 * g], b = in[3 * g + 1], c = in[3 * g + 2];
    // bytes: a = p0c0 p0c1 p0c2 p1c0 | b = p1c1 p1c2 p2c0 p2c1 | c = p2c2 p3c0 p3c1 p3c2
    float4 c0 = make_float4((float)(a & 255u), (float)(a >> 24), (float)((b >> 16) & 255u), (float)((c >> 8) & 255u));
    float4 c1 = make_float4((float)((a >> 8) & 255u), (float)(b & 255u), (float)(b >> 24), (float)((c >> 16) & 255u));
    float4 c2 = make_float4((float)((a >> 16) & 255u), (float)((b >> 8) & 255u), (float)(c & 255u), (float)(c >> 24));
    long long img = g / plane4, r = g - img * plane4;
    float4* o = reinterpret_cast<float4*>(out) + img * 3 * plane4 + r;
    o[0] = c0;
    o[plane4] = c1;
    o[2 * (long long)plane4] = c2;
}

__global__ __launch_bounds__(256) void ingest_hr1_kernel(const uint8_t* __restrict__ in, float* __restrict__ out,
                                                         long long pixels, int plane) {
    long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= pixels) return;
    long long img = p / plane, r = p - img * plane;
    float* o = out + img * 3 * plane + r;
    o[0] = (float)in[3 * p];
    o[plane] = (float)in[3 * p + 1];
    o[2 * (long long)plane] = (float)in[3 * p + 2];
}

// ---------------------------------------------------------------------------------------------------------------
// cv2.resize INTER_LINEAR for uint8 (OpenCV 4.5.1 resize.cpp) fused with `/127.5 - 1.0` (test_online_tra.py:259-262).
// Tap tables are rebuilt per thread exactly as the host code builds them: double (d+0.5)*scale-0.5 -> float,
// floor, float fraction, round-half-even to 11-bit fixed point (the file is compiled with -ffp-contract=off).
struct LinTap {
    int s;
    int w0, w1;
};
__device__ __forceinline__ int sat_short_round(float v) {
    float r = rintf(v);
    r = fminf(fmaxf(r, -32768.f), 32767.f);
    return (int)r;
}
__device__ __forceinline__ LinTap lin_tap(int d, double scale, int src, bool clamp_frac) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f = f - (float)s;
    if (clamp_frac) {
        if (s < 0) { f = 0.f; s = 0; }
        if (s >= src - 1) { f = 0.f; s = src - 1; }
    }
    LinTap t;
    t.s = s;
    t.w0 = sat_short_round((1.f - f) * 2048.f);
    t.w1 = sat_short_round(f * 2048.f);
    return t;
}
__device__ __forceinline__ float lr_norm(int v) { return __fsub_rn(__fdiv_rn((float)v, 127.5f), 1.0f); }

// mode 0: general linear, 1: exact 2x2 area, 2: same size
template <int MODE>
__global__ __launch_bounds__(256) void ingest_lr_kernel(const uint8_t* __restrict__ in, float* __restrict__ out, int n,
                                                        int h, int w, int lr_h, int lr_w, double scale_x,
                                                        double scale_y) {
    int x = blockIdx.x * 64 + (threadIdx.x & 63);
    int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    int img = blockIdx.z;
    if (x >= lr_w || y >= lr_h) return;
    const uint8_t* f = in + (long long)img * h * w * 3;
    int v[3];
    if (MODE == 2) {
        const uint8_t* p = f + ((long long)y * w + x) * 3;
        v[0] = p[0]; v[1] = p[1]; v[2] = p[2];
    } else if (MODE == 1) {
        const uint8_t* p0 = f + ((long long)(2 * y) * w + 2 * x) * 3;
        const uint8_t* p1 = p0 + (long long)w * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = (p0[c] + p0[3 + c] + p1[c] + p1[3 + c] + 2) >> 2;
    } else {
        LinTap tx = lin_tap(x, scale_x, w, true);
        LinTap ty = lin_tap(y, scale_y, h, false);
        int x0 = tx.s, x1 = min(tx.s + 1, w - 1);
        int r0 = min(max(ty.s, 0), h - 1), r1 = min(max(ty.s + 1, 0), h - 1);
        const uint8_t* p0 = f + (long long)r0 * w * 3;
        const uint8_t* p1 = f + (long long)r1 * w * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            int s0 = p0[x0 * 3 + c] * tx.w0 + p0[x1 * 3 + c] * tx.w1;
            int s1 = p1[x0 * 3 + c] * tx.w0 + p1[x1 * 3 + c] * tx.w1;
            int r = (((ty.w0 * (s0 >> 4)) >> 16) + ((ty.w1 * (s1 >> 4)) >> 16) + 2) >> 2;
            v[c] = min(max(r, 0), 255);
        }
    }
    long long plane = (long long)lr_h * lr_w;
    float* o = out + (long long)img * 3 * plane + (long long)y * lr_w + x;
    o[0] = lr_norm(v[0]);
    o[plane] = lr_norm(v[1]);
    o[2 * plane] = lr_norm(v[2]);
}

extern "C" int ss_ingest_u8(const unsigned char* frames, float* hr, float* lr, int n, int h, int w, int lr_h, int lr_w,
                            void* stream) {
    if (!frames || (!hr && !lr) || n < 0 || h < 1 || w < 1) return SS_ERR_ARG;
    if (lr && (lr_h < 1 || lr_w < 1)) return SS_ERR_ARG;
    if (n == 0) return SS_OK;
    hipStream_t st = (hipStream_t)stream;
    long long pixels = (long long)n * h * w;
    if (hr) {
        if ((w & 3) == 0 && (((uintptr_t)frames) & 3) == 0 && (((uintptr_t)hr) & 15) == 0) {
            long long groups = pixels / 4;
            if (groups / 256 + 1 > 0x7fffffffLL) return SS_ERR_ARG;
            hipLaunchKernelGGL(ingest_hr4_kernel, dim3(ss_cdiv(groups, 256)), dim3(256), 0, st,
                               reinterpret_cast<const uint32_t*>(frames), hr, groups, h * w / 4);
        } else {
            if (pixels / 256 + 1 > 0x7fffffffLL) return SS_ERR_ARG;
            hipLaunchKernelGGL(ingest_hr1_kernel, dim3(ss_cdiv(pixels, 256)), dim3(256), 0, st, frames, hr, pixels,
                               h * w);
        }
    }
    if (lr) {
        if (n > 65535) return SS_ERR_ARG;
        dim3 grid(ss_cdiv(lr_w, 64), ss_cdiv(lr_h, 4), n);
        double scale_x = 1.0 / ((double)lr_w / (double)w), scale_y = 1.0 / ((double)lr_h / (double)h);
        if (lr_w == w && lr_h == h)
            hipLaunchKernelGGL(ingest_lr_kernel<2>, grid, dim3(256), 0, st, frames, lr, n, h, w, lr_h, lr_w, scale_x,
                               scale_y);
        else if (w == 2 * lr_w && h == 2 * lr_h)
            hipLaunchKernelGGL(ingest_lr_kernel<1>, grid, dim3(256), 0, st, frames, lr, n, h, w, lr_h, lr_w, scale_x,
                               scale_y);
        else
            hipLaunchKernelGGL(ingest_lr_kernel<0>, grid, dim3(256), 0, st, frames, lr, n, h, w, lr_h, lr_w, scale_x,
                               scale_y);
    }
    return ss_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------
// `.astype(np.uint8)` of the fused canvas (test_online_tra.py:151, :413): fp32 [n][3][h][w] -> uint8 [n][h][w][3]
__device__ __forceinline__ uint32_t to_u8(float v) {
    if (!(fabsf(v) < 2147483648.f)) return 0u;     // x86 cvttss2si "integer indefinite" 0x80000000 -> low byte 0
    return (uint32_t)((int)v) & 255u;
}

__global__ __launch_bounds__(256) void canvas_u8x4_kernel(const float* __restrict__ in, uint32_t* __restrict__ out,
                                                          long long groups, int plane4) {
    long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    if (g >= groups) return;
    long long img = g / plane4, r = g - img * plane4;
    const float4* p = reinterpret_cast<const float4*>(in) + img * 3 * plane4 + r;
    float4 c0 = p[0], c1 = p[plane4], c2 = p[2 * (long long)plane4];
    uint32_t a = to_u8(c0.x) | (to_u8(c1.x) << 8) | (to_u8(c2.x) << 16) | (to_u8(c0.y) << 24);
    uint32_t b = to_u8(c1.y) | (to_u8(c2.y) << 8) | (to_u8(c0.z) << 16) | (to_u8(c1.z) << 24);
    uint32_t c = to_u8(c2.z) | (to_u8(c0.w) << 8) | (to_u8(c1.w) << 16) | (to_u8(c2.w) << 24);
    out[3 * g] = a;
    out[3 * g + 1] = b;
    out[3 * g + 2] = c;
}

__global__ __launch_bounds__(256) void canvas_u8x1_kernel(const float* __restrict__ in, uint8_t* __restrict__ out,
                                                          long long pixels, int plane) {
    long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= pixels) return;
    long long img = p / plane, r = p - img * plane;
    const float* s = in + img * 3 * plane + r;
    out[3 * p] = (uint8_t)to_u8(s[0]);
    out[3 * p + 1] = (uint8_t)to_u8(s[plane]);
    out[3 * p + 2] = (uint8_t)to_u8(s[2 * (long long)plane]);
}

extern "C" int ss_canvas_to_u8(const float* canvas, unsigned char* out, int n, int h, int w, void* stream) {
    if (!canvas || !out || n < 0 || h < 1 || w < 1) return SS_ERR_ARG;
    if (n == 0) return SS_OK;
    hipStream_t st = (hipStream_t)stream;
    long long pixels = (long long)n * h * w;
    if (pixels / 256 + 1 > 0x7fffffffLL) return SS_ERR_ARG;
    if (((h * (long long)w) & 3) == 0 && (((uintptr_t)canvas) & 15) == 0 && (((uintptr_t)out) & 3) == 0) {
        long long groups = pixels / 4;
        hipLaunchKernelGGL(canvas_u8x4_kernel, dim3(ss_cdiv(groups, 256)), dim3(256), 0, st, canvas,
                           reinterpret_cast<uint32_t*>(out), groups, (int)(h * (long long)w / 4));
    } else {
        hipLaunchKernelGGL(canvas_u8x1_kernel, dim3(ss_cdiv(pixels, 256)), dim3(256), 0, st, canvas, out, pixels,
                           h * w);
    }
    return ss_launch_status();
}
