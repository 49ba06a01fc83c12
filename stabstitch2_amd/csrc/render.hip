// Dense thin-plate-spline warp and fusion (K12/K13).
//
// The reference materialises a [B,66,Hc*Wc] basis tensor (~500 MB at 720p) and makes ~15
// elementwise passes over it before a bmm and a gather.  Here one thread owns one
// canvas pixel: the 63 RBF terms of both views are evaluated together in registers (packed fp32 math,
// hardware log2) from wave-uniform control points / coefficients (scalar loads, no LDS), the sampling coordinate never leaves the register file, the
// bilinear taps are gathered straight from the planar source frames, and for the AVERAGE mode the
// two (three) views are fused before the single store.  HBM traffic = source frames once (L2 absorbs
// the 4-tap overlap) + the canvas once.
#include "common.h"
#include "device_math.h"

// grid_sample(bilinear, zeros, align_corners=True) taps: x = (xn+1)/2*(W-1), out-of-range taps -> 0
__device__ __forceinline__ float sample_fast(const float* __restrict__ pl, float xn, float yn, int W, int H) {
    float x = ((xn + 1.f) / 2.f) * (float)(W - 1);
    float y = ((yn + 1.f) / 2.f) * (float)(H - 1);
    float xf = fminf(fmaxf(floorf(x), -4.f), (float)W + 4.f);
    float yf = fminf(fmaxf(floorf(y), -4.f), (float)H + 4.f);
    int x0 = (int)xf, y0 = (int)yf, x1 = x0 + 1, y1 = y0 + 1;
    // ATen grid_sampler_2d: nw = (ix_se - ix)(iy_se - iy), ne = (ix - ix_sw)(iy_sw - iy), ...
    float w00 = (xf + 1.f - x) * (yf + 1.f - y), w01 = (x - xf) * (yf + 1.f - y);
    float w10 = (xf + 1.f - x) * (y - yf), w11 = (x - xf) * (y - yf);
    bool vx0 = (unsigned)x0 < (unsigned)W, vx1 = (unsigned)x1 < (unsigned)W;
    bool vy0 = (unsigned)y0 < (unsigned)H, vy1 = (unsigned)y1 < (unsigned)H;
    float r = 0.f;
    if (vx0 && vy0) r += pl[(long long)y0 * W + x0] * w00;
    if (vx1 && vy0) r += pl[(long long)y0 * W + x1] * w01;
    if (vx0 && vy1) r += pl[(long long)y1 * W + x0] * w10;
    if (vx1 && vy1) r += pl[(long long)y1 * W + x1] * w11;
    return r;
}

// zero padding on an all-ones plane: sum of the in-range tap weights (grid_sample of a ones image)
__device__ __forceinline__ float fast_mask(float xn, float yn, int w, int h) {
    float xx = ((xn + 1.f) / 2.f) * (float)(w - 1), yy = ((yn + 1.f) / 2.f) * (float)(h - 1);
    float xf = fminf(fmaxf(floorf(xx), -4.f), (float)w + 4.f);
    float yf = fminf(fmaxf(floorf(yy), -4.f), (float)h + 4.f);
    int x0 = (int)xf, y0 = (int)yf;
    bool vx0 = (unsigned)x0 < (unsigned)w, vx1 = (unsigned)(x0 + 1) < (unsigned)w;
    bool vy0 = (unsigned)y0 < (unsigned)h, vy1 = (unsigned)(y0 + 1) < (unsigned)h;
    float r = 0.f;
    if (vx0 && vy0) r += (xf + 1.f - xx) * (yf + 1.f - yy);
    if (vx1 && vy0) r += (xx - xf) * (yf + 1.f - yy);
    if (vx0 && vy1) r += (xf + 1.f - xx) * (yy - yf);
    if (vx1 && vy1) r += (xx - xf) * (yy - yf);
    return r;
}

// append_mask: emit one extra channel = warp of an all-ones plane (test_online_tra.py:144-147)
__global__ __launch_bounds__(256) void tps_warp_kernel(const float* __restrict__ U, const float* __restrict__ source,
                                                       const float* __restrict__ T, float* __restrict__ out, int c,
                                                       int h, int w, int hc, int wc, int mode, int append_mask) {
    const int b = blockIdx.z;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= wc || y >= hc) return;
    const float* src = source + (long long)b * SS_NV * 2;
    const float* Tx = T + (long long)b * 2 * SS_NT;
    const float* Ty = Tx + SS_NT;
    float gx = linspace_at(-1.f, 1.f, wc, x), gy = linspace_at(-1.f, 1.f, hc, y);
    float xn, yn;
    tps_eval_fast(src, Tx, Ty, gx, gy, xn, yn);
    const long long hw = (long long)h * w, ohw = (long long)hc * wc;
    const int co = c + (append_mask ? 1 : 0);
    float* o = out + (long long)b * co * ohw + (long long)y * wc + x;
    const float* in = U + (long long)b * c * hw;
    if (mode == SS_WARP_NORMAL) {
        SsTaps t = taps_normal(xn, yn, w, h);
        long long ia = (long long)t.y0 * w + t.x0, ib = (long long)t.y1 * w + t.x0;
        long long ic = (long long)t.y0 * w + t.x1, id = (long long)t.y1 * w + t.x1;
        for (int ch = 0; ch < c; ++ch) {
            const float* pl = in + ch * hw;
            o[ch * ohw] = blend4(t, pl[ia], pl[ib], pl[ic], pl[id]);
        }
        if (append_mask) o[c * ohw] = blend4(t, 1.f, 1.f, 1.f, 1.f);
    } else {
        for (int ch = 0; ch < c; ++ch) o[ch * ohw] = sample_fast(in + ch * hw, xn, yn, w, h);
        if (append_mask) o[c * ohw] = fast_mask(xn, yn, w, h);
    }
}

static int launch_warp(const float* U, const float* source, const float* T, float* out, int b, int c, int h, int w,
                       int hc, int wc, int mode, int append_mask, void* stream) {
    if (!U || !source || !T || !out || b <= 0 || c <= 0 || h <= 1 || w <= 1 || hc <= 1 || wc <= 1 ||
        (mode != SS_WARP_NORMAL && mode != SS_WARP_FAST))
        return SS_ERR_ARG;
    dim3 g(ss_cdiv(wc, 64), ss_cdiv(hc, 4), b);
    hipLaunchKernelGGL(tps_warp_kernel, g, dim3(256), 0, (hipStream_t)stream, U, source, T, out, c, h, w, hc, wc, mode,
                       append_mask);
    return ss_launch_status();
}

extern "C" int ss_tps_warp_nchw(const float* U, const float* source, const float* T, float* out, int b, int c, int h,
                                int w, int hc, int wc, int mode, void* stream) {
    return launch_warp(U, source, T, out, b, c, h, w, hc, wc, mode, 0, stream);
}

extern "C" int ss_tps_warp_mask_nchw(const float* U, const float* source, const float* T, float* out, int b, int c,
                                     int h, int w, int hc, int wc, int mode, void* stream) {
    return launch_warp(U, source, T, out, b, c, h, w, hc, wc, mode, 1, stream);
}

// ------------------------------------------------------------------------------------------------
// fused render, AVERAGE fusion (test_online_tra.py:138-142; three-view chaining threeview:486-490)
struct RenderViews {
    const float* img[3];
};

__global__ __launch_bounds__(256) void tps_warp_views_kernel(RenderViews rv, const float* __restrict__ source,
                                                             const float* __restrict__ T, float* __restrict__ out,
                                                             int h, int w, int hc, int wc, int mode) {
    const int b = blockIdx.z;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= wc || y >= hc) return;
    const float* src = source + (long long)b * SS_NV * 2;
    const float* Tx = T + (long long)b * 2 * SS_NT;
    float gx = linspace_at(-1.f, 1.f, wc, x), gy = linspace_at(-1.f, 1.f, hc, y);
    float xn, yn;
    tps_eval_fast(src, Tx, Tx + SS_NT, gx, gy, xn, yn);
    const long long hw = (long long)h * w, ohw = (long long)hc * wc;
    float* o = out + (long long)b * 4 * ohw + (long long)y * wc + x;
    const float* in = b == 0 ? rv.img[0] : (b == 1 ? rv.img[1] : rv.img[2]);
    if (mode == SS_WARP_NORMAL) {
        SsTaps t = taps_normal(xn, yn, w, h);
        long long ia = (long long)t.y0 * w + t.x0, ib = (long long)t.y1 * w + t.x0;
        long long ic = (long long)t.y0 * w + t.x1, id = (long long)t.y1 * w + t.x1;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const float* pl = in + ch * hw;
            o[ch * ohw] = blend4(t, pl[ia], pl[ib], pl[ic], pl[id]);
        }
        o[3 * ohw] = blend4(t, 1.f, 1.f, 1.f, 1.f);
    } else {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) o[ch * ohw] = sample_fast(in + ch * hw, xn, yn, w, h);
        o[3 * ohw] = fast_mask(xn, yn, w, h);
    }
}

extern "C" int ss_tps_warp_views(const float* const* imgs, const float* source, const float* T, float* out, int views,
                                 int h, int w, int hc, int wc, int mode, void* stream) {
    if (!imgs || !source || !T || !out || views < 1 || views > 3 || h <= 1 || w <= 1 || hc <= 1 || wc <= 1 ||
        (mode != SS_WARP_NORMAL && mode != SS_WARP_FAST))
        return SS_ERR_ARG;
    RenderViews rv;
    for (int i = 0; i < 3; ++i) rv.img[i] = i < views ? imgs[i] : nullptr;
    for (int i = 0; i < views; ++i)
        if (!rv.img[i]) return SS_ERR_ARG;
    dim3 g(ss_cdiv(wc, 64), ss_cdiv(hc, 4), views);
    hipLaunchKernelGGL(tps_warp_views_kernel, g, dim3(256), 0, (hipStream_t)stream, rv, source, T, out, h, w, hc, wc,
                       mode);
    return ss_launch_status();
}

// small canvas-sized elementwise helpers of the harnesses
__global__ void affine_kernel(const float* __restrict__ in, float* __restrict__ out, float add, float mul, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = __fmul_rn(__fadd_rn(in[i], add), mul);
}
extern "C" int ss_add_mul(const float* in, float* out, float add, float mul, long long n, void* stream) {
    if (!in || !out || n <= 0) return SS_ERR_ARG;
    hipLaunchKernelGGL(affine_kernel, dim3(ss_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, in, out, add, mul, n);
    return ss_launch_status();
}
__global__ void mask_union_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                  long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = __fsub_rn(__fadd_rn(a[i], b[i]), __fmul_rn(a[i], b[i]));
}
extern "C" int ss_mask_union(const float* a, const float* b, float* out, long long n, void* stream) {
    if (!a || !b || !out || n <= 0) return SS_ERR_ARG;
    hipLaunchKernelGGL(mask_union_kernel, dim3(ss_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, a, b, out, n);
    return ss_launch_status();
}

__device__ __forceinline__ float avg_fuse(float a, float b) {
    float s = __fadd_rn(__fadd_rn(a, b), 1e-6f);
    return __fadd_rn(__fmul_rn(a, a / s), __fmul_rn(b, b / s));
}

template <int VIEWS>
__global__ __launch_bounds__(256) void render_average_kernel(RenderViews rv, const float* __restrict__ source,
                                                             const float* __restrict__ T, float* __restrict__ out,
                                                             int h, int w, int hc, int wc, int mode) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= wc || y >= hc) return;
    float gx = linspace_at(-1.f, 1.f, wc, x), gy = linspace_at(-1.f, 1.f, hc, y);
    const long long hw = (long long)h * w, ohw = (long long)hc * wc;
    float v[VIEWS][3];
    float xs[VIEWS], ys[VIEWS];
    {
        ss_f2 px, py;
        tps_eval_pair(source, source + SS_NV * 2, T, T + 2 * SS_NT, gx, gy, px, py);
        xs[0] = px.x; xs[1] = px.y; ys[0] = py.x; ys[1] = py.y;
        if (VIEWS == 3) {
            tps_eval_pair(source + 2 * SS_NV * 2, source + 2 * SS_NV * 2, T + 4 * SS_NT, T + 4 * SS_NT, gx, gy, px, py);
            xs[VIEWS - 1] = px.x; ys[VIEWS - 1] = py.x;
        }
    }
#pragma unroll
    for (int k = 0; k < VIEWS; ++k) {
        const float xn = xs[k], yn = ys[k];
        const float* in = rv.img[k];
        if (mode == SS_WARP_NORMAL) {
            SsTaps t = taps_normal(xn, yn, w, h);
            long long ia = (long long)t.y0 * w + t.x0, ib = (long long)t.y1 * w + t.x0;
            long long ic = (long long)t.y0 * w + t.x1, id = (long long)t.y1 * w + t.x1;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const float* pl = in + ch * hw;
                v[k][ch] = blend4(t, pl[ia], pl[ib], pl[ic], pl[id]);
            }
        } else {
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) v[k][ch] = sample_fast(in + ch * hw, xn, yn, w, h);
        }
    }
    float* o = out + (long long)y * wc + x;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        float f = avg_fuse(v[0][ch], v[1][ch]);
        if (VIEWS == 3) f = avg_fuse(f, v[2][ch]);
        o[ch * ohw] = f;
    }
}

extern "C" int ss_render_average(const float* const* imgs, const float* source, const float* T, float* out, int views,
                                 int h, int w, int hc, int wc, int mode, void* stream) {
    if (!imgs || !source || !T || !out || (views != 2 && views != 3) || h <= 1 || w <= 1 || hc <= 1 || wc <= 1 ||
        (mode != SS_WARP_NORMAL && mode != SS_WARP_FAST))
        return SS_ERR_ARG;
    RenderViews rv;
    for (int i = 0; i < 3; ++i) rv.img[i] = i < views ? imgs[i] : nullptr;
    for (int i = 0; i < views; ++i)
        if (!rv.img[i]) return SS_ERR_ARG;
    dim3 g(ss_cdiv(wc, 64), ss_cdiv(hc, 4), 1);
    hipStream_t st = (hipStream_t)stream;
    if (views == 2)
        hipLaunchKernelGGL((render_average_kernel<2>), g, dim3(256), 0, st, rv, source, T, out, h, w, hc, wc, mode);
    else
        hipLaunchKernelGGL((render_average_kernel<3>), g, dim3(256), 0, st, rv, source, T, out, h, w, hc, wc, mode);
    return ss_launch_status();
}

// ------------------------------------------------------------------------------------------------
// LINEAR fusion (linear_blender, test_online_tra.py:34-58).  Whole-canvas statistics (mask centroids,
// projection range of the overlap) are accumulated with device atomics into a 16-word scalar block at
// the head of the workspace; no host round trip.
//   ws: [0..15] scalars | X [hc*wc] | tmp [hc*wc]
//   scalars (as 64-bit words): 0 cnt1, 1 sumr1, 2 sumc1, 3 cnt2, 4 sumr2, 5 sumc2; floats at [12] pmin-key, [13] pmax-key
__device__ __forceinline__ unsigned f2key(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__global__ void lb_init_kernel(unsigned long long* s) {
    if (threadIdx.x < 6) s[threadIdx.x] = 0ull;
    unsigned* u = reinterpret_cast<unsigned*>(s);
    if (threadIdx.x == 0) { u[12] = 0xffffffffu; u[13] = 0u; }
}

// 2-D indexing (block = 64 columns x 4 rows, grid-stride over row groups): no per-pixel integer division
__global__ __launch_bounds__(256) void lb_centroid_kernel(const float* __restrict__ m1, const float* __restrict__ m2,
                                                          unsigned long long* s, int hc, int wc) {
    __shared__ unsigned long long red[6];
    if (threadIdx.x < 6) red[threadIdx.x] = 0ull;
    __syncthreads();
    unsigned c1 = 0, r1 = 0, k1 = 0, c2 = 0, r2 = 0, k2 = 0;      // per-thread partials fit 32 bits (<= hc/ (4 gridDim.y) rows)
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    if (c < wc) {
        for (int r = blockIdx.y * 4 + (threadIdx.x >> 6); r < hc; r += gridDim.y * 4) {
            long long i = (long long)r * wc + c;
            if (m1[i] != 0.f) { k1 += 1; r1 += r; c1 += c; }
            if (m2[i] != 0.f) { k2 += 1; r2 += r; c2 += c; }
        }
    }
    unsigned v[6] = {k1, r1, c1, k2, r2, c2};       // a wave's sums still fit 32 bits (<= 64 lanes x ~24 rows x 1882)
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        unsigned t = v[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
        if ((threadIdx.x & 63) == 0) atomicAdd(&red[k], (unsigned long long)t);
    }
    __syncthreads();
    if (threadIdx.x < 6) atomicAdd(&s[threadIdx.x], red[threadIdx.x]);
}

struct LbCenters {
    float c1r, c1c, v0, v1;
};
__device__ __forceinline__ LbCenters lb_centers(const unsigned long long* s) {
    LbCenters L;
    L.c1r = (float)((double)s[1] / (double)s[0]);
    L.c1c = (float)((double)s[2] / (double)s[0]);
    float c2r = (float)((double)s[4] / (double)s[3]);
    float c2c = (float)((double)s[5] / (double)s[3]);
    L.v0 = __fsub_rn(c2r, L.c1r);
    L.v1 = __fsub_rn(c2c, L.c1c);
    return L;
}
__device__ __forceinline__ float lb_proj(const LbCenters& L, int r, int c) {
    return __fadd_rn(__fmul_rn(__fsub_rn((float)r, L.c1r), L.v0), __fmul_rn(__fsub_rn((float)c, L.c1c), L.v1));
}
__device__ __forceinline__ bool lb_overlap(float a, float b) { return rintf(__fmul_rn(a, b)) != 0.f; }

__global__ __launch_bounds__(256) void lb_range_kernel(const float* __restrict__ m1, const float* __restrict__ m2,
                                                       unsigned long long* s, int hc, int wc) {
    // the centres cost four fp64 divisions: once per workgroup, not per thread (that was 55 us per frame)
    __shared__ LbCenters Ls;
    __shared__ float smn[4], smx[4];
    if (threadIdx.x == 0) Ls = lb_centers(s);
    __syncthreads();
    const LbCenters L = Ls;
    float mn = INFINITY, mx = -INFINITY;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    if (c < wc) {
        for (int r = blockIdx.y * 4 + (threadIdx.x >> 6); r < hc; r += gridDim.y * 4) {
            long long i = (long long)r * wc + c;
            if (lb_overlap(m1[i], m2[i])) {
                float p = lb_proj(L, r, c);
                mn = fminf(mn, p);
                mx = fmaxf(mx, p);
            }
        }
    }
    mn = ss_wave_min(mn);
    mx = ss_wave_max(mx);
    if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        mn = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
        mx = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
        unsigned* u = reinterpret_cast<unsigned*>(s);
        if (mn != INFINITY) atomicMin(&u[12], f2key(mn));
        if (mx != -INFINITY) atomicMax(&u[13], f2key(mx));
    }
}

// X = ref_only + (1 - ovl_mask) * m1
__global__ void lb_premask_kernel(const float* __restrict__ m1, const float* __restrict__ m2,
                                  const unsigned long long* s, float* __restrict__ X, int hc, int wc) {
    __shared__ LbCenters Ls;
    if (threadIdx.x == 0) Ls = lb_centers(s);
    __syncthreads();
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), r = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (c >= wc || r >= hc) return;
    const long long i = (long long)r * wc + c;
    const unsigned* u = reinterpret_cast<const unsigned*>(s);
    float a = m1[i], b = m2[i];
    float ovl = rintf(__fmul_rn(a, b));
    float ref_only = __fsub_rn(a, ovl);
    float om = 0.f;
    if (ovl != 0.f) {
        const LbCenters L = Ls;
        float pmin = key2f(u[12]), pmax = key2f(u[13]);
        om = __fsub_rn(lb_proj(L, r, c), pmin) / __fadd_rn(__fsub_rn(pmax, pmin), 1e-3f);
    }
    X[i] = __fadd_rn(ref_only, __fmul_rn(__fsub_rn(1.f, om), a));
}

struct Gauss21 {
    float k[21];
};

__device__ __forceinline__ int reflect_idx(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}

// separable 21-tap Gaussian, reflect border (torchvision GaussianBlur((21,21), 20))
__global__ void lb_blur_kernel(const float* __restrict__ in, float* __restrict__ out, int hc, int wc, int vertical,
                               Gauss21 g) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), r = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (c >= wc || r >= hc) return;
    const long long i = (long long)r * wc + c;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 21; ++k) {
        int rr = vertical ? reflect_idx(r + k - 10, hc) : r;
        int cc = vertical ? c : reflect_idx(c + k - 10, wc);
        acc = fmaf(g.k[k], in[(long long)rr * wc + cc], acc);
    }
    out[i] = acc;
}

__global__ void lb_final_kernel(const float* __restrict__ ref, const float* __restrict__ tgt,
                                const float* __restrict__ m1, const float* __restrict__ m2,
                                const float* __restrict__ blur, float* __restrict__ out, float* __restrict__ mask1_out,
                                int hc, int wc) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), r = blockIdx.y * 4 + (threadIdx.x >> 6);
    const long long n = (long long)hc * wc;
    if (c >= wc || r >= hc) return;
    const long long i = (long long)r * wc + c;
    float a = m1[i], b = m2[i];
    float ovl = rintf(__fmul_rn(a, b));
    float ref_only = __fsub_rn(a, ovl);
    float mk = fminf(fmaxf(__fadd_rn(__fmul_rn(blur[i], a), ref_only), 0.f), 1.f);
    if (mask1_out) mask1_out[i] = mk;
    if (out) {
        float mk2 = __fmul_rn(__fsub_rn(1.f, mk), b);
#pragma unroll
        for (int ch = 0; ch < 3; ++ch)
            out[ch * n + i] = __fadd_rn(__fmul_rn(ref[ch * n + i], mk), __fmul_rn(tgt[ch * n + i], mk2));
    }
}

extern "C" long long ss_linear_blend_workspace_floats(int hc, int wc) { return 32 + 2ll * hc * wc; }

extern "C" int ss_linear_blend(const float* ref, const float* tgt, const float* ref_m, const float* tgt_m, float* out,
                               float* mask1_out, int hc, int wc, float* ws, void* stream) {
    if (!ref_m || !tgt_m || !ws || (!out && !mask1_out) || (out && (!ref || !tgt)) || hc < 11 || wc < 11)
        return SS_ERR_ARG;
    // 1-D kernel exp(-0.5 (t/sigma)^2) on linspace(-10,10,21), normalised, fp32 like torchvision
    Gauss21 g;
    {
        float s = 0.f;
        for (int i = 0; i < 21; ++i) {
            float t = (float)(i - 10) / 20.0f;
            g.k[i] = expf(-0.5f * (t * t));
            s += g.k[i];
        }
        for (int i = 0; i < 21; ++i) g.k[i] /= s;
    }
    hipStream_t st = (hipStream_t)stream;
    unsigned long long* sc = reinterpret_cast<unsigned long long*>(ws);
    float* X = ws + 32;
    float* tmp = X + (long long)hc * wc;
    hipLaunchKernelGGL(lb_init_kernel, dim3(1), dim3(64), 0, st, sc);
    // few, long-running workgroups for the two reductions: every workgroup ends in atomics on the same scalars
    dim3 rg(ss_cdiv(wc, 64), ss_cdiv(hc, 16) < 8 ? ss_cdiv(hc, 16) : 8);
    hipLaunchKernelGGL(lb_centroid_kernel, rg, dim3(256), 0, st, ref_m, tgt_m, sc, hc, wc);
    hipLaunchKernelGGL(lb_range_kernel, rg, dim3(256), 0, st, ref_m, tgt_m, sc, hc, wc);
    const dim3 pg(ss_cdiv(wc, 64), ss_cdiv(hc, 4));
    hipLaunchKernelGGL(lb_premask_kernel, pg, dim3(256), 0, st, ref_m, tgt_m, (const unsigned long long*)sc, X, hc, wc);
    hipLaunchKernelGGL(lb_blur_kernel, pg, dim3(256), 0, st, (const float*)X, tmp, hc, wc, 0, g);
    hipLaunchKernelGGL(lb_blur_kernel, pg, dim3(256), 0, st, (const float*)tmp, X, hc, wc, 1, g);
    hipLaunchKernelGGL(lb_final_kernel, pg, dim3(256), 0, st, ref, tgt, ref_m, tgt_m, (const float*)X, out,
                       mask1_out, hc, wc);
    return ss_launch_status();
}
