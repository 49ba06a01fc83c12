// Correlation kernels: local cost volume (K8) and the contextual correlation layer (K3).
#include "common.h"

extern "C" int ss_conv_nhwc(const float*, const float*, const float*, const float*, float*, int, int, int, int, int,
                            int, int, int, int, int, int, int, int, int, int, int, long long, long long, long long,
                            float*, long long, void*);

// ------------------------------------------------------------------------------------------------
// cost volume.  Block = 4x16 output pixels of one image; channels walked in chunks of 32 staged
// channel-major in LDS (x2 window incl. halo + x1 tile); thread = (pixel, displacement class d%4),
// ~(2R+1)^2/4 accumulators in registers; results go back through LDS for coalesced NHWC stores.
#define CV_TY 4
#define CV_TX 16
#define CV_CC 32

template <int R>
__global__ __launch_bounds__(256) void cost_volume_kernel(const float* __restrict__ x1, const float* __restrict__ x2,
                                                          float* __restrict__ out, int h, int w, int c, int out_cs) {
    constexpr int KD = 2 * R + 1;
    constexpr int D = KD * KD;
    constexpr int NACC = (D + 3) / 4;
    constexpr int WH = CV_TY + 2 * R;
    constexpr int WW = CV_TX + 2 * R;
    constexpr int WPIX = WH * WW;
    constexpr int OUTF = CV_TY * CV_TX * (D + 3);   // staging for the epilogue
    constexpr int X2F = CV_CC * WPIX;
    constexpr int LDSF = (X2F > OUTF ? X2F : OUTF);
    __shared__ float s2[LDSF];
    __shared__ float s1[CV_CC][CV_TY * CV_TX];

    const int tid = threadIdx.x;
    const int n = blockIdx.z;
    const int y0 = blockIdx.y * CV_TY, x0 = blockIdx.x * CV_TX;
    const int px = tid & 63, dg = tid >> 6;
    const int py = px >> 4, pxx = px & 15;
    const float* x1n = x1 + (long long)n * h * w * c;
    const float* x2n = x2 + (long long)n * h * w * c;

    float acc[NACC];
    int off[NACC];   // LDS offset of displacement d = dg + 4e inside the window (0 for the unused tail)
#pragma unroll
    for (int e = 0; e < NACC; ++e) {
        acc[e] = 0.f;
        int d = dg + 4 * e;
        int j = d / KD, i = d - j * KD;
        off[e] = d < D ? j * WW + i : 0;
    }

    for (int c0 = 0; c0 < c; c0 += CV_CC) {
        // stage x2 window [cc][row][col] and x1 tile [cc][pixel]
        for (int e = tid; e < WPIX * (CV_CC / 4); e += 256) {
            int q = e % (CV_CC / 4);
            int wp = e / (CV_CC / 4);
            int wy = wp / WW, wx = wp - wy * WW;
            int yy = y0 - R + wy, xx = x0 - R + wx;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)w && c0 + q * 4 < c)
                v = *reinterpret_cast<const float4*>(x2n + ((long long)yy * w + xx) * c + c0 + q * 4);
            s2[(q * 4 + 0) * WPIX + wp] = v.x;
            s2[(q * 4 + 1) * WPIX + wp] = v.y;
            s2[(q * 4 + 2) * WPIX + wp] = v.z;
            s2[(q * 4 + 3) * WPIX + wp] = v.w;
        }
        for (int e = tid; e < CV_TY * CV_TX * (CV_CC / 4); e += 256) {
            int q = e % (CV_CC / 4);
            int p = e / (CV_CC / 4);
            int yy = y0 + (p >> 4), xx = x0 + (p & 15);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (yy < h && xx < w && c0 + q * 4 < c)
                v = *reinterpret_cast<const float4*>(x1n + ((long long)yy * w + xx) * c + c0 + q * 4);
            s1[q * 4 + 0][p] = v.x;
            s1[q * 4 + 1][p] = v.y;
            s1[q * 4 + 2][p] = v.z;
            s1[q * 4 + 3][p] = v.w;
        }
        __syncthreads();
#pragma unroll 4
        for (int cc = 0; cc < CV_CC; ++cc) {
            float a = s1[cc][px];
            const float* win = s2 + cc * WPIX + py * WW + pxx;
#pragma unroll
            for (int e = 0; e < NACC; ++e) acc[e] = fmaf(a, win[off[e]], acc[e]);
        }
        __syncthreads();
    }
    // epilogue through LDS: [pixel][D+3]
    const float inv_c = (float)c;
#pragma unroll
    for (int e = 0; e < NACC; ++e) {
        int d = dg + 4 * e;
        if (d < D) {
            float v = acc[e] / inv_c;
            s2[px * (D + 3) + d] = v > 0.f ? v : 0.1f * v;
        }
    }
    __syncthreads();
    for (int e = tid; e < CV_TY * CV_TX * out_cs; e += 256) {
        int ch = e % out_cs;
        int p = e / out_cs;
        int yy = y0 + (p >> 4), xx = x0 + (p & 15);
        if (yy < h && xx < w)
            out[(((long long)n * h + yy) * w + xx) * out_cs + ch] = ch < D ? s2[p * (D + 3) + ch] : 0.f;
    }
}

extern "C" int ss_cost_volume(const float* x1, const float* x2, float* out, int n, int h, int w, int c, int r,
                              int out_cs, void* stream) {
    if (!x1 || !x2 || !out || n <= 0 || h <= 0 || w <= 0 || c <= 0 || (c & 3)) return SS_ERR_ARG;
    int D = (2 * r + 1) * (2 * r + 1);
    if (out_cs < D) return SS_ERR_ARG;
    dim3 g(ss_cdiv(w, CV_TX), ss_cdiv(h, CV_TY), n);
    hipStream_t st = (hipStream_t)stream;
    if (r == 5) hipLaunchKernelGGL((cost_volume_kernel<5>), g, dim3(256), 0, st, x1, x2, out, h, w, c, out_cs);
    else if (r == 3) hipLaunchKernelGGL((cost_volume_kernel<3>), g, dim3(256), 0, st, x1, x2, out, h, w, c, out_cs);
    else return SS_ERR_UNSUPPORTED;
    return ss_launch_status();
}

// ------------------------------------------------------------------------------------------------
// CCL = L2 normalise -> per-pixel Gram D[p][k] = <n1[p], n2[k]> (fp32 MFMA, 1x1 conv with n2 as the
// filter bank) -> 3x3 patch correlation as 9 shifted sums of D -> softmax(10 x) over k -> expected shift.
// (the reference convolves with 690 3x3x256 filters, 2.19 GFLOP; the shifted-sum identity needs 0.24.)
__global__ void l2norm_kernel(const float* __restrict__ in, float* __restrict__ out, long long npix, int c) {
    long long pix = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    if (pix >= npix) return;
    const float* p = in + pix * c;
    float ss = 0.f;
    for (int i = lane; i < c; i += 64) ss = fmaf(p[i], p[i], ss);
    ss = ss_wave_sum(ss);
    float d = fmaxf(sqrtf(ss), 1e-12f);
    for (int i = lane; i < c; i += 64) out[pix * c + i] = p[i] / d;
}

// one wave per query position p
__global__ void ccl_softmax_kernel(const float* __restrict__ Dm, float* __restrict__ flow_nchw,
                                   float* __restrict__ flow_nhwc4, int h, int w, float scale) {
    const int P = h * w;
    int n = blockIdx.y;
    int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    if (p >= P) return;
    const float* Db = Dm + (long long)n * P * P;
    int py = p / w, pxx = p - py * w;
    float g[12];   // P <= 768
    float mx = -INFINITY;
#pragma unroll
    for (int cnt = 0; cnt < 12; ++cnt) {
        int k = lane + 64 * cnt;
        g[cnt] = -INFINITY;
        if (k >= P) continue;
        int ky = k / w, kx = k - ky * w;
        float s = 0.f;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                int qy = py + dy, qx = pxx + dx, ry = ky + dy, rx = kx + dx;
                if ((unsigned)qy < (unsigned)h && (unsigned)qx < (unsigned)w && (unsigned)ry < (unsigned)h &&
                    (unsigned)rx < (unsigned)w)
                    s += Db[(long long)(qy * w + qx) * P + (ry * w + rx)];
            }
        }
        s *= scale;
        g[cnt] = s;
        mx = fmaxf(mx, s);
    }
    mx = ss_wave_max(mx);
    float se = 0.f, sx = 0.f, sy = 0.f;
#pragma unroll
    for (int cnt = 0; cnt < 12; ++cnt) {
        int k = lane + 64 * cnt;
        if (k >= P) continue;
        int ky = k / w, kx = k - ky * w;
        float e = expf(g[cnt] - mx);
        se += e;
        sx = fmaf(e, (float)(kx - pxx), sx);
        sy = fmaf(e, (float)(ky - py), sy);
    }
    se = ss_wave_sum(se); sx = ss_wave_sum(sx); sy = ss_wave_sum(sy);
    if (lane == 0) {
        float fx = sx / se, fy = sy / se;
        if (flow_nchw) {
            flow_nchw[((long long)n * 2 + 0) * P + p] = fx;
            flow_nchw[((long long)n * 2 + 1) * P + p] = fy;
        }
        if (flow_nhwc4)
            *reinterpret_cast<float4*>(flow_nhwc4 + ((long long)n * P + p) * 4) = make_float4(fx, fy, 0.f, 0.f);
    }
}

extern "C" long long ss_ccl_workspace_floats(int n, int h, int w, int c) {
    long long P = (long long)h * w;
    return (long long)n * P * (2ll * c + P);
}

extern "C" int ss_ccl(const float* f1, const float* f2, float* flow_nchw, float* flow_nhwc4, int n, int h, int w,
                      int c, float softmax_scale, float* ws, void* stream) {
    if (!f1 || !f2 || !ws || (!flow_nchw && !flow_nhwc4) || n <= 0 || h <= 0 || w <= 0 || c <= 0 || (c & 3))
        return SS_ERR_ARG;
    int P = h * w;
    if (P > 768) return SS_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    float* n1 = ws;
    float* n2 = n1 + (long long)n * P * c;
    float* Dm = n2 + (long long)n * P * c;
    long long npix = (long long)n * P;
    hipLaunchKernelGGL(l2norm_kernel, dim3(ss_cdiv(npix, 4)), dim3(256), 0, st, f1, n1, npix, c);
    hipLaunchKernelGGL(l2norm_kernel, dim3(ss_cdiv(npix, 4)), dim3(256), 0, st, f2, n2, npix, c);
    int rc = ss_launch_status();
    if (rc) return rc;
    // D[p][k] = sum_c n1[p][c] n2[k][c]: "image" = n1 as a 1 x P strip, "filters" = n2 rows, one group per batch item
    rc = ss_conv_nhwc(n1, n2, nullptr, nullptr, Dm, 1, 1, 1, P, c, P, 1, 1, 1, 1, 0, 0, 0, 0, P, n, (long long)P * c,
                      (long long)P * c, (long long)P * P, nullptr, 0, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(ccl_softmax_kernel, dim3(ss_cdiv(P, 4), n), dim3(256), 0, st, (const float*)Dm, flow_nchw,
                       flow_nhwc4, h, w, softmax_scale);
    return ss_launch_status();
}
