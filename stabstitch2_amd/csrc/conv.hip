// Convolution / pooling / FC engine for gfx950 (CDNA4).
//
// conv: fp32 implicit GEMM on v_mfma_f32_32x32x2_f32 (exact fp32 fma chain, 157 TF peak).
//   GEMM view  D[m][co] = sum_k A[m][k] * Wt[co][k]
//     m  = (n, to, ho, wo)  output position          (rows, M = n*to*ho*wo)
//     k  = (dt, dh, dw, ci) filter tap x in-channel  (ci fastest, cin % 4 == 0)
//   Block = 256 threads = 2x2 waves, each wave owns WM x WN tiles of 32x32 (16 acc VGPRs each).
//   K is walked in tiles of 16: global -> registers (one float4 per thread per 64 rows, 4 lanes
//   cover one 64-byte run of a row) -> LDS stored k-major ([16][rows+4]) so that the MFMA
//   operand read (lane l: row l&31, k = 2j + (l>>5)) is a conflict-free ds_read_b32;
//   double-buffered LDS, one barrier per K tile, next tile's global loads in flight under the MFMAs.
//   Epilogue: bias (folded BatchNorm), residual, ReLU; for a fixed accumulator register the 32
//   lanes of a half-wave hold 32 consecutive output channels -> 128-byte coalesced NHWC stores.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvP {
    const float* in;
    const float* wgt;
    const float* bias;
    const float* res;
    float* out;
    int T, H, W, C;
    int To, Ho, Wo, Co;
    int kh, kw, s;
    int pt, ph, pw;
    int K, M;
    uint32_t mulC, mulKw, mulKh;
    int relu, out_cs;
    long long in_gs, w_gs, out_gs;
};

#define BK 16

template <int WM, int WN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvP p) {
    constexpr int BM = 64 * WM;
    constexpr int BN = 64 * WN;
    constexpr int LDA = BM + 4;
    constexpr int LDB = BN + 4;
    __shared__ float As[2][BK][LDA];
    __shared__ float Bs[2][BK][LDB];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const float* __restrict__ in = p.in + (long long)blockIdx.z * p.in_gs;
    const float* __restrict__ wgt = p.wgt + (long long)blockIdx.z * p.w_gs;

    const int lrow = tid >> 2;   // 0..63
    const int kq = tid & 3;      // which float4 of the 16-wide K tile

    // per-thread row bookkeeping (fixed over the K loop)
    int r_ti0[WM], r_hi0[WM], r_wi0[WM], r_nb[WM];
    bool r_ok[WM];
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        int m = m0 + lrow + i * 64;
        r_ok[i] = m < p.M;
        int mm = r_ok[i] ? m : 0;
        int wo = mm % p.Wo;
        int t1 = mm / p.Wo;
        int ho = t1 % p.Ho;
        int t2 = t1 / p.Ho;
        int to = t2 % p.To;
        int n = t2 / p.To;
        r_ti0[i] = to - p.pt;
        r_hi0[i] = ho * p.s - p.ph;
        r_wi0[i] = wo * p.s - p.pw;
        r_nb[i] = n * p.T;
    }
    long long w_off[WN];
    bool w_ok[WN];
#pragma unroll
    for (int i = 0; i < WN; ++i) {
        int co = n0 + lrow + i * 64;
        w_ok[i] = co < p.Co;
        w_off[i] = (long long)(w_ok[i] ? co : 0) * p.K;
    }

    float4 ra[WM], rb[WN];
    auto gload = [&](int kt) {
        int k = kt * BK + kq * 4;
        bool kok = k < p.K;
        uint32_t tap = ss_fastdiv((uint32_t)k, p.mulC);
        int ci = k - (int)tap * p.C;
        uint32_t t2 = ss_fastdiv(tap, p.mulKw);
        int dw = (int)tap - (int)t2 * p.kw;
        uint32_t dt = ss_fastdiv(t2, p.mulKh);
        int dh = (int)t2 - (int)dt * p.kh;
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            int ti = r_ti0[i] + (int)dt, hi = r_hi0[i] + dh, wi = r_wi0[i] + dw;
            bool ok = kok && r_ok[i] && (unsigned)ti < (unsigned)p.T && (unsigned)hi < (unsigned)p.H &&
                      (unsigned)wi < (unsigned)p.W;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) {
                long long a = ((((long long)(r_nb[i] + ti)) * p.H + hi) * p.W + wi) * p.C + ci;
                v = *reinterpret_cast<const float4*>(in + a);
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < WN; ++i) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kok && w_ok[i]) v = *reinterpret_cast<const float4*>(wgt + w_off[i] + k);
            rb[i] = v;
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            int r = lrow + i * 64;
            As[buf][kq * 4 + 0][r] = ra[i].x;
            As[buf][kq * 4 + 1][r] = ra[i].y;
            As[buf][kq * 4 + 2][r] = ra[i].z;
            As[buf][kq * 4 + 3][r] = ra[i].w;
        }
#pragma unroll
        for (int i = 0; i < WN; ++i) {
            int r = lrow + i * 64;
            Bs[buf][kq * 4 + 0][r] = rb[i].x;
            Bs[buf][kq * 4 + 1][r] = rb[i].y;
            Bs[buf][kq * 4 + 2][r] = rb[i].z;
            Bs[buf][kq * 4 + 3][r] = rb[i].w;
        }
    };

    f32x16 acc[WM][WN];
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int b = 0; b < WN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int nk = (p.K + BK - 1) / BK;
    const int li = lane & 31, lh = lane >> 5;
    const int arow = wr * 32 * WM + li;
    const int bcol = wc * 32 * WN + li;

    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);
#pragma unroll
        for (int j = 0; j < BK / 2; ++j) {
            float a[WM], b[WN];
#pragma unroll
            for (int x = 0; x < WM; ++x) a[x] = As[buf][2 * j + lh][arow + x * 32];
#pragma unroll
            for (int y = 0; y < WN; ++y) b[y] = Bs[buf][2 * j + lh][bcol + y * 32];
#pragma unroll
            for (int x = 0; x < WM; ++x)
#pragma unroll
                for (int y = 0; y < WN; ++y)
                    acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[x], b[y], acc[x][y], 0, 0, 0);
        }
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }

    // epilogue
    float* __restrict__ out = p.out + (long long)blockIdx.z * p.out_gs;
    const float* __restrict__ res = p.res ? p.res + (long long)blockIdx.z * p.out_gs : nullptr;
#pragma unroll
    for (int y = 0; y < WN; ++y) {
        int co = n0 + wc * 32 * WN + y * 32 + li;
        if (co >= p.Co) continue;
        float bias = p.bias ? p.bias[co] : 0.f;
#pragma unroll
        for (int x = 0; x < WM; ++x) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int m = m0 + wr * 32 * WM + x * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < p.M) {
                    long long o = (long long)m * p.out_cs + co;
                    float v = acc[x][y][r] + bias;
                    if (res) v += res[o];
                    if (p.relu) v = fmaxf(v, 0.f);
                    out[o] = v;
                }
            }
        }
    }
}

extern "C" int ss_conv_nhwc(const float* in, const float* wgt, const float* bias, const float* res, float* out,
                            int n, int t, int h, int w, int cin, int cout, int kt, int kh, int kw, int stride,
                            int pad_t, int pad_h, int pad_w, int relu, int out_cs, int groups, long long in_gs,
                            long long w_gs, long long out_gs, void* stream) {
    if (!in || !wgt || !out || n <= 0 || t <= 0 || h <= 0 || w <= 0 || cin <= 0 || (cin & 3) || cout <= 0 ||
        kt <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || groups <= 0 || out_cs < cout)
        return SS_ERR_ARG;
    ConvP p;
    p.in = in; p.wgt = wgt; p.bias = bias; p.res = res; p.out = out;
    p.T = t; p.H = h; p.W = w; p.C = cin;
    p.To = t + 2 * pad_t - kt + 1;
    p.Ho = (h + 2 * pad_h - kh) / stride + 1;
    p.Wo = (w + 2 * pad_w - kw) / stride + 1;
    p.Co = cout;
    if (p.To <= 0 || p.Ho <= 0 || p.Wo <= 0) return SS_ERR_ARG;
    p.kh = kh; p.kw = kw; p.s = stride; p.pt = pad_t; p.ph = pad_h; p.pw = pad_w;
    long long K = (long long)kt * kh * kw * cin;
    long long M = (long long)n * p.To * p.Ho * p.Wo;
    if (K >= 65536 || M >= (1ll << 31)) return SS_ERR_UNSUPPORTED;
    p.K = (int)K; p.M = (int)M;
    p.mulC = ss_fastdiv_magic((uint32_t)cin);
    p.mulKw = ss_fastdiv_magic((uint32_t)kw);
    p.mulKh = ss_fastdiv_magic((uint32_t)kh);
    p.relu = relu; p.out_cs = out_cs;
    p.in_gs = in_gs; p.w_gs = w_gs; p.out_gs = out_gs;
    hipStream_t st = (hipStream_t)stream;
    // tile choice: keep >= ~2 workgroups per CU where the problem allows it
    long long t22 = (long long)ss_cdiv(M, 128) * ss_cdiv(cout, 128) * groups;
    long long t21 = (long long)ss_cdiv(M, 128) * ss_cdiv(cout, 64) * groups;
    if (cout >= 128 && t22 >= 512) {
        dim3 g(ss_cdiv(M, 128), ss_cdiv(cout, 128), groups);
        hipLaunchKernelGGL((conv_igemm_kernel<2, 2>), g, dim3(256), 0, st, p);
    } else if (t21 >= 512) {
        dim3 g(ss_cdiv(M, 128), ss_cdiv(cout, 64), groups);
        hipLaunchKernelGGL((conv_igemm_kernel<2, 1>), g, dim3(256), 0, st, p);
    } else {
        dim3 g(ss_cdiv(M, 64), ss_cdiv(cout, 64), groups);
        hipLaunchKernelGGL((conv_igemm_kernel<1, 1>), g, dim3(256), 0, st, p);
    }
    return ss_launch_status();
}

// ------------------------------------------------------------------------------------------------
// max pooling, nhwc, floor mode, -inf padding
__global__ void maxpool_kernel(const float* __restrict__ in, float* __restrict__ out, int n, int h, int w, int c4,
                               int ho, int wo, int k, int s, int pad) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long total = (long long)n * ho * wo * c4;
    if (idx >= total) return;
    int cq = (int)(idx % c4);
    long long r = idx / c4;
    int x = (int)(r % wo);
    r /= wo;
    int y = (int)(r % ho);
    int b = (int)(r / ho);
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    for (int dy = 0; dy < k; ++dy) {
        int yy = y * s - pad + dy;
        if ((unsigned)yy >= (unsigned)h) continue;
        for (int dx = 0; dx < k; ++dx) {
            int xx = x * s - pad + dx;
            if ((unsigned)xx >= (unsigned)w) continue;
            float4 v = reinterpret_cast<const float4*>(in)[(((long long)b * h + yy) * w + xx) * c4 + cq];
            m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        }
    }
    reinterpret_cast<float4*>(out)[idx] = m;
}

extern "C" int ss_maxpool_nhwc(const float* in, float* out, int n, int h, int w, int c, int k, int stride, int pad,
                               void* stream) {
    if (!in || !out || n <= 0 || h <= 0 || w <= 0 || c <= 0 || (c & 3) || k <= 0 || stride <= 0) return SS_ERR_ARG;
    int ho = (h + 2 * pad - k) / stride + 1, wo = (w + 2 * pad - k) / stride + 1;
    if (ho <= 0 || wo <= 0) return SS_ERR_ARG;
    long long total = (long long)n * ho * wo * (c / 4);
    hipLaunchKernelGGL(maxpool_kernel, dim3(ss_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, in, out, n, h, w,
                       c / 4, ho, wo, k, stride, pad);
    return ss_launch_status();
}

// ------------------------------------------------------------------------------------------------
// FC: one wave per output neuron, weight row streamed once with 16-byte loads, up to MT batch rows
// accumulated per pass (x comes from L2).
template <int MT>
__global__ __launch_bounds__(256) void linear_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ b, float* __restrict__ y, int m,
                                                     int k, int nout, int relu) {
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int o = blockIdx.x * 4 + wave;
    if (o >= nout) return;
    const float* wr = w + (long long)o * k;
    int m0 = blockIdx.y * MT;
    float acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = 0.f;
    if ((k & 3) == 0) {
        int k4 = k >> 2;
        for (int q = lane; q < k4; q += 64) {
            float4 wv = reinterpret_cast<const float4*>(wr)[q];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                if (m0 + i < m) {
                    float4 xv = reinterpret_cast<const float4*>(x + (long long)(m0 + i) * k)[q];
                    acc[i] = fmaf(wv.x, xv.x, acc[i]);
                    acc[i] = fmaf(wv.y, xv.y, acc[i]);
                    acc[i] = fmaf(wv.z, xv.z, acc[i]);
                    acc[i] = fmaf(wv.w, xv.w, acc[i]);
                }
            }
        }
    } else {
        for (int q = lane; q < k; q += 64) {
            float wv = wr[q];
#pragma unroll
            for (int i = 0; i < MT; ++i)
                if (m0 + i < m) acc[i] = fmaf(wv, x[(long long)(m0 + i) * k + q], acc[i]);
        }
    }
    float bias = b ? b[o] : 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        float v = ss_wave_sum(acc[i]);
        if (lane == 0 && m0 + i < m) {
            v += bias;
            if (relu) v = fmaxf(v, 0.f);
            y[(long long)(m0 + i) * nout + o] = v;
        }
    }
}

extern "C" int ss_linear(const float* x, const float* w, const float* b, float* y, int m, int k, int nout, int relu,
                         void* stream) {
    if (!x || !w || !y || m <= 0 || k <= 0 || nout <= 0) return SS_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (m <= 1) {
        hipLaunchKernelGGL((linear_kernel<1>), dim3(ss_cdiv(nout, 4), 1), dim3(256), 0, st, x, w, b, y, m, k, nout, relu);
    } else if (m <= 4) {
        hipLaunchKernelGGL((linear_kernel<4>), dim3(ss_cdiv(nout, 4), 1), dim3(256), 0, st, x, w, b, y, m, k, nout, relu);
    } else {
        hipLaunchKernelGGL((linear_kernel<8>), dim3(ss_cdiv(nout, 4), ss_cdiv(m, 8)), dim3(256), 0, st, x, w, b, y, m, k,
                           nout, relu);
    }
    return ss_launch_status();
}

// ------------------------------------------------------------------------------------------------
// layout plumbing
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out, int n, int c, int h, int w,
                                    int cp) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long total = (long long)n * h * w * cp;
    if (idx >= total) return;
    int ch = (int)(idx % cp);
    long long pix = idx / cp;
    long long hw = (long long)h * w;
    int b = (int)(pix / hw);
    long long rem = pix - (long long)b * hw;
    out[idx] = ch < c ? in[((long long)b * c + ch) * hw + rem] : 0.f;
}

__global__ void nhwc_to_nchw_kernel(const float* __restrict__ in, float* __restrict__ out, int n, int c, int h, int w,
                                    int cs) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long hw = (long long)h * w;
    long long total = (long long)n * c * hw;
    if (idx >= total) return;
    long long rem = idx % hw;
    long long bc = idx / hw;
    int ch = (int)(bc % c);
    int b = (int)(bc / c);
    out[idx] = in[((long long)b * hw + rem) * cs + ch];
}

extern "C" int ss_nchw_to_nhwc(const float* in, float* out, int n, int c, int h, int w, int c_pad, void* stream) {
    if (!in || !out || n <= 0 || c <= 0 || h <= 0 || w <= 0 || c_pad < c) return SS_ERR_ARG;
    long long total = (long long)n * h * w * c_pad;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(ss_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, in, out, n, c,
                       h, w, c_pad);
    return ss_launch_status();
}

extern "C" int ss_nhwc_to_nchw(const float* in, float* out, int n, int c, int h, int w, int c_stride, void* stream) {
    if (!in || !out || n <= 0 || c <= 0 || h <= 0 || w <= 0 || c_stride < c) return SS_ERR_ARG;
    long long total = (long long)n * c * h * w;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(ss_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, in, out, n, c,
                       h, w, c_stride);
    return ss_launch_status();
}

extern "C" int ss_version(void) { return 100; }

extern "C" const char* ss_error_string(int code) {
    switch (code) {
        case SS_OK: return "ok";
        case SS_ERR_ARG: return "bad argument";
        case SS_ERR_LAUNCH: return "kernel launch failed";
        case SS_ERR_UNSUPPORTED: return "unsupported size";
        default: return "unknown error";
    }
}
