// Metric harness on device (test_metric_ssd.py:444-482, 513-527).
//   * alignment PSNR / SSIM of the two masked LR warps, computed in fp64 like scikit-image 0.15
//     (compare_psnr / compare_ssim: 7x7 uniform window, K1 = 0.01, K2 = 0.03, sample covariance, the SSIM map is
//     averaged over the interior [3:-3, 3:-3] only, so the filter's border mode never matters, channel mean);
//   * stability score (7-tap path differences, weights 0.1 / 0.3 / 0.9) and distortion score (inter + intra grid
//     terms, maximum over frames), reproduced as the reference executes them on its 5-D mesh tensors.
#include "common.h"

__device__ __forceinline__ double block_sum(double v, double* red) {
    // 256 threads
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// w1, w2: [frames][4][h][w] (3 colour planes 0..255 + validity mask plane) from ss_tps_warp_mask_nchw.
// acc (fp64, zeroed by the caller): per frame [0] = sum of squared error, [1] = sum of SSIM over interior x channels
__global__ __launch_bounds__(256) void psnr_ssim_kernel(const float* __restrict__ w1, const float* __restrict__ w2,
                                                        double* __restrict__ acc, int h, int w) {
    __shared__ double red[4];
    const int f = blockIdx.z;
    const long long hw = (long long)h * w;
    const float* a = w1 + (long long)f * 4 * hw;
    const float* b = w2 + (long long)f * 4 * hw;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    double se = 0.0, ss = 0.0;
    if (x < w && y < h) {
        const bool interior = x >= 3 && x < w - 3 && y >= 3 && y < h - 3;
        const double C1 = (0.01 * 255.0) * (0.01 * 255.0), C2 = (0.03 * 255.0) * (0.03 * 255.0);
        const double cov_norm = 49.0 / 48.0;
        for (int ch = 0; ch < 3; ++ch) {
            const float* pa = a + ch * hw;
            const float* pb = b + ch * hw;
            {
                long long i = (long long)y * w + x;
                float ov = __fmul_rn(a[3 * hw + i], b[3 * hw + i]);
                double d = (double)__fmul_rn(pa[i], ov) - (double)__fmul_rn(pb[i], ov);
                se += d * d;
            }
            if (interior) {
                double sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
                for (int dy = -3; dy <= 3; ++dy) {
                    long long row = (long long)(y + dy) * w;
                    for (int dx = -3; dx <= 3; ++dx) {
                        long long i = row + x + dx;
                        float ov = __fmul_rn(a[3 * hw + i], b[3 * hw + i]);
                        double vx = (double)__fmul_rn(pa[i], ov), vy = (double)__fmul_rn(pb[i], ov);
                        sx += vx; sy += vy; sxx += vx * vx; syy += vy * vy; sxy += vx * vy;
                    }
                }
                double ux = sx / 49.0, uy = sy / 49.0, uxx = sxx / 49.0, uyy = syy / 49.0, uxy = sxy / 49.0;
                double vx = cov_norm * (uxx - ux * ux), vy = cov_norm * (uyy - uy * uy), vxy = cov_norm * (uxy - ux * uy);
                ss += ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux * ux + uy * uy + C1) * (vx + vy + C2));
            }
        }
    }
    se = block_sum(se, red);
    ss = block_sum(ss, red);
    if (threadIdx.x == 0) {
        atomicAdd(&acc[f * 2 + 0], se);
        atomicAdd(&acc[f * 2 + 1], ss);
    }
}

__global__ void psnr_ssim_finish_kernel(const double* __restrict__ acc, double* __restrict__ out, int frames, int h,
                                        int w) {
    int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= frames) return;
    double mse = acc[f * 2] / ((double)h * w * 3.0);
    out[f * 2 + 0] = 10.0 * log10(255.0 * 255.0 / mse);
    out[f * 2 + 1] = acc[f * 2 + 1] / ((double)(h - 6) * (w - 6) * 3.0);
}

extern "C" int ss_alignment_psnr_ssim(const float* w1, const float* w2, double* out, double* ws, int frames, int h,
                                      int w, void* stream) {
    if (!w1 || !w2 || !out || !ws || frames <= 0 || h < 7 || w < 7) return SS_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(ws, 0, sizeof(double) * 2 * frames, st) != hipSuccess) return SS_ERR_LAUNCH;
    hipLaunchKernelGGL(psnr_ssim_kernel, dim3(ss_cdiv(w, 64), ss_cdiv(h, 4), frames), dim3(256), 0, st, w1, w2, ws, h, w);
    hipLaunchKernelGGL(psnr_ssim_finish_kernel, dim3(ss_cdiv(frames, 64)), dim3(64), 0, st, (const double*)ws, out,
                       frames, h, w);
    return ss_launch_status();
}

// ---- stability (test_metric_ssd.py:459-468): path [T][63][2]; fp32 arithmetic, torch.mean over all elements
__global__ __launch_bounds__(256) void stability_kernel(const float* __restrict__ path, float* __restrict__ out, int t) {
    __shared__ double red[4];
    const int n = (t - 6) * SS_NV * 2;       // elements of one slice
    const int stride = SS_NV * 2;
    double s[3] = {0, 0, 0};                 // lag 3, 2, 1 (both sides)
    for (int e = threadIdx.x; e < n; e += 256) {
        float mid = path[e + 3 * stride];
        for (int lag = 1; lag <= 3; ++lag) {
            float dl = path[e + (3 - lag) * stride] - mid, dr = path[e + (3 + lag) * stride] - mid;
            s[3 - lag] += (double)fabsf(dl * dl) + (double)fabsf(dr * dr);
        }
    }
    double a3 = block_sum(s[0], red), a2 = block_sum(s[1], red), a1 = block_sum(s[2], red);
    if (threadIdx.x == 0) out[0] = (float)((a3 * 0.1 + a2 * 0.3 + a1 * 0.9) / (double)n);
}

extern "C" int ss_stability_score(const float* path, float* out, int t, void* stream) {
    if (!path || !out || t < 7) return SS_ERR_ARG;
    hipLaunchKernelGGL(stability_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, path, out, t);
    return ss_launch_status();
}

// ---- distortion (test_metric_ssd.py:38-87, 473-482): mesh [T][7][9][2]; per frame inter + intra, max over frames.
// inter_grid_loss on a [1,1,7,9,2] tensor reduces the edge products over dim 3 (vertex columns), see oracle/metrics.py.
__global__ __launch_bounds__(64) void distortion_kernel(const float* __restrict__ mesh, float* __restrict__ per_frame, int t) {
    int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= t) return;
    const float* m = mesh + (long long)f * SS_NV * 2;
    auto V = [&](int i, int j, int c) { return m[(i * 9 + j) * 2 + c]; };
    // horizontal: w_edges[i][j][c] = V(i,j) - V(i,j+1), j < 8; products of successive edges summed over j (7 terms)
    float dw_sum = 0.f;
    float dwa[7][2];
    for (int i = 0; i < 7; ++i)
        for (int c = 0; c < 2; ++c) {
            float ab = 0.f, aa = 0.f, bb = 0.f;
            for (int j = 0; j < 7; ++j) {
                float a = V(i, j, c) - V(i, j + 1, c), b = V(i, j + 1, c) - V(i, j + 2, c);
                ab += a * b; aa += a * a; bb += b * b;
            }
            dwa[i][c] = 1.f - ab / (sqrtf(aa) * sqrtf(bb));
        }
    for (int i = 0; i < 6; ++i)
        for (int c = 0; c < 2; ++c) dw_sum += dwa[i][c] + dwa[i + 1][c];
    float err_w = dw_sum / 12.f;
    // vertical: h_edges[i][j][c] = V(i,j) - V(i+1,j), i < 6; products of successive edges (i, i+1) summed over j (9)
    float dha[5][2];
    for (int i = 0; i < 5; ++i)
        for (int c = 0; c < 2; ++c) {
            float ab = 0.f, aa = 0.f, bb = 0.f;
            for (int j = 0; j < 9; ++j) {
                float a = V(i, j, c) - V(i + 1, j, c), b = V(i + 1, j, c) - V(i + 2, j, c);
                ab += a * b; aa += a * a; bb += b * b;
            }
            dha[i][c] = 1.f - ab / (sqrtf(aa) * sqrtf(bb));
        }
    // delta_h[..., 0:8] + delta_h[..., 1:9] on a last dim of size 2: [c0, c1] + [c1] (broadcast)
    float dh_sum = 0.f;
    for (int i = 0; i < 5; ++i) dh_sum += (dha[i][0] + dha[i][1]) + (dha[i][1] + dha[i][1]);
    float err_h = dh_sum / 10.f;
    // intra: relu(dx - 120), relu(dy - 120) means
    float sx = 0.f, sy = 0.f;
    for (int i = 0; i < 7; ++i)
        for (int j = 0; j < 8; ++j) sx += fmaxf(V(i, j + 1, 0) - V(i, j, 0) - 120.f, 0.f);
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 9; ++j) sy += fmaxf(V(i + 1, j, 1) - V(i, j, 1) - 120.f, 0.f);
    per_frame[f] = (err_w + err_h) + (sx / 56.f + sy / 54.f);
}

__global__ void max_reduce_kernel(const float* __restrict__ v, float* __restrict__ out, int n) {
    float m = -INFINITY;
    for (int i = threadIdx.x; i < n; i += 64) m = fmaxf(m, v[i]);
    m = ss_wave_max(m);
    if (threadIdx.x == 0) out[0] = m;
}

extern "C" int ss_distortion_score(const float* mesh, float* out, float* ws, int t, void* stream) {
    if (!mesh || !out || !ws || t <= 0) return SS_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(distortion_kernel, dim3(ss_cdiv(t, 64)), dim3(64), 0, st, mesh, ws, t);
    hipLaunchKernelGGL(max_reduce_kernel, dim3(1), dim3(64), 0, st, (const float*)ws, out, t);
    return ss_launch_status();
}
