// Exact-math geometry kernels: 4-point DLT, bidirectional decomposition, H -> mesh, homography
// sampler, thin-plate-spline solve / point evaluation, tsmotion composition, canvas bbox.
// These are tiny (63-vertex meshes) but accuracy-critical: linear solves run in fp64 on device
// (the reference's fp32 torch.inverse of the 8x8 DLT system is itself +-0.02 px);
// everything the reference evaluates in fp32 (RBF kernel, normalisation, sampling) stays fp32
// with the same operation order (no fma contraction where the reference has separate ops).
#include "common.h"
#include "device_math.h"

// ------------------------------------------------------------------------------------------------
// small dense helpers (fp64, one thread)
// Every loop is unrolled and the row exchange is a chain of predicated moves, so all indices are compile-time constants and the
// 8 x 9 system stays in REGISTERS.  (With `A[piv][k]` indexed by a run-time row the array lived in scratch memory: one
// thread's pair of solves took 25-38 us -- the whole spatial_decompose / spatial_meshes launch.)  Same operations on the same
// values in the same order as the indexed form: bit-identical results.
__device__ __forceinline__ void solve8(double (&A)[8][9]) {   // in-place Gauss-Jordan with partial pivoting, solution in A[.][8]
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        int piv = c;
        double best = fabs(A[c][c]);
#pragma unroll
        for (int r = c + 1; r < 8; ++r) {
            const double v = fabs(A[r][c]);
            if (v > best) { best = v; piv = r; }
        }
#pragma unroll
        for (int r = c + 1; r < 8; ++r) {
            const bool sw = piv == r;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const double x = A[c][k], y = A[r][k];
                A[c][k] = sw ? y : x;
                A[r][k] = sw ? x : y;
            }
        }
        const double inv = 1.0 / A[c][c];
#pragma unroll
        for (int k = c; k < 9; ++k) A[c][k] *= inv;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (r == c) continue;
            const double f = A[r][c];
#pragma unroll
            for (int k = c; k < 9; ++k) A[r][k] -= f * A[c][k];
        }
    }
}

// homography src -> dst from 4 correspondences (rows as utils/torch_DLT.py:29-38)
__device__ void dlt4(const float sx[4], const float sy[4], const float dx[4], const float dy[4], double H[9]) {
    double A[8][9];
    for (int i = 0; i < 4; ++i) {
        double x = sx[i], y = sy[i], u = dx[i], v = dy[i];
        double* r0 = A[2 * i];
        double* r1 = A[2 * i + 1];
        r0[0] = x; r0[1] = y; r0[2] = 1; r0[3] = 0; r0[4] = 0; r0[5] = 0; r0[6] = -u * x; r0[7] = -u * y; r0[8] = u;
        r1[0] = 0; r1[1] = 0; r1[2] = 0; r1[3] = x; r1[4] = y; r1[5] = 1; r1[6] = -v * x; r1[7] = -v * y; r1[8] = v;
    }
    solve8(A);
    for (int i = 0; i < 8; ++i) H[i] = A[i][8];
    H[8] = 1.0;
}

__device__ void inv3(const double m[9], double o[9]) {
    double c0 = m[4] * m[8] - m[5] * m[7];
    double c1 = m[5] * m[6] - m[3] * m[8];
    double c2 = m[3] * m[7] - m[4] * m[6];
    double det = m[0] * c0 + m[1] * c1 + m[2] * c2;
    double id = 1.0 / det;
    o[0] = c0 * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[3] = c1 * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[6] = c2 * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

__device__ void mul3(const double a[9], const double b[9], double o[9]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) o[i * 3 + j] = a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j] + a[i * 3 + 2] * b[6 + j];
}

// H, H_tgt, H_ref for one offset (spatial_network.py:72-93 / :291-300), corner points / scale in fp32
__device__ void decompose(const float* off, float img_h, float img_w, float scale, double H[9], double Ht[9],
                          double Hr[9]) {
    const float cx[4] = {0.f, img_w, 0.f, img_w};
    const float cy[4] = {0.f, 0.f, img_h, img_h};
    float sx[4], sy[4], dx[4], dy[4], hx[4], hy[4];
    for (int i = 0; i < 4; ++i) {
        float mx = off[2 * i], my = off[2 * i + 1];
        sx[i] = cx[i] / scale;
        sy[i] = cy[i] / scale;
        dx[i] = __fadd_rn(cx[i], mx) / scale;
        dy[i] = __fadd_rn(cy[i], my) / scale;
        hx[i] = __fadd_rn(cx[i], mx / 2.f) / scale;
        hy[i] = __fadd_rn(cy[i], my / 2.f) / scale;
    }
    dlt4(sx, sy, dx, dy, H);
    dlt4(sx, sy, hx, hy, Ht);
    double Hi[9];
    inv3(H, Hi);
    mul3(Hi, Ht, Hr);
}

__global__ __launch_bounds__(64) void tensor_dlt_kernel(const float* __restrict__ src, const float* __restrict__ dst, float* __restrict__ H,
                                  int n) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n) return;
    float sx[4], sy[4], dx[4], dy[4];
    for (int i = 0; i < 4; ++i) {
        sx[i] = src[b * 8 + 2 * i]; sy[i] = src[b * 8 + 2 * i + 1];
        dx[i] = dst[b * 8 + 2 * i]; dy[i] = dst[b * 8 + 2 * i + 1];
    }
    double h[9];
    dlt4(sx, sy, dx, dy, h);
    for (int i = 0; i < 9; ++i) H[b * 9 + i] = (float)h[i];
}

extern "C" int ss_tensor_dlt(const float* src, const float* dst, float* H, int n, void* stream) {
    if (!src || !dst || !H || n <= 0) return SS_ERR_ARG;
    hipLaunchKernelGGL(tensor_dlt_kernel, dim3(ss_cdiv(n, 64)), dim3(64), 0, (hipStream_t)stream, src, dst, H, n);
    return ss_launch_status();
}

__global__ __launch_bounds__(64) void spatial_decompose_kernel(const float* __restrict__ off, float* __restrict__ th_ref,
                                         float* __restrict__ th_tgt, int n, float img_h, float img_w) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n) return;
    double H[9], Ht[9], Hr[9];
    decompose(off + b * 8, img_h, img_w, 8.f, H, Ht, Hr);
    // theta = M^-1 H M,  M = [[w/2,0,w/2],[0,h/2,h/2],[0,0,1]] at feature size (w,h) = (img_w/8, img_h/8)
    double fw = (double)(img_w / 8.f / 2.f), fh = (double)(img_h / 8.f / 2.f);
    double M[9] = {fw, 0, fw, 0, fh, fh, 0, 0, 1};
    double Mi[9] = {1.0 / fw, 0, -1.0, 0, 1.0 / fh, -1.0, 0, 0, 1};
    double t[9], o[9];
    mul3(Mi, Hr, t); mul3(t, M, o);
    for (int i = 0; i < 9; ++i) th_ref[b * 9 + i] = (float)o[i];
    mul3(Mi, Ht, t); mul3(t, M, o);
    for (int i = 0; i < 9; ++i) th_tgt[b * 9 + i] = (float)o[i];
}

extern "C" int ss_spatial_decompose(const float* offset8, float* theta_ref, float* theta_tgt, int n, float img_h,
                                    float img_w, void* stream) {
    if (!offset8 || !theta_ref || !theta_tgt || n <= 0) return SS_ERR_ARG;
    hipLaunchKernelGGL(spatial_decompose_kernel, dim3(ss_cdiv(n, 64)), dim3(64), 0, (hipStream_t)stream, offset8,
                       theta_ref, theta_tgt, n, img_h, img_w);
    return ss_launch_status();
}

__device__ __forceinline__ void rigid_vertex(int v, float img_h, float img_w, float& x, float& y) {
    int i = v / (SS_GRID_W + 1), j = v - i * (SS_GRID_W + 1);
    x = linspace_at(0.f, img_w, SS_GRID_W + 1, j);
    y = linspace_at(0.f, img_h, SS_GRID_H + 1, i);
}

// one block (64 threads) per batch item; thread v < 63 = vertex
__global__ __launch_bounds__(64) void spatial_meshes_kernel(const float* __restrict__ off, const float* __restrict__ off_ref,
                                      const float* __restrict__ off_tgt, float* __restrict__ motion1,
                                      float* __restrict__ motion2, float img_h, float img_w) {
    __shared__ double sHr[9], sHt[9];
    int b = blockIdx.x, v = threadIdx.x;
    if (v == 0) {
        double H[9], Ht[9], Hr[9], inv[9];
        decompose(off + b * 8, img_h, img_w, 1.f, H, Ht, Hr);
        inv3(Hr, inv);
        for (int i = 0; i < 9; ++i) sHr[i] = inv[i];
        inv3(Ht, inv);
        for (int i = 0; i < 9; ++i) sHt[i] = inv[i];
    }
    __syncthreads();
    if (v >= SS_NV) return;
    float rx, ry;
    rigid_vertex(v, img_h, img_w, rx, ry);
    for (int which = 0; which < 2; ++which) {
        const double* Hi = which ? sHt : sHr;
        double X = Hi[0] * rx + Hi[1] * ry + Hi[2];
        double Y = Hi[3] * rx + Hi[4] * ry + Hi[5];
        double Z = Hi[6] * rx + Hi[7] * ry + Hi[8];
        float mx = (float)(X / Z), my = (float)(Y / Z);
        const float* o2 = (which ? off_tgt : off_ref) + (long long)b * 126 + v * 2;
        float* out = (which ? motion2 : motion1) + (long long)b * 126 + v * 2;
        out[0] = __fsub_rn(__fadd_rn(mx, o2[0]), rx);
        out[1] = __fsub_rn(__fadd_rn(my, o2[1]), ry);
    }
}

extern "C" int ss_spatial_meshes(const float* offset8, const float* off_ref, const float* off_tgt, float* motion1,
                                 float* motion2, int n, float img_h, float img_w, void* stream) {
    if (!offset8 || !off_ref || !off_tgt || !motion1 || !motion2 || n <= 0) return SS_ERR_ARG;
    hipLaunchKernelGGL(spatial_meshes_kernel, dim3(n), dim3(64), 0, (hipStream_t)stream, offset8, off_ref, off_tgt,
                       motion1, motion2, img_h, img_w);
    return ss_launch_status();
}

// ------------------------------------------------------------------------------------------------
template <bool NHWC>
__global__ void homo_warp_kernel(const float* __restrict__ in, const float* __restrict__ theta,
                                 float* __restrict__ out, int n, int c, int h, int w, int oh, int ow, int split, int shift) {
    // NHWC: thread = (pixel, channel quad); NCHW: thread = (pixel), loops channels
    // output image b samples input image b + (b >= split ? shift : 0): two batches whose inputs overlap in memory as one launch
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int cq = NHWC ? c / 4 : 1;
    long long total = (long long)n * oh * ow * cq;
    if (idx >= total) return;
    int q = (int)(idx % cq);
    long long pix = idx / cq;
    int x = (int)(pix % ow);
    long long r2 = pix / ow;
    int y = (int)(r2 % oh);
    int b = (int)(r2 / oh);
    const float* th = theta + b * 9;
    float gx = linspace_at(-1.f, 1.f, ow, x), gy = linspace_at(-1.f, 1.f, oh, y);
    float xs = fmaf(th[2], 1.f, fmaf(th[1], gy, th[0] * gx));
    float ys = fmaf(th[5], 1.f, fmaf(th[4], gy, th[3] * gx));
    float ts = fmaf(th[8], 1.f, fmaf(th[7], gy, th[6] * gx));
    if (!(fabsf(ts) >= 1e-7f)) ts = __fadd_rn(ts, 1e-6f);
    float xn = xs / ts, yn = ys / ts;
    SsTaps t = taps_normal(xn, yn, w, h);
    if (NHWC) {
        const int ib = b + (b >= split ? shift : 0);
        const float4* base = reinterpret_cast<const float4*>(in) + (long long)ib * h * w * cq;
        float4 a = base[((long long)t.y0 * w + t.x0) * cq + q];
        float4 bb = base[((long long)t.y1 * w + t.x0) * cq + q];
        float4 cc = base[((long long)t.y0 * w + t.x1) * cq + q];
        float4 d = base[((long long)t.y1 * w + t.x1) * cq + q];
        float4 o;
        o.x = blend4(t, a.x, bb.x, cc.x, d.x);
        o.y = blend4(t, a.y, bb.y, cc.y, d.y);
        o.z = blend4(t, a.z, bb.z, cc.z, d.z);
        o.w = blend4(t, a.w, bb.w, cc.w, d.w);
        reinterpret_cast<float4*>(out)[idx] = o;
    } else {
        long long hw = (long long)h * w, ohw = (long long)oh * ow;
        for (int ch = 0; ch < c; ++ch) {
            const float* pl = in + ((long long)(b + (b >= split ? shift : 0)) * c + ch) * hw;
            float v = blend4(t, pl[(long long)t.y0 * w + t.x0], pl[(long long)t.y1 * w + t.x0],
                             pl[(long long)t.y0 * w + t.x1], pl[(long long)t.y1 * w + t.x1]);
            out[((long long)b * c + ch) * ohw + (long long)y * ow + x] = v;
        }
    }
}

extern "C" int ss_homo_warp_nhwc(const float* in, const float* theta, float* out, int n, int h, int w, int c,
                                 int out_h, int out_w, void* stream) {
    if (!in || !theta || !out || n <= 0 || h <= 0 || w <= 0 || c <= 0 || (c & 3) || out_h < 2 || out_w < 2)
        return SS_ERR_ARG;
    long long total = (long long)n * out_h * out_w * (c / 4);
    hipLaunchKernelGGL((homo_warp_kernel<true>), dim3(ss_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, in,
                       theta, out, n, c, h, w, out_h, out_w, n, 0);
    return ss_launch_status();
}

// (warp(in1, theta[0:n]), warp(in2, theta[n:2n])) -> out [2n] as ONE launch; in1 and in2 must be images of ONE tensor a whole
// number of images apart (they may overlap: a chain of pairs (view 1, view 2), (view 2, view 3) reads views [0:n] and [1:n+1])
extern "C" int ss_homo_warp_pair_nhwc(const float* in1, const float* in2, const float* theta, float* out, int n, int h, int w, int c,
                                      int out_h, int out_w, void* stream) {
    if (!in1 || !in2 || !theta || !out || n <= 0 || h <= 0 || w <= 0 || c <= 0 || (c & 3) || out_h < 2 || out_w < 2) return SS_ERR_ARG;
    const long long img = (long long)h * w * c, d = in2 - in1;
    if (d < 0 || d % img != 0 || d / img > (1 << 20)) return SS_ERR_ARG;
    const long long total = 2ll * n * out_h * out_w * (c / 4);
    hipLaunchKernelGGL((homo_warp_kernel<true>), dim3(ss_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, in1,
                       theta, out, 2 * n, c, h, w, out_h, out_w, n, (int)(d / img) - n);
    return ss_launch_status();
}

extern "C" int ss_homo_warp_nchw(const float* in, const float* theta, float* out, int n, int c, int h, int w,
                                 int out_h, int out_w, void* stream) {
    if (!in || !theta || !out || n <= 0 || h <= 0 || w <= 0 || c <= 0 || out_h < 2 || out_w < 2) return SS_ERR_ARG;
    long long total = (long long)n * out_h * out_w;
    hipLaunchKernelGGL((homo_warp_kernel<false>), dim3(ss_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, in,
                       theta, out, n, c, h, w, out_h, out_w, n, 0);
    return ss_launch_status();
}

// ------------------------------------------------------------------------------------------------
// TPS: 66x66 system [[P,R],[0,P^T]] assembled in fp32 as the reference does, solved in fp64.
#define TPS_LD 68
#define TPS_TQ 17        // columns per thread: 4 x 17 = 66 system columns + 2 right-hand sides
// Round 6: one workgroup of TPS_NW waves per system; wave q owns columns TPS_CPW q .. TPS_CPW q + TPS_CPW - 1 of ALL rows, lane l
// holds row l (`lo`) and lanes 0, 1 rows 64, 65 as well (`hi`).  Gauss-Jordan with the pivot rule and the arithmetic of the round-4
// kernel below (same pivots, same factors; the update is an fma now), but a step costs ONE barrier, no row travels through LDS, and
// the pivot search runs under the previous step's update:
//   * the wave that owns the NEXT column updates that column first, finds its pivot (keys of 66 rows: 4 DPP rotations + 4
//     readlanes), divides once and publishes the 66 factors f = A[r][col] / pivot lane-aligned (8 bytes per lane, double-buffered by
//     step parity) with the pivot's row index -- and only then updates its other columns, while the other waves are still busy with
//     theirs;
//   * behind the barrier every wave fetches ITS entries of the pivot row from its own registers (`v_readlane`, the lane is
//     wave-uniform) and updates the columns that are still live: none in the waves left of the pivot column, the ones right of it
//     in the owner (columns left of the pivot are unit columns: never read again).
// Round 4 (thread = (row, column quarter), 5 waves): key reduction -> barrier -> pivot row + 1 / pivot through LDS -> barrier ->
// update: 47.5 us per launch back to back, 49.6 us cold; this form 39.9 / 44.1 us (tools/ab_tps_solve.py); LAB_NOTES R6.6 has the
// anatomy and why it is not 2x: broadcasting a pivot-row entry costs as much as five fp64 fmas, whichever way it travels.
// src_stride = 0 shares one source mesh across the batch.
#ifndef TPS_NW
#define TPS_NW 4
#endif
#define TPS_CPW ((68 + TPS_NW - 1) / TPS_NW)           // columns per wave: 66 system columns + 2 right-hand sides (+ padding)
#define TPS_RHS (SS_NT - (TPS_NW - 1) * TPS_CPW)       // local index of the first right-hand side in the last wave

__device__ __forceinline__ double tps_entry(int r, int c, const float* sx, const float* sy, const float* tgt) {
    if (c >= SS_NT + 2) return 0.0;
    if (r < SS_NV) {
        if (c == 0) return 1.0;
        if (c == 1) return (double)sx[r];
        if (c == 2) return (double)sy[r];
        if (c < SS_NT) {
            // fp32 kernel entries like the reference, but with a correctly rounded log (via fp64): the
            // 66x66 system amplifies last-bit differences of logf ~100x into T (measured 3e-5 on |T|<2)
            float dx = __fsub_rn(sx[r], sx[c - 3]), dy = __fsub_rn(sy[r], sy[c - 3]);
            float d2 = __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));
            return (double)__fmul_rn(d2, (float)log((double)__fadd_rn(d2, 1e-6f)));
        }
        return (double)tgt[r * 2 + (c - SS_NT)];
    }
    return (c >= 3 && c < SS_NT) ? 1.0 : 0.0;            // row 63: the ones row of P^T
}

__device__ __forceinline__ double tps_readlane(double v, int l) {
    const long long bits = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)bits, l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(bits >> 32), l);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

__device__ __forceinline__ unsigned tps_key(double v, int r) {
    // top 25 bits of the fp64 magnitude (11 exponent + 14 mantissa bits: monotone in |v| over the whole range) above 127 - row:
    // a maximum carries its row.  RELAXED partial pivoting: the largest candidate up to a relative 2^-14, the lowest row among
    // candidates that close; tests/test_gpu_round4.py checks residuals on near-degenerate control points against an fp64 solve
    return ((unsigned)((unsigned long long)__double_as_longlong(fabs(v)) >> 38) << 7) | (unsigned)(127 - r);
}

struct TpsShared {
    float sx[SS_NV], sy[SS_NV];
    double fx[2][2][64];          // [step parity][lo / hi][lane]
    int piv[2];                   // pivot lane | 64 when the pivot row sits in `hi` of that lane (the waves swap it into `lo`)
    double diag[SS_NT];           // pivot value by COLUMN
};

// pivot of column `col`, held in (cl, ch), among the unused rows; publishes the factors (as they are AFTER the pivot lane's
// lo / hi exchange, if there is one), the pivot lane and the pivot value for step parity `slot`, and hands them to the caller
// (the wave that searches a column is the one that uses the result first: no LDS round trip on its own chain).
// The reciprocals of BOTH candidates of every lane are formed beside the key reduction (the division's ~15 dependent fp64
// instructions are the longest chain of the search; forming only `lo`'s and dividing again for rows 64, 65 measured slower);
// the pivot's is fetched once the lane is known.
__device__ __forceinline__ void tps_search(TpsShared& sh, int slot, int col, int lane, double cl, double ch, int rid_lo, int rid_hi,
                                           bool used_lo, bool used_hi, int& pcode, double& f0, double& f1) {
    const double r0 = 1.0 / cl, r1 = 1.0 / ch;
    unsigned key = max(used_lo ? 0u : tps_key(cl, rid_lo), used_hi ? 0u : tps_key(ch, rid_hi));
    key = max(key, (unsigned)__builtin_amdgcn_update_dpp(0, (int)key, 0x121, 0xF, 0xF, false));   // row_ror:1
    key = max(key, (unsigned)__builtin_amdgcn_update_dpp(0, (int)key, 0x122, 0xF, 0xF, false));   // row_ror:2
    key = max(key, (unsigned)__builtin_amdgcn_update_dpp(0, (int)key, 0x124, 0xF, 0xF, false));   // row_ror:4
    key = max(key, (unsigned)__builtin_amdgcn_update_dpp(0, (int)key, 0x128, 0xF, 0xF, false));   // row_ror:8
    const unsigned k0 = __builtin_amdgcn_readlane((int)key, 0), k1 = __builtin_amdgcn_readlane((int)key, 16);
    const unsigned k2 = __builtin_amdgcn_readlane((int)key, 32), k3 = __builtin_amdgcn_readlane((int)key, 48);
    const int prow = 127 - (int)(max(max(k0, k1), max(k2, k3)) & 127u);         // the pivot's ROW
    // the lane that holds it: rows 0..63 start in lane = row, rows 64, 65 in lanes 0, 1 -- and they only ever trade places
    // with the other row of their lane
    const unsigned long long in_lo = __ballot(rid_lo == prow), in_hi = __ballot(rid_hi == prow);
    const bool inhi = in_lo == 0ull;
    const int pl = __builtin_ctzll(inhi ? in_hi : in_lo);
    const double pinv = tps_readlane(inhi ? r1 : r0, pl);
    f0 = cl * pinv;
    f1 = ch * pinv;
    if (lane == pl) {
        sh.diag[col] = inhi ? ch : cl;
        if (inhi) f1 = f0;
        f0 = 0.0;                                                 // the pivot row (in `lo` after the exchange) stays as it is
    }
    if (lane >= 2) f1 = 0.0;
    pcode = pl | (inhi ? 64 : 0);
    sh.fx[slot][0][lane] = f0;
    if (lane < 2) sh.fx[slot][1][lane] = f1;
    if (lane == 0) sh.piv[slot] = pcode;
}

// The elimination of one system by the 64 * TPS_NW threads of a workgroup: control points in sh.sx / sh.sy (filled and
// synchronised by the caller), targets tgt [63][2], coefficients to t [2][66] (either may live in LDS: generic pointers).  Ends
// behind a barrier-free tail: the caller synchronises before it reads t or reuses `sh`.
__device__ __forceinline__ void tps_eliminate(TpsShared& sh, const float* tgt, float* t, int lane, int q) {
    double lo[TPS_CPW], hi[TPS_CPW];
#pragma unroll
    for (int j = 0; j < TPS_CPW; ++j) {
        const int c = q * TPS_CPW + j;
        lo[j] = tps_entry(lane, c, sh.sx, sh.sy, tgt);
        double bottom = 0.0;                  // rows 64, 65 = the x / y rows of P^T
        if (lane < 2 && c >= 3 && c < SS_NT) bottom = (double)(lane == 0 ? sh.sx[c - 3] : sh.sy[c - 3]);
        hi[j] = bottom;
    }
    int rid_lo = lane, rid_hi = 64 + lane;
    bool used_lo = false, used_hi = lane >= 2;
    int col_lo = 0, col_hi = 0;
    int my_pcode = 0;                    // the result of this wave's last search
    double my_f0 = 0.0, my_f1 = 0.0;
    if (q == 0) tps_search(sh, 0, 0, lane, lo[0], hi[0], rid_lo, rid_hi, used_lo, used_hi, my_pcode, my_f0, my_f1);
#define TPS_UPD(jj_) do { const double pr_ = tps_readlane(lo[jj_], pl); \
                          lo[jj_] = fma(-f0, pr_, lo[jj_]); hi[jj_] = fma(-f1, pr_, hi[jj_]); } while (0)
    for (int qq = 0; qq < TPS_NW; ++qq) {         // rolled on purpose (all 66 steps in one straight-line body: instruction-cache bound)
#pragma unroll
        for (int j = 0; j < TPS_CPW; ++j) {
            const int col = qq * TPS_CPW + j;
            if (col < SS_NT) {
                const int slot = col & 1;
                __syncthreads();
                int pcode;
                double f0, f1;
                if (q == qq) { pcode = my_pcode; f0 = my_f0; f1 = my_f1; }      // this wave searched column `col` itself
                else {
                    pcode = __builtin_amdgcn_readfirstlane(sh.piv[slot]);
                    f0 = sh.fx[slot][0][lane];
                    f1 = lane < 2 ? sh.fx[slot][1][lane & 1] : 0.0;
                }
                const int pl = pcode & 63;
                if (pcode & 64) {          // rows 64, 65 become pivots once each: they move into `lo` of their lane
                    if (lane == pl) {
#pragma unroll
                        for (int jj = 0; jj < TPS_CPW; ++jj) { const double t_ = lo[jj]; lo[jj] = hi[jj]; hi[jj] = t_; }
                        const int r_ = rid_lo; rid_lo = rid_hi; rid_hi = r_;
                        const bool u_ = used_lo; used_lo = used_hi; used_hi = u_;
                        const int c_ = col_lo; col_lo = col_hi; col_hi = c_;
                    }
                }
                if (lane == pl) { used_lo = true; col_lo = col; }
                if (j + 1 < TPS_CPW) {
                    // the next pivot column is this wave's column j + 1 (the owner) or nobody's business yet
                    if (q == qq) {
                        const int jn = j + 1 < TPS_CPW ? j + 1 : 0;
                        TPS_UPD(jn);
                        if (col + 1 < SS_NT) tps_search(sh, slot ^ 1, col + 1, lane, lo[jn], hi[jn], rid_lo, rid_hi, used_lo, used_hi, my_pcode, my_f0, my_f1);
#pragma unroll
                        for (int jj = 0; jj < TPS_CPW; ++jj) if (jj >= j + 2) TPS_UPD(jj);
                    } else if (q > qq) {
#pragma unroll
                        for (int jj = 0; jj < TPS_CPW; ++jj) TPS_UPD(jj);
                    }
                } else {
                    // last column of wave qq: wave qq + 1 owns the next one
                    if (q == qq + 1) {
                        TPS_UPD(0);
                        if (col + 1 < SS_NT) tps_search(sh, slot ^ 1, col + 1, lane, lo[0], hi[0], rid_lo, rid_hi, used_lo, used_hi, my_pcode, my_f0, my_f1);
#pragma unroll
                        for (int jj = 0; jj < TPS_CPW; ++jj) if (jj >= 1) TPS_UPD(jj);
                    } else if (q > qq) {
#pragma unroll
                        for (int jj = 0; jj < TPS_CPW; ++jj) TPS_UPD(jj);
                    }
                }
            }
        }
    }
#undef TPS_UPD
    __syncthreads();
    if (q == TPS_NW - 1) {     // right-hand sides are columns 66, 67
        const double d0 = sh.diag[col_lo];
        t[col_lo] = (float)(lo[TPS_RHS] / d0);
        t[SS_NT + col_lo] = (float)(lo[TPS_RHS + 1] / d0);
        if (lane < 2) {
            const double d1 = sh.diag[col_hi];
            t[col_hi] = (float)(hi[TPS_RHS] / d1);
            t[SS_NT + col_hi] = (float)(hi[TPS_RHS + 1] / d1);
        }
    }
}

__global__ __launch_bounds__(64 * TPS_NW) void tps_solve_kernel(const float* __restrict__ source, long long src_stride,
                                                               const float* __restrict__ target, long long tgt_stride,
                                                               float* __restrict__ T) {
    __shared__ TpsShared sh;
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* src = source + (long long)b * src_stride;
    if (tid < SS_NV) { sh.sx[tid] = src[tid * 2]; sh.sy[tid] = src[tid * 2 + 1]; }
    __syncthreads();
    tps_eliminate(sh, target + (long long)b * tgt_stride, T + (long long)b * 2 * SS_NT, tid & 63, __builtin_amdgcn_readfirstlane(tid >> 6));
}

#ifdef SS_TUNING
// The round-4 kernel (A/B and bit-identity checks: tools/ab_tps_solve.py; tuning build only).
// One workgroup of 320 threads per system, the augmented 66x68 matrix lives in REGISTERS: thread (r, q) = (tid>>2,
// tid&3) owns columns 17q..17q+16 of row r.  Gauss-Jordan with partial pivoting and no physical row swaps (a used-row
// flag instead); per column: pivot = the unused row with the largest |A[r][col]| (wave-level DPP reduction, see below) ->
// pivot row (and 1 / pivot) broadcast through LDS -> every row subtracts f * pivot_row with f fetched from its 4-lane
// row group by a shuffle.  All register indices are static (steps unrolled per 17-column quarter); two barriers per
// step.  src_stride = 0 shares one source mesh across the batch.
__global__ __launch_bounds__(320) void tps_solve_r4_kernel(const float* __restrict__ source, long long src_stride,
                                                        const float* __restrict__ target, long long tgt_stride,
                                                        float* __restrict__ T) {
    __shared__ float sx[SS_NV], sy[SS_NV];
    __shared__ double prow[TPS_LD], diag[SS_NT];
    __shared__ unsigned wkey[8];
    __shared__ double s_pinv;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int r = tid >> 2, q = tid & 3;
    const bool rowok = r < SS_NT;
    const float* src = source + (long long)b * src_stride;
    const float* tgt = target + (long long)b * tgt_stride;
    if (tid < SS_NV) { sx[tid] = src[tid * 2]; sy[tid] = src[tid * 2 + 1]; }
    __syncthreads();
    double a[TPS_TQ];
#pragma unroll
    for (int j = 0; j < TPS_TQ; ++j) {
        const int c = q * TPS_TQ + j;
        double v = 0.0;
        if (rowok) {
            if (r < SS_NV) {
                if (c == 0) v = 1.0;
                else if (c == 1) v = sx[r];
                else if (c == 2) v = sy[r];
                else if (c < SS_NT) {
                    // fp32 kernel entries like the reference, but with a correctly rounded log (via fp64): the
                    // 66x66 system amplifies last-bit differences of logf ~100x into T (measured 3e-5 on |T|<2)
                    float dx = __fsub_rn(sx[r], sx[c - 3]), dy = __fsub_rn(sy[r], sy[c - 3]);
                    float d2 = __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));
                    v = (double)__fmul_rn(d2, (float)log((double)__fadd_rn(d2, 1e-6f)));
                } else v = tgt[r * 2 + (c - SS_NT)];
            } else if (c >= 3 && c < SS_NT) {
                const int k = r - SS_NV;
                v = k == 0 ? 1.0 : (k == 1 ? (double)sx[c - 3] : (double)sy[c - 3]);
            }
        }
        a[j] = v;
    }
    bool used = false;
    int mycol = 0;
    // Pivot search without a barrier of its own (tools/micro/tps_solve_probe.hip has the step's anatomy): a 32-bit key = the
    // top 25 bits of the fp64 magnitude (all 11 exponent bits + 14 mantissa bits: monotone in |a| over the whole fp64 range,
    // nothing underflows) above 127 - row (0 for used rows), so a maximum carries its row.  RELAXED partial pivoting: the
    // pivot is the largest candidate up to a relative 2^-14, the lowest row among candidates that close (an exact arg-max is
    // not what elimination needs -- the growth bound moves by that factor); tests/test_gpu_round4.py checks residuals on
    // near-degenerate control points against an fp64 solve.  The candidates of a wave sit in lanes 4 i + qq: two
    // DPP row rotations + four readlanes give the wave's maximum, one lane publishes it, and after the barrier every thread
    // takes the largest of the five.  1360 clocks per column; the history: LDS-resident matrix 196 us per launch; registers +
    // arg-max by 6 rounds of 64-bit shuffles in wave 0 between two extra barriers 94 us; one 64-bit LDS atomic max per row
    // 85 us (66 lanes on one address: 1700 of a column's 2740 clocks); this form 42 us.  Publishing every wave's candidate
    // row speculatively to save the second barrier was measured slower (1760 clocks per column).
    for (int qq = 0; qq < 4; ++qq) {         // rolled on purpose (all 66 steps in one straight-line body: instruction-cache bound)
#pragma unroll
        for (int j = 0; j < TPS_TQ; ++j) {
            const int col = qq * TPS_TQ + j;
            if (col < SS_NT) {
                unsigned key = 0u;
                if (rowok && q == qq && !used)
                    key = ((unsigned)((unsigned long long)__double_as_longlong(fabs(a[j])) >> 38) << 7) | (unsigned)(127 - r);
                key = max(key, (unsigned)__builtin_amdgcn_update_dpp(0, (int)key, 0x124, 0xF, 0xF, false));   // row_ror:4
                key = max(key, (unsigned)__builtin_amdgcn_update_dpp(0, (int)key, 0x128, 0xF, 0xF, false));   // row_ror:8
                {
                    const unsigned k0 = __builtin_amdgcn_readlane((int)key, qq), k1 = __builtin_amdgcn_readlane((int)key, 16 + qq);
                    const unsigned k2 = __builtin_amdgcn_readlane((int)key, 32 + qq), k3 = __builtin_amdgcn_readlane((int)key, 48 + qq);
                    if ((tid & 63) == 0) wkey[tid >> 6] = max(max(k0, k1), max(k2, k3));      // (everyone read the last column's keys before its second barrier)
                }
                // A[r][col] sits in lane (row group, qq); fetched before the barrier, it does not depend on the pivot
                const double arc = __shfl(a[j], (tid & 60) | qq, 64);
                __syncthreads();
                const int piv = 127 - (int)(max(max(max(wkey[0], wkey[1]), max(wkey[2], wkey[3])), wkey[4]) & 127u);
                if (r == piv) {
#pragma unroll
                    for (int jj = 0; jj < TPS_TQ; ++jj) prow[q * TPS_TQ + jj] = a[jj];
                    if (q == qq) { diag[r] = a[j]; s_pinv = 1.0 / a[j]; }
                    used = true;
                    mycol = col;
                }
                __syncthreads();
                const double f = arc * s_pinv;
                if (rowok && r != piv) {
#pragma unroll
                    for (int jj = 0; jj < TPS_TQ; ++jj) a[jj] -= f * prow[q * TPS_TQ + jj];
                }
            }
        }
    }
    __syncthreads();
    if (rowok && q == 3) {     // right-hand sides are columns 66, 67 = entries 15, 16 of quarter 3
        const double d = diag[r];
        T[(long long)b * 2 * SS_NT + mycol] = (float)(a[SS_NT - 3 * TPS_TQ] / d);
        T[(long long)b * 2 * SS_NT + SS_NT + mycol] = (float)(a[SS_NT + 1 - 3 * TPS_TQ] / d);
    }
}

extern "C" __attribute__((visibility("default"))) int ss_tps_solve_r4(const float* source, const float* target, float* T, int n, void* stream) {
    if (!source || !target || !T || n <= 0) return SS_ERR_ARG;
    hipLaunchKernelGGL(tps_solve_r4_kernel, dim3(n), dim3(320), 0, (hipStream_t)stream, source, (long long)SS_NV * 2,
                       target, (long long)SS_NV * 2, T);
    return ss_launch_status();
}
#endif

extern "C" int ss_tps_solve(const float* source, const float* target, float* T, int n, void* stream) {
    if (!source || !target || !T || n <= 0) return SS_ERR_ARG;
    hipLaunchKernelGGL(tps_solve_kernel, dim3(n), dim3(64 * TPS_NW), 0, (hipStream_t)stream, source, (long long)SS_NV * 2,
                       target, (long long)SS_NV * 2, T);
    return ss_launch_status();
}

// n systems with n control-point sets and ONE target shared by all of them (the render's splines: every frame's mesh maps
// onto the same rigid mesh, test_online_tra.py:129-137) -- no [n,63,2] broadcast copy of the target
extern "C" int ss_tps_solve_shared_target(const float* source, const float* target, float* T, int n, void* stream) {
    if (!source || !target || !T || n <= 0) return SS_ERR_ARG;
    hipLaunchKernelGGL(tps_solve_kernel, dim3(n), dim3(64 * TPS_NW), 0, (hipStream_t)stream, source, (long long)SS_NV * 2,
                       target, 0ll, T);
    return ss_launch_status();
}

__global__ void tps_points_kernel(const float* __restrict__ point, const float* __restrict__ source,
                                  long long src_stride, const float* __restrict__ T, float* __restrict__ out, int q) {
    __shared__ float sx[SS_NV], sy[SS_NV], Tx[SS_NT], Ty[SS_NT];
    int b = blockIdx.y;
    if (threadIdx.x < SS_NV) {
        sx[threadIdx.x] = source[(long long)b * src_stride + threadIdx.x * 2];
        sy[threadIdx.x] = source[(long long)b * src_stride + threadIdx.x * 2 + 1];
    }
    if (threadIdx.x < SS_NT) {
        Tx[threadIdx.x] = T[(long long)b * 2 * SS_NT + threadIdx.x];
        Ty[threadIdx.x] = T[(long long)b * 2 * SS_NT + SS_NT + threadIdx.x];
    }
    __syncthreads();
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= q) return;
    float x = point[((long long)b * q + i) * 2], y = point[((long long)b * q + i) * 2 + 1];
    float ox, oy;
    tps_eval(sx, sy, Tx, Ty, x, y, ox, oy);
    out[((long long)b * q + i) * 2] = ox;
    out[((long long)b * q + i) * 2 + 1] = oy;
}

extern "C" int ss_tps_points(const float* point, const float* source, const float* T, float* out, int n, int q,
                             void* stream) {
    if (!point || !source || !T || !out || n <= 0 || q <= 0) return SS_ERR_ARG;
    hipLaunchKernelGGL(tps_points_kernel, dim3(ss_cdiv(q, 128), n), dim3(128), 0, (hipStream_t)stream, point, source,
                       (long long)SS_NV * 2, T, out, q);
    return ss_launch_status();
}

// ------------------------------------------------------------------------------------------------
// W^-1 of the 66x66 TPS system of ONE control-point set, in fp64 (torch.inverse(W.double()), utils/
// torch_tps_transform_point.py:113).  Where the source mesh is a constant -- the tsmotion composition always solves from the
// RIGID mesh (test_online_tra.py:330, 338) -- the reference's `T = W_inv @ [target; 0]` is a 66x63 matrix product per system
// once W^-1 is known; the caller caches W^-1 per (image size, device) instead of paying a 94 us latency-bound elimination
// per call.  In-place Gauss-Jordan inversion with partial pivoting, one workgroup, matrix in LDS.
__global__ __launch_bounds__(256) void tps_inverse_kernel(const float* __restrict__ source, double* __restrict__ winv) {
    constexpr int N = SS_NT, LD = SS_NT + 1;
    __shared__ double A[N * LD];
    __shared__ float sx[SS_NV], sy[SS_NV];
    __shared__ double colk[N];
    __shared__ int perm[N];
    __shared__ int s_piv;
    const int tid = threadIdx.x;
    if (tid < SS_NV) { sx[tid] = source[tid * 2]; sy[tid] = source[tid * 2 + 1]; }
    __syncthreads();
    for (int i = tid; i < N * N; i += 256) {
        const int r = i / N, c = i - r * N;
        double v = 0.0;
        if (r < SS_NV) {
            if (c == 0) v = 1.0;
            else if (c == 1) v = sx[r];
            else if (c == 2) v = sy[r];
            else {           // same fp32 kernel entries as tps_solve_kernel (correctly rounded log)
                float dx = __fsub_rn(sx[r], sx[c - 3]), dy = __fsub_rn(sy[r], sy[c - 3]);
                float d2 = __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));
                v = (double)__fmul_rn(d2, (float)log((double)__fadd_rn(d2, 1e-6f)));
            }
        } else if (c >= 3) {
            const int k = r - SS_NV;
            v = k == 0 ? 1.0 : (k == 1 ? (double)sx[c - 3] : (double)sy[c - 3]);
        }
        A[r * LD + c] = v;
    }
    __syncthreads();
    for (int k = 0; k < N; ++k) {
        if (tid < 64) {                      // pivot: largest |A[r][k]|, r >= k (lowest row wins ties)
            double best = -1.0;
            int piv = k;
            for (int r = k + tid; r < N; r += 64) {
                const double v = fabs(A[r * LD + k]);
                if (v > best) { best = v; piv = r; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const double ob = __shfl_xor(best, o, 64);
                const int op = __shfl_xor(piv, o, 64);
                if (ob > best || (ob == best && op < piv)) { best = ob; piv = op; }
            }
            if (tid == 0) { s_piv = piv; perm[k] = piv; }
        }
        __syncthreads();
        const int piv = s_piv;
        if (piv != k && tid < N) {           // swap rows k and piv
            const double t = A[k * LD + tid];
            A[k * LD + tid] = A[piv * LD + tid];
            A[piv * LD + tid] = t;
        }
        __syncthreads();
        if (tid < N) colk[tid] = A[tid * LD + k];
        __syncthreads();
        const double pinv = 1.0 / colk[k];
        if (tid < N) A[k * LD + tid] = (tid == k ? 1.0 : A[k * LD + tid]) * pinv;
        __syncthreads();
        for (int i = tid; i < N * N; i += 256) {
            const int r = i / N, c = i - r * N;
            if (r != k) {
                const double f = colk[r];
                const double base = c == k ? 0.0 : A[r * LD + c];
                A[r * LD + c] = base - f * A[k * LD + c];
            }
        }
        __syncthreads();
    }
    for (int k = N - 1; k >= 0; --k) {       // undo the row exchanges as column exchanges, in reverse order
        const int pk = perm[k];
        if (pk != k && tid < N) {
            const double t = A[tid * LD + k];
            A[tid * LD + k] = A[tid * LD + pk];
            A[tid * LD + pk] = t;
        }
        __syncthreads();
    }
    for (int i = tid; i < N * N; i += 256) winv[i] = A[(i / N) * LD + (i % N)];
}

extern "C" int ss_tps_inverse(const float* source, double* winv, void* stream) {
    if (!source || !winv) return SS_ERR_ARG;
    hipLaunchKernelGGL(tps_inverse_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, source, winv);
    return ss_launch_status();
}

// ------------------------------------------------------------------------------------------------
// tsmotion (test_online_tra.py:309-347), one view, all frames
// ws layout: [nrigid 126][ntgt n*126][npts n*126][T n*132]
// lag: frame k pairs with frame k - lag (1 = one stream, frames in time order; S = S interleaved streams advancing together,
// frame index = time * S + stream); the first `lag` frames have no predecessor (tsmotion 0)
__global__ void tsm_prepare_kernel(const float* __restrict__ smotion, const float* __restrict__ tmotion,
                                   float* __restrict__ smesh, float* __restrict__ ws, int n, float img_h,
                                   float img_w, int lag) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * SS_NV) return;
    int k = idx / SS_NV, v = idx - k * SS_NV;
    float rx, ry;
    rigid_vertex(v, img_h, img_w, rx, ry);
    float* nrigid = ws;
    float* ntgt = ws + 126;
    float* npts = ntgt + (long long)n * 126;
    if (k == 0) { nrigid[v * 2] = norm1(rx, img_w); nrigid[v * 2 + 1] = norm1(ry, img_h); }
    float smx = __fadd_rn(rx, smotion[idx * 2]), smy = __fadd_rn(ry, smotion[idx * 2 + 1]);
    smesh[idx * 2] = smx;
    smesh[idx * 2 + 1] = smy;
    if (k + lag < n) {   // target of frame k+lag = normalised spatial mesh of frame k
        ntgt[(long long)(k + lag) * 126 + v * 2] = norm1(smx, img_w);
        ntgt[(long long)(k + lag) * 126 + v * 2 + 1] = norm1(smy, img_h);
    }
    if (k < lag) { ntgt[(long long)k * 126 + v * 2] = norm1(rx, img_w); ntgt[(long long)k * 126 + v * 2 + 1] = norm1(ry, img_h); }   // unused, keep defined
    npts[(long long)k * 126 + v * 2] = norm1(__fadd_rn(rx, tmotion[idx * 2]), img_w);
    npts[(long long)k * 126 + v * 2 + 1] = norm1(__fadd_rn(ry, tmotion[idx * 2 + 1]), img_h);
}

// winv != nullptr: T = W^-1 [target; 0] computed here (66 x 63 fp64 products per coordinate, rounded to fp32 like the
// reference's `T.type(torch.float32)`); else T comes from the per-system elimination
__global__ __launch_bounds__(256) void tsm_finish_kernel(const float* __restrict__ ws, const float* __restrict__ smesh,
                                                         const double* __restrict__ winv, float* __restrict__ tsmotion,
                                                         int n, float img_h, float img_w, int lag) {
    __shared__ float sx[SS_NV], sy[SS_NV], Tx[SS_NT], Ty[SS_NT];
    __shared__ float tg[SS_NV * 2];
    int k = blockIdx.x, v = threadIdx.x;
    const float* nrigid = ws;
    const float* ntgt = ws + 126;
    const float* npts = ws + 126 + (long long)n * 126;
    const float* T = npts + (long long)n * 126 + (long long)k * 132;
    if (v < SS_NV) { sx[v] = nrigid[v * 2]; sy[v] = nrigid[v * 2 + 1]; }
    if (winv) {
        if (v < SS_NV * 2) tg[v] = ntgt[(long long)k * 126 + v];
        __syncthreads();
        if (v < 2 * SS_NT) {
            const int c = v / SS_NT, r = v - c * SS_NT;
            double acc = 0.0;
            for (int j = 0; j < SS_NV; ++j) acc = fma(winv[r * SS_NT + j], (double)tg[j * 2 + c], acc);
            if (c == 0) Tx[r] = (float)acc; else Ty[r] = (float)acc;
        }
    } else if (v < SS_NT) { Tx[v] = T[v]; Ty[v] = T[SS_NT + v]; }
    __syncthreads();
    if (v >= SS_NV) return;
    long long o = ((long long)k * SS_NV + v) * 2;
    if (k < lag) { tsmotion[o] = 0.f; tsmotion[o + 1] = 0.f; return; }
    float ox, oy;
    tps_eval(sx, sy, Tx, Ty, npts[(long long)k * 126 + v * 2], npts[(long long)k * 126 + v * 2 + 1], ox, oy);
    tsmotion[o] = __fsub_rn(recover1(ox, img_w), smesh[o]);
    tsmotion[o + 1] = __fsub_rn(recover1(oy, img_h), smesh[o + 1]);
}

// tsm_prepare_kernel + tsm_finish_kernel as ONE launch where the rigid-mesh inverse is at hand (round 6: the streaming push pays
// >= 4.5 us per launch whatever it does).  Frame k needs nothing another block computes: its target is frame k - lag's spatial
// mesh, its points are its own temporal mesh -- the same expressions, evaluated here instead of being read back from the workspace:
// bit-identical results, no workspace traffic.
__global__ __launch_bounds__(256) void tsm_fused_kernel(const float* __restrict__ smotion, const float* __restrict__ tmotion,
                                                        const double* __restrict__ winv, float* __restrict__ smesh,
                                                        float* __restrict__ tsmotion, int n, float img_h, float img_w, int lag) {
    __shared__ float sx[SS_NV], sy[SS_NV], Tx[SS_NT], Ty[SS_NT];
    __shared__ float tg[SS_NV * 2];
    const int k = blockIdx.x, v = threadIdx.x;
    float rx = 0.f, ry = 0.f, px = 0.f, py = 0.f, smx = 0.f, smy = 0.f;
    const long long o = ((long long)k * SS_NV + v) * 2;
    if (v < SS_NV) {
        rigid_vertex(v, img_h, img_w, rx, ry);
        sx[v] = norm1(rx, img_w);
        sy[v] = norm1(ry, img_h);
        smx = __fadd_rn(rx, smotion[o]);
        smy = __fadd_rn(ry, smotion[o + 1]);
        smesh[o] = smx;
        smesh[o + 1] = smy;
        if (k >= lag) {        // target of frame k = normalised spatial mesh of frame k - lag
            const long long op = ((long long)(k - lag) * SS_NV + v) * 2;
            tg[v * 2] = norm1(__fadd_rn(rx, smotion[op]), img_w);
            tg[v * 2 + 1] = norm1(__fadd_rn(ry, smotion[op + 1]), img_h);
        }
        px = norm1(__fadd_rn(rx, tmotion[o]), img_w);
        py = norm1(__fadd_rn(ry, tmotion[o + 1]), img_h);
    }
    if (k < lag) {
        if (v < SS_NV) { tsmotion[o] = 0.f; tsmotion[o + 1] = 0.f; }
        return;
    }
    __syncthreads();
    if (v < 2 * SS_NT) {
        const int c = v / SS_NT, r = v - c * SS_NT;
        double acc = 0.0;
        for (int j = 0; j < SS_NV; ++j) acc = fma(winv[r * SS_NT + j], (double)tg[j * 2 + c], acc);
        if (c == 0) Tx[r] = (float)acc; else Ty[r] = (float)acc;
    }
    __syncthreads();
    if (v >= SS_NV) return;
    float ox, oy;
    tps_eval(sx, sy, Tx, Ty, px, py, ox, oy);
    tsmotion[o] = __fsub_rn(recover1(ox, img_w), smx);
    tsmotion[o + 1] = __fsub_rn(recover1(oy, img_h), smy);
}

extern "C" long long ss_tsmotion_workspace_floats(int n) { return 126 + (long long)n * (126 + 126 + 132); }

extern "C" int ss_tsmotion_lag(const float* smotion, const float* tmotion, float* smesh, float* tsmotion, int n, int lag,
                               float img_h, float img_w, const double* rigid_winv, float* ws, void* stream) {
    if (!smotion || !tmotion || !smesh || !tsmotion || !ws || n <= 0 || lag < 1) return SS_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (rigid_winv) {
        hipLaunchKernelGGL(tsm_fused_kernel, dim3(n), dim3(256), 0, st, smotion, tmotion, rigid_winv, smesh, tsmotion, n, img_h,
                           img_w, lag);
        return ss_launch_status();
    }
    hipLaunchKernelGGL(tsm_prepare_kernel, dim3(ss_cdiv(n * SS_NV, 128)), dim3(128), 0, st, smotion, tmotion, smesh, ws,
                       n, img_h, img_w, lag);
    float* ntgt = ws + 126;
    float* T = ntgt + (long long)n * 252;
    if (!rigid_winv)
        hipLaunchKernelGGL(tps_solve_kernel, dim3(n), dim3(64 * TPS_NW), 0, st, (const float*)ws, 0ll, (const float*)ntgt, (long long)SS_NV * 2, T);
    hipLaunchKernelGGL(tsm_finish_kernel, dim3(n), dim3(256), 0, st, (const float*)ws, (const float*)smesh, rigid_winv,
                       tsmotion, n, img_h, img_w, lag);
    return ss_launch_status();
}

extern "C" int ss_tsmotion(const float* smotion, const float* tmotion, float* smesh, float* tsmotion, int n,
                           float img_h, float img_w, const double* rigid_winv, float* ws, void* stream) {
    return ss_tsmotion_lag(smotion, tmotion, smesh, tsmotion, n, 1, img_h, img_w, rigid_winv, ws, stream);
}

// ------------------------------------------------------------------------------------------------
// canvas bbox over scaled meshes (test_online_tra.py:103-120); single block
__global__ void mesh_bbox_kernel(const float* __restrict__ mesh, int npts, float img_h, float img_w,
                                 float* __restrict__ bbox, int accumulate) {
    __shared__ float red[4][4];
    float wmin = INFINITY, wmax = -INFINITY, hmin = INFINITY, hmax = -INFINITY;
    for (int i = threadIdx.x; i < npts; i += blockDim.x) {
        float x = img_w > 0.f ? __fmul_rn(mesh[i * 2], img_w) / 480.0f : mesh[i * 2];
        float y = img_h > 0.f ? __fmul_rn(mesh[i * 2 + 1], img_h) / 360.0f : mesh[i * 2 + 1];
        wmin = fminf(wmin, x); wmax = fmaxf(wmax, x);
        hmin = fminf(hmin, y); hmax = fmaxf(hmax, y);
    }
    wmin = ss_wave_min(wmin); wmax = ss_wave_max(wmax); hmin = ss_wave_min(hmin); hmax = ss_wave_max(hmax);
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { red[wave][0] = wmin; red[wave][1] = wmax; red[wave][2] = hmin; red[wave][3] = hmax; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int wv = 1; wv < 4; ++wv) {
            wmin = fminf(wmin, red[wv][0]); wmax = fmaxf(wmax, red[wv][1]);
            hmin = fminf(hmin, red[wv][2]); hmax = fmaxf(hmax, red[wv][3]);
        }
        if (accumulate) {
            wmin = fminf(wmin, bbox[0]); wmax = fmaxf(wmax, bbox[1]);
            hmin = fminf(hmin, bbox[2]); hmax = fmaxf(hmax, bbox[3]);
        }
        bbox[0] = wmin; bbox[1] = wmax; bbox[2] = hmin; bbox[3] = hmax;
    }
}

extern "C" int ss_mesh_bbox(const float* mesh, int n_points, float img_h, float img_w, float* bbox, int accumulate,
                            void* stream) {
    if (!mesh || !bbox || n_points <= 0) return SS_ERR_ARG;
    hipLaunchKernelGGL(mesh_bbox_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, mesh, n_points, img_h, img_w, bbox,
                       accumulate);
    return ss_launch_status();
}

// test_online_tra.py:103-104, 129-136: scale to HR, translate by (-wmin,-hmin), normalise by the float canvas size
__global__ void mesh_normalize_kernel(const float* __restrict__ mesh, const float* __restrict__ bbox,
                                      float* __restrict__ out, int npts, float img_h, float img_w) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npts) return;
    float wmin = bbox[0], wmax = bbox[1], hmin = bbox[2], hmax = bbox[3];
    float ow = __fsub_rn(wmax, wmin), oh = __fsub_rn(hmax, hmin);
    float x = img_w > 0.f ? __fmul_rn(mesh[i * 2], img_w) / 480.0f : mesh[i * 2];
    float y = img_h > 0.f ? __fmul_rn(mesh[i * 2 + 1], img_h) / 360.0f : mesh[i * 2 + 1];
    out[i * 2] = norm1(__fsub_rn(x, wmin), ow);
    out[i * 2 + 1] = norm1(__fsub_rn(y, hmin), oh);
}

// Three-view composition (threeview:381-420): the five aligned meshes normalised on the first canvas in ONE launch, laid out for
// ONE batched TPS solve + ONE point evaluation of both re-projections: out [6][npts][2] = {a1, b2 (the points), a2, b1 (the
// sources), mid, mid (the targets)}.  Same arithmetic per point as mesh_normalize_kernel (meshes are HR pixels already).
struct FiveMeshes {
    const float* m[5];          // a1, a2, b1, b2, mid
};
__global__ void three_view_normalize_kernel(FiveMeshes ms, const float* __restrict__ bbox, float* __restrict__ out, int npts) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npts) return;
    const float wmin = bbox[0], wmax = bbox[1], hmin = bbox[2], hmax = bbox[3];
    const float ow = __fsub_rn(wmax, wmin), oh = __fsub_rn(hmax, hmin);
    const int slot[6] = {0, 3, 1, 2, 4, 4};
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int j = slot[k];
        const float* m = j == 0 ? ms.m[0] : j == 1 ? ms.m[1] : j == 2 ? ms.m[2] : j == 3 ? ms.m[3] : ms.m[4];
        out[((long long)k * npts + i) * 2] = norm1(__fsub_rn(m[i * 2], wmin), ow);
        out[((long long)k * npts + i) * 2 + 1] = norm1(__fsub_rn(m[i * 2 + 1], hmin), oh);
    }
}
extern "C" int ss_three_view_normalize(const float* a1, const float* a2, const float* b1, const float* b2, const float* mid,
                                       const float* bbox, float* out, long long n_points, void* stream) {
    if (!a1 || !a2 || !b1 || !b2 || !mid || !bbox || !out || n_points <= 0 || n_points >= (1ll << 28)) return SS_ERR_ARG;
    FiveMeshes ms;
    ms.m[0] = a1; ms.m[1] = a2; ms.m[2] = b1; ms.m[3] = b2; ms.m[4] = mid;
    hipLaunchKernelGGL(three_view_normalize_kernel, dim3(ss_cdiv(n_points, 256)), dim3(256), 0, (hipStream_t)stream, ms, bbox, out,
                       (int)n_points);
    return ss_launch_status();
}

// the same for view `view` of `views`, written where the render wants it: frame f's 63 points at out[(f * views + view) * 126]
// (source [frames][views][63][2] assembled by `views` launches, no torch.stack)
// bbox_fs: floats between the canvas boxes of consecutive frames (0: one box for all; 4: a box per frame -- S live streams with a
// canvas each); mesh_fs: floats between consecutive frames' meshes (126 = packed)
__global__ void mesh_normalize_views_kernel(const float* __restrict__ mesh, const float* __restrict__ bbox,
                                            float* __restrict__ out, int npts, int view, int views, float img_h, float img_w,
                                            int bbox_fs, long long mesh_fs) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npts) return;
    const int f = i / SS_NV, k = i - f * SS_NV;
    const float* bb = bbox + (long long)f * bbox_fs;
    const float* m = mesh + (long long)f * mesh_fs + k * 2;
    float wmin = bb[0], wmax = bb[1], hmin = bb[2], hmax = bb[3];
    float ow = __fsub_rn(wmax, wmin), oh = __fsub_rn(hmax, hmin);
    float x = img_w > 0.f ? __fmul_rn(m[0], img_w) / 480.0f : m[0];
    float y = img_h > 0.f ? __fmul_rn(m[1], img_h) / 360.0f : m[1];
    float* o = out + (((long long)f * views + view) * SS_NV + k) * 2;
    o[0] = norm1(__fsub_rn(x, wmin), ow);
    o[1] = norm1(__fsub_rn(y, hmin), oh);
}

extern "C" int ss_mesh_normalize_views_boxes(const float* mesh, long long mesh_frame_stride, const float* bboxes, float* out,
                                             int frames, int view, int views, float img_h, float img_w, void* stream) {
    if (!mesh || !bboxes || !out || frames <= 0 || views <= 0 || view < 0 || view >= views || mesh_frame_stride < 126)
        return SS_ERR_ARG;
    const int npts = frames * SS_NV;
    hipLaunchKernelGGL(mesh_normalize_views_kernel, dim3(ss_cdiv(npts, 256)), dim3(256), 0, (hipStream_t)stream, mesh, bboxes,
                       out, npts, view, views, img_h, img_w, 4, mesh_frame_stride);
    return ss_launch_status();
}

// Streaming mode: does the current frame of every live stream still fit its FIXED canvas?  The reference sizes the canvas from ALL
// frames of the clip (test_online_tra.py:106-120); a stream fixes it after the first window, so a mesh that later drifts past it is
// cropped.  src [streams][views][63][2] = this push's control points, already normalised to each stream's canvas ([-1, 1] = inside):
// per stream (one wave) the extremes of both coordinates; watch_i [streams][4] = {frames seen, frames with a point outside the
// canvas, index of the first such frame (-1), frames with a point closer than `guard` to an edge or outside}, watch_f [streams][4] =
// running {xmin, xmax, ymin, ymax} of the normalised coordinates over all frames seen -- what a grown canvas must cover.  State
// lives on the device and is only read when somebody asks (no sync on the push path); capturable (one fixed-size launch).
// (the wave's running extremes, NaN flag OR-ed over the lanes -> the stream's watcher state; lane 0 writes)
__global__ __launch_bounds__(64) void canvas_watch_kernel(const float* __restrict__ src, int npts, float guard, int* __restrict__ watch_i,
                                                          float* __restrict__ watch_f) {
    const float* s = src + (long long)blockIdx.x * npts * 2;
    float xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
    bool bad = false;
    for (int i = threadIdx.x; i < npts; i += 64) {
        const float x = s[2 * i], y = s[2 * i + 1];
        bad = bad || x != x || y != y;
        xmin = fminf(xmin, x); xmax = fmaxf(xmax, x);
        ymin = fminf(ymin, y); ymax = fmaxf(ymax, y);
    }
    canvas_watch_update(xmin, xmax, ymin, ymax, bad, guard, watch_i + blockIdx.x * 4, watch_f + blockIdx.x * 4);
}
extern "C" int ss_canvas_watch(const float* src, int streams, int views, float guard, int* watch_i, float* watch_f, void* stream) {
    if (!src || !watch_i || !watch_f || streams <= 0 || views <= 0 || !(guard >= 0.f)) return SS_ERR_ARG;
    hipLaunchKernelGGL(canvas_watch_kernel, dim3(streams), dim3(64), 0, (hipStream_t)stream, src, views * SS_NV, guard, watch_i, watch_f);
    return ss_launch_status();
}

// The streaming push's normalisation of ALL views + the overflow watcher as ONE launch (round 6; it was `views` launches of
// mesh_normalize_views_kernel + canvas_watch_kernel per push): one wave per stream, same arithmetic per point, same watcher update.
struct StreamMeshes {
    const float* m[3];
};
__global__ __launch_bounds__(64) void stream_normalize_watch_kernel(StreamMeshes ms, long long mesh_fs, const float* __restrict__ bbox,
                                                                    int bbox_fs, float* __restrict__ out, int views, float img_h,
                                                                    float img_w, float guard, int* __restrict__ watch_i,
                                                                    float* __restrict__ watch_f) {
    const int f = blockIdx.x;
    const float* bb = bbox + (long long)f * bbox_fs;
    const float wmin = bb[0], wmax = bb[1], hmin = bb[2], hmax = bb[3];
    const float ow = __fsub_rn(wmax, wmin), oh = __fsub_rn(hmax, hmin);
    float xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
    bool bad = false;
    for (int v = 0; v < views; ++v) {
        const float* mesh = v == 0 ? ms.m[0] : (v == 1 ? ms.m[1] : ms.m[2]);
        const int k = threadIdx.x;
        if (k < SS_NV) {
            const float* m = mesh + (long long)f * mesh_fs + k * 2;
            const float x = img_w > 0.f ? __fmul_rn(m[0], img_w) / 480.0f : m[0];
            const float y = img_h > 0.f ? __fmul_rn(m[1], img_h) / 360.0f : m[1];
            const float nx = norm1(__fsub_rn(x, wmin), ow), ny = norm1(__fsub_rn(y, hmin), oh);
            float* o = out + (((long long)f * views + v) * SS_NV + k) * 2;
            o[0] = nx;
            o[1] = ny;
            bad = bad || nx != nx || ny != ny;
            xmin = fminf(xmin, nx); xmax = fmaxf(xmax, nx);
            ymin = fminf(ymin, ny); ymax = fmaxf(ymax, ny);
        }
    }
    if (watch_i) canvas_watch_update(xmin, xmax, ymin, ymax, bad, guard, watch_i + f * 4, watch_f + f * 4);
}
extern "C" int ss_stream_normalize_watch(const float* const* meshes, int views, long long mesh_frame_stride, const float* bboxes,
                                         int bbox_frame_stride, float* out, int streams, float img_h, float img_w, float guard,
                                         int* watch_i, float* watch_f, void* stream) {
    if (!meshes || !bboxes || !out || streams <= 0 || views <= 0 || views > 3 || mesh_frame_stride < 0 ||
        (bbox_frame_stride != 0 && bbox_frame_stride != 4) || !(guard >= 0.f) || (!watch_i != !watch_f))
        return SS_ERR_ARG;
    StreamMeshes ms;
    for (int v = 0; v < 3; ++v) {
        ms.m[v] = v < views ? meshes[v] : nullptr;
        if (v < views && !ms.m[v]) return SS_ERR_ARG;
    }
    hipLaunchKernelGGL(stream_normalize_watch_kernel, dim3(streams), dim3(64), 0, (hipStream_t)stream, ms, mesh_frame_stride, bboxes,
                       bbox_frame_stride, out, views, img_h, img_w, guard, watch_i, watch_f);
    return ss_launch_status();
}
extern "C" int ss_mesh_normalize_views(const float* mesh, const float* bbox, float* out, int frames, int view, int views,
                                       float img_h, float img_w, void* stream) {
    if (!mesh || !bbox || !out || frames <= 0 || views <= 0 || view < 0 || view >= views) return SS_ERR_ARG;
    const int npts = frames * SS_NV;
    hipLaunchKernelGGL(mesh_normalize_views_kernel, dim3(ss_cdiv(npts, 256)), dim3(256), 0, (hipStream_t)stream, mesh, bbox,
                       out, npts, view, views, img_h, img_w, 0, (long long)SS_NV * 2);
    return ss_launch_status();
}

__global__ void fill_kernel(float* __restrict__ p, float v, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
extern "C" int ss_fill_f32(float* p, float value, long long n, void* stream) {
    if (!p || n <= 0) return SS_ERR_ARG;
    hipLaunchKernelGGL(fill_kernel, dim3(ss_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, p, value, n);
    return ss_launch_status();
}

extern "C" int ss_mesh_normalize(const float* mesh, const float* bbox, float* out, int n_points, float img_h,
                                 float img_w, void* stream) {
    if (!mesh || !bbox || !out || n_points <= 0) return SS_ERR_ARG;
    hipLaunchKernelGGL(mesh_normalize_kernel, dim3(ss_cdiv(n_points, 256)), dim3(256), 0, (hipStream_t)stream, mesh,
                       bbox, out, n_points, img_h, img_w);
    return ss_launch_status();
}

// ------------------------------------------------------------------------------------------------
// H2Mesh (spatial_network.py:20-36): mesh = persp_divide(H^-1 [x y 1]^T) over the vertices of a mesh; 3 x 3 inverse and
// the products in fp64 (the reference's fp32 torch.inverse is itself +-0.02 px from it), result fp32.
__global__ void h2mesh_kernel(const float* __restrict__ H, const float* __restrict__ mesh, float* __restrict__ out, int npts) {
    __shared__ double Hi[9];
    const int b = blockIdx.y;
    if (threadIdx.x == 0) {
        double h[9], inv[9];
        for (int i = 0; i < 9; ++i) h[i] = (double)H[b * 9 + i];
        inv3(h, inv);
        for (int i = 0; i < 9; ++i) Hi[i] = inv[i];
    }
    __syncthreads();
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= npts) return;
    const double x = (double)mesh[((long long)b * npts + v) * 2], y = (double)mesh[((long long)b * npts + v) * 2 + 1];
    const double X = Hi[0] * x + Hi[1] * y + Hi[2], Y = Hi[3] * x + Hi[4] * y + Hi[5], Z = Hi[6] * x + Hi[7] * y + Hi[8];
    out[((long long)b * npts + v) * 2] = (float)(X / Z);
    out[((long long)b * npts + v) * 2 + 1] = (float)(Y / Z);
}

extern "C" int ss_h2mesh(const float* H, const float* mesh, float* out, int n, int n_points, void* stream) {
    if (!H || !mesh || !out || n <= 0 || n_points <= 0 || n > 65535) return SS_ERR_ARG;
    hipLaunchKernelGGL(h2mesh_kernel, dim3(ss_cdiv(n_points, 64), n), dim3(64), 0, (hipStream_t)stream, H, mesh, out, n_points);
    return ss_launch_status();
}

// ------------------------------------------------------------------------------------------------
// three-view mesh alignment (test_online_tra_threeview.py:345-420).  Per frame (one 64-thread block, thread = vertex):
//   scale the four LR meshes to HR (x * W / 480, y * H / 360); offset = mean over the 63 vertices of (w12_m2 - w23_m1);
//   add it to both meshes of pair (2,3); middle = (w12_m2 + w23_m1) / 2.   -> a1, a2, b1, b2, mid [n][63][2] in HR pixels,
//   not yet translated to the first canvas (ss_mesh_bbox / ss_mesh_normalize / ss_three_view_finish do that).
__global__ void three_view_align_kernel(const float* __restrict__ m12_1, const float* __restrict__ m12_2,
                                        const float* __restrict__ m23_1, const float* __restrict__ m23_2,
                                        float* __restrict__ a1, float* __restrict__ a2, float* __restrict__ b1,
                                        float* __restrict__ b2, float* __restrict__ mid, float img_h, float img_w) {
    const long long f = blockIdx.x;
    const int v = threadIdx.x;
    const bool on = v < SS_NV;
    const long long o = (f * SS_NV + (on ? v : 0)) * 2;
    auto sx = [&](float x) { return __fmul_rn(x, img_w) / 480.0f; };
    auto sy = [&](float y) { return __fmul_rn(y, img_h) / 360.0f; };
    const float a1x = sx(m12_1[o]), a1y = sy(m12_1[o + 1]), a2x = sx(m12_2[o]), a2y = sy(m12_2[o + 1]);
    float b1x = sx(m23_1[o]), b1y = sy(m23_1[o + 1]), b2x = sx(m23_2[o]), b2y = sy(m23_2[o + 1]);
    const float ox = ss_wave_sum(on ? __fsub_rn(a2x, b1x) : 0.f) / (float)SS_NV;
    const float oy = ss_wave_sum(on ? __fsub_rn(a2y, b1y) : 0.f) / (float)SS_NV;
    if (!on) return;
    b1x = __fadd_rn(b1x, ox); b1y = __fadd_rn(b1y, oy);
    b2x = __fadd_rn(b2x, ox); b2y = __fadd_rn(b2y, oy);
    a1[o] = a1x; a1[o + 1] = a1y; a2[o] = a2x; a2[o + 1] = a2y;
    b1[o] = b1x; b1[o + 1] = b1y; b2[o] = b2x; b2[o + 1] = b2y;
    mid[o] = __fadd_rn(a2x, b1x) / 2.0f;
    mid[o + 1] = __fadd_rn(a2y, b1y) / 2.0f;
}

extern "C" int ss_three_view_align(const float* w12_m1, const float* w12_m2, const float* w23_m1, const float* w23_m2,
                                   float* a1, float* a2, float* b1, float* b2, float* mid, int frames, float img_h,
                                   float img_w, void* stream) {
    if (!w12_m1 || !w12_m2 || !w23_m1 || !w23_m2 || !a1 || !a2 || !b1 || !b2 || !mid || frames <= 0) return SS_ERR_ARG;
    hipLaunchKernelGGL(three_view_align_kernel, dim3(frames), dim3(64), 0, (hipStream_t)stream, w12_m1, w12_m2, w23_m1, w23_m2,
                       a1, a2, b1, b2, mid, img_h, img_w);
    return ss_launch_status();
}

// back from the first canvas' normalised coordinates: mesh = (n + 1) * extent / 2 for the re-projected outer meshes (n1, n3),
// middle = mid - (wmin, hmin); bbox (device) = wmin, wmax, hmin, hmax of the first canvas (threeview:405-420)
__global__ void three_view_finish_kernel(const float* __restrict__ n1, const float* __restrict__ n3,
                                         const float* __restrict__ mid, const float* __restrict__ bbox,
                                         float* __restrict__ mesh1, float* __restrict__ middle, float* __restrict__ mesh3,
                                         long long npts) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npts) return;
    const float wmin = bbox[0], hmin = bbox[2];
    const float ow = __fsub_rn(bbox[1], wmin), oh = __fsub_rn(bbox[3], hmin);
    mesh1[i * 2] = recover1(n1[i * 2], ow); mesh1[i * 2 + 1] = recover1(n1[i * 2 + 1], oh);
    mesh3[i * 2] = recover1(n3[i * 2], ow); mesh3[i * 2 + 1] = recover1(n3[i * 2 + 1], oh);
    middle[i * 2] = __fsub_rn(mid[i * 2], wmin); middle[i * 2 + 1] = __fsub_rn(mid[i * 2 + 1], hmin);
}

extern "C" int ss_three_view_finish(const float* n1, const float* n3, const float* mid, const float* bbox, float* mesh1,
                                    float* middle, float* mesh3, long long n_points, void* stream) {
    if (!n1 || !n3 || !mid || !bbox || !mesh1 || !middle || !mesh3 || n_points <= 0) return SS_ERR_ARG;
    hipLaunchKernelGGL(three_view_finish_kernel, dim3(ss_cdiv(n_points, 256)), dim3(256), 0, (hipStream_t)stream, n1, n3, mid,
                       bbox, mesh1, middle, mesh3, n_points);
    return ss_launch_status();
}

// Streaming three-view push: everything between the pair chains' smoothed meshes and the render's splines as ONE launch
// (round 6; it was align, normalise, solve, points, finish, normalise + watch, solve = 7 graph nodes on the push's critical path).
// Workgroup (view k, frame f): the alignment of frame f (three_view_align_kernel's arithmetic, recomputed by each of the three
// workgroups: 63 vertices); for the outer views the re-projection through the pair's spline on the FIRST canvas (three_view_normalize
// -> tps_solve -> tps_points -> three_view_finish, same device functions, same order); then the view's final mesh normalised on the
// OUTPUT canvas and the render's spline onto the rigid mesh (stream_normalize_watch's arithmetic -> tps_solve_shared_target).  The two
// eliminations of an outer view are the same code run twice (second pass: instruction cache warm).  Bit-identical to the launches it
// replaces (tests/test_gpu_round6.py); the overflow watcher moves into the footprint launch (ss_render_footprints_watch).
struct TvSplinesP {
    const float *m12_1, *m12_2, *m23_1, *m23_2;       // smoothed pair meshes, LR scale; frame f at + f * mesh_fs floats
    long long mesh_fs;
    const float *first_box, *out_box, *nrigid;        // (wmin, wmax, hmin, hmax) x 2; normalised rigid mesh [63][2]
    float *mesh1, *middle, *mesh3;                    // [frames][63][2], first-canvas pixels
    float *src, *T;                                   // [frames][3][63][2] normalised on the output canvas; [frames][3][2][66]
    float img_h, img_w;
};
__global__ __launch_bounds__(64 * TPS_NW) void three_view_splines_kernel(TvSplinesP p) {
    __shared__ TpsShared sh;
    __shared__ float al[5][SS_NV * 2];                // a1, a2, b1, b2, mid in HR pixels
    __shared__ float tg[SS_NV * 2], Tl[2 * SS_NT], fin[SS_NV * 2];
    const int k = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const long long f = blockIdx.y;
    const int q = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (q == 0) {                                      // three_view_align_kernel, thread = vertex
        const int v = lane;
        const bool on = v < SS_NV;
        const long long o = f * p.mesh_fs + (on ? v : 0) * 2;
        auto sx = [&](float x) { return __fmul_rn(x, p.img_w) / 480.0f; };
        auto sy = [&](float y) { return __fmul_rn(y, p.img_h) / 360.0f; };
        const float a1x = sx(p.m12_1[o]), a1y = sy(p.m12_1[o + 1]), a2x = sx(p.m12_2[o]), a2y = sy(p.m12_2[o + 1]);
        float b1x = sx(p.m23_1[o]), b1y = sy(p.m23_1[o + 1]), b2x = sx(p.m23_2[o]), b2y = sy(p.m23_2[o + 1]);
        const float ox = ss_wave_sum(on ? __fsub_rn(a2x, b1x) : 0.f) / (float)SS_NV;
        const float oy = ss_wave_sum(on ? __fsub_rn(a2y, b1y) : 0.f) / (float)SS_NV;
        if (on) {
            b1x = __fadd_rn(b1x, ox); b1y = __fadd_rn(b1y, oy);
            b2x = __fadd_rn(b2x, ox); b2y = __fadd_rn(b2y, oy);
            al[0][2 * v] = a1x; al[0][2 * v + 1] = a1y; al[1][2 * v] = a2x; al[1][2 * v + 1] = a2y;
            al[2][2 * v] = b1x; al[2][2 * v + 1] = b1y; al[3][2 * v] = b2x; al[3][2 * v + 1] = b2y;
            al[4][2 * v] = __fadd_rn(a2x, b1x) / 2.0f;
            al[4][2 * v + 1] = __fadd_rn(a2y, b1y) / 2.0f;
        }
    }
    __syncthreads();
    const float fwmin = p.first_box[0], fhmin = p.first_box[2];
    const float fow = __fsub_rn(p.first_box[1], fwmin), foh = __fsub_rn(p.first_box[3], fhmin);
    const float owmin = p.out_box[0], ohmin = p.out_box[2];
    const float oow = __fsub_rn(p.out_box[1], owmin), ooh = __fsub_rn(p.out_box[3], ohmin);
    float* mesh_out = (k == 0 ? p.mesh1 : (k == 1 ? p.middle : p.mesh3)) + f * SS_NV * 2;
    float px = 0.f, py = 0.f;                          // the point this thread re-projects (pass 0)
    if (k == 1 && tid < SS_NV) {                       // the middle plane needs no spline: three_view_finish's translation
        fin[2 * tid] = __fsub_rn(al[4][2 * tid], fwmin);
        fin[2 * tid + 1] = __fsub_rn(al[4][2 * tid + 1], fhmin);
        mesh_out[2 * tid] = fin[2 * tid];
        mesh_out[2 * tid + 1] = fin[2 * tid + 1];
    }
#pragma unroll 1
    for (int pass = (k == 1 ? 1 : 0); pass < 2; ++pass) {
        if (tid < SS_NV) {
            if (pass == 0) {                           // three_view_normalize_kernel: points a1 | b2, sources a2 | b1, targets mid
                const float* P = k == 0 ? al[0] : al[3];
                const float* S = k == 0 ? al[1] : al[2];
                px = norm1(__fsub_rn(P[2 * tid], fwmin), fow); py = norm1(__fsub_rn(P[2 * tid + 1], fhmin), foh);
                sh.sx[tid] = norm1(__fsub_rn(S[2 * tid], fwmin), fow); sh.sy[tid] = norm1(__fsub_rn(S[2 * tid + 1], fhmin), foh);
                tg[2 * tid] = norm1(__fsub_rn(al[4][2 * tid], fwmin), fow); tg[2 * tid + 1] = norm1(__fsub_rn(al[4][2 * tid + 1], fhmin), foh);
            } else {                                   // stream_normalize_watch_kernel (meshes are canvas pixels already)
                const float nx = norm1(__fsub_rn(fin[2 * tid], owmin), oow), ny = norm1(__fsub_rn(fin[2 * tid + 1], ohmin), ooh);
                sh.sx[tid] = nx; sh.sy[tid] = ny;
                float* so = p.src + ((f * 3 + k) * SS_NV + tid) * 2;
                so[0] = nx; so[1] = ny;
                tg[2 * tid] = p.nrigid[2 * tid]; tg[2 * tid + 1] = p.nrigid[2 * tid + 1];
            }
        }
        __syncthreads();
        tps_eliminate(sh, tg, pass == 0 ? Tl : p.T + (f * 3 + k) * 2 * SS_NT, lane, q);
        __syncthreads();
        if (pass == 0 && tid < SS_NV) {                // tps_points_kernel + three_view_finish_kernel
            float ox, oy;
            tps_eval(sh.sx, sh.sy, Tl, Tl + SS_NT, px, py, ox, oy);
            fin[2 * tid] = recover1(ox, fow); fin[2 * tid + 1] = recover1(oy, foh);
            mesh_out[2 * tid] = fin[2 * tid];
            mesh_out[2 * tid + 1] = fin[2 * tid + 1];
        }
        __syncthreads();
    }
}

extern "C" int ss_three_view_splines(const float* w12_m1, const float* w12_m2, const float* w23_m1, const float* w23_m2,
                                     long long mesh_frame_stride, const float* first_box, const float* out_box, const float* nrigid,
                                     float* mesh1, float* middle, float* mesh3, float* src, float* T, int frames, float img_h,
                                     float img_w, void* stream) {
    if (!w12_m1 || !w12_m2 || !w23_m1 || !w23_m2 || !first_box || !out_box || !nrigid || !mesh1 || !middle || !mesh3 || !src || !T ||
        frames <= 0 || frames > 65535 || mesh_frame_stride < 0)
        return SS_ERR_ARG;
    TvSplinesP p;
    p.m12_1 = w12_m1; p.m12_2 = w12_m2; p.m23_1 = w23_m1; p.m23_2 = w23_m2;
    p.mesh_fs = mesh_frame_stride;
    p.first_box = first_box; p.out_box = out_box; p.nrigid = nrigid;
    p.mesh1 = mesh1; p.middle = middle; p.mesh3 = mesh3; p.src = src; p.T = T;
    p.img_h = img_h; p.img_w = img_w;
    hipLaunchKernelGGL(three_view_splines_kernel, dim3(3, frames), dim3(64 * TPS_NW), 0, (hipStream_t)stream, p);
    return ss_launch_status();
}

// A streaming push's render splines as ONE launch: workgroup (view v, stream f) normalises its mesh on the stream's canvas
// (stream_normalize_watch_kernel's arithmetic) and solves the spline onto the rigid mesh (tps_solve_shared_target) -- it was a
// normalisation launch and a solve launch; the watcher moves into the footprint launch (ss_render_footprints_watch).
struct StreamSplinesP {
    const float* m[3];
    long long mesh_fs;
    const float *bbox, *nrigid;
    int bbox_fs;
    float *src, *T;
    int views;
    float img_h, img_w;
};
__global__ __launch_bounds__(64 * TPS_NW) void stream_splines_kernel(StreamSplinesP p) {
    __shared__ TpsShared sh;
    __shared__ float tg[SS_NV * 2];
    const int v = blockIdx.x, tid = threadIdx.x;
    const long long f = blockIdx.y;
    if (tid < SS_NV) {
        const float* bb = p.bbox + f * p.bbox_fs;
        const float wmin = bb[0], hmin = bb[2];
        const float ow = __fsub_rn(bb[1], wmin), oh = __fsub_rn(bb[3], hmin);
        const float* mesh = v == 0 ? p.m[0] : (v == 1 ? p.m[1] : p.m[2]);
        const float* m = mesh + f * p.mesh_fs + tid * 2;
        const float x = p.img_w > 0.f ? __fmul_rn(m[0], p.img_w) / 480.0f : m[0];
        const float y = p.img_h > 0.f ? __fmul_rn(m[1], p.img_h) / 360.0f : m[1];
        const float nx = norm1(__fsub_rn(x, wmin), ow), ny = norm1(__fsub_rn(y, hmin), oh);
        sh.sx[tid] = nx; sh.sy[tid] = ny;
        float* so = p.src + ((f * p.views + v) * SS_NV + tid) * 2;
        so[0] = nx; so[1] = ny;
        tg[2 * tid] = p.nrigid[2 * tid]; tg[2 * tid + 1] = p.nrigid[2 * tid + 1];
    }
    __syncthreads();
    tps_eliminate(sh, tg, p.T + (f * p.views + v) * 2 * SS_NT, tid & 63, __builtin_amdgcn_readfirstlane(tid >> 6));
}

extern "C" int ss_stream_splines(const float* const* meshes, int views, long long mesh_frame_stride, const float* bboxes,
                                 int bbox_frame_stride, const float* nrigid, float* src, float* T, int streams, float img_h,
                                 float img_w, void* stream) {
    if (!meshes || !bboxes || !nrigid || !src || !T || streams <= 0 || streams > 65535 || views <= 0 || views > 3 || mesh_frame_stride < 0 ||
        (bbox_frame_stride != 0 && bbox_frame_stride != 4))
        return SS_ERR_ARG;
    StreamSplinesP p;
    for (int v = 0; v < 3; ++v) {
        p.m[v] = v < views ? meshes[v] : nullptr;
        if (v < views && !p.m[v]) return SS_ERR_ARG;
    }
    p.mesh_fs = mesh_frame_stride; p.bbox = bboxes; p.bbox_fs = bbox_frame_stride; p.nrigid = nrigid; p.src = src; p.T = T;
    p.views = views; p.img_h = img_h; p.img_w = img_w;
    hipLaunchKernelGGL(stream_splines_kernel, dim3(views, streams), dim3(64 * TPS_NW), 0, (hipStream_t)stream, p);
    return ss_launch_status();
}
