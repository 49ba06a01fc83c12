// Micro-benchmark: where do the ~2900 clocks per elimination step of tps_solve_kernel (csrc/geom.hip: 66x68 fp64 Gauss-Jordan in
// registers, 320 threads, two barriers per column) go?  The same loop with pieces compiled out (results then wrong; timing only):
//   1 no pivot search (atomic max)   2 no fp64 division   4 no pivot-row broadcast (LDS write + 17 reads per thread)
//   8 no row-group shuffle           16 no barriers       32 no fp64 FMAs
//   hipcc --offload-arch=gfx950 -O3 tps_solve_probe.hip -o tps_solve_probe && ./tps_solve_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define NT 66
#define LD 68
#define TQ 17

template <int ABL, int SEARCH = 0>
__global__ __launch_bounds__(320) void probe(const double* __restrict__ A0, double* __restrict__ out) {
    __shared__ double prow[LD], diag[NT];
    __shared__ unsigned long long pkey[NT];
    __shared__ double s_pinv;
    __shared__ unsigned wkey[2][8];
    __shared__ double cand[2][5][LD + 2];            // SEARCH 3: every wave's candidate pivot row (+ 1 / pivot at [LD])
    const int tid = threadIdx.x, r = tid >> 2, q = tid & 3;
    const int wave = tid >> 6;
    const bool rowok = r < NT;
    if (tid < NT) pkey[tid] = 0ull;
    if (tid == 0) s_pinv = 1.0;
    double a[TQ];
#pragma unroll
    for (int j = 0; j < TQ; ++j) a[j] = rowok ? A0[(blockIdx.x * NT + r) * LD + q * TQ + j] : 0.0;
    __syncthreads();
    bool used = false;
    int mycol = 0;
    for (int qq = 0; qq < 4; ++qq) {
#pragma unroll
        for (int j = 0; j < TQ; ++j) {
            const int col = qq * TQ + j;
            if (col < NT) {
                if (!(ABL & 1) && SEARCH == 0) {
                    if (rowok && q == qq && !used) {
                        const unsigned long long key = ((unsigned long long)__double_as_longlong(fabs(a[j])) & ~127ull) | (unsigned long long)(127 - r);
                        atomicMax(&pkey[col], key);
                    }
                }
                if (SEARCH >= 1) {
                    // 32-bit key: the magnitude as fp32 bits (16 mantissa bits kept) | (127 - row); 0 for rows already used
                    unsigned key = 0u;
                    if (rowok && q == qq && !used) key = (__float_as_uint((float)fabs(a[j])) & ~127u) | (unsigned)(127 - r);
                    if (SEARCH == 1) {
#pragma unroll
                        for (int o = 4; o < 64; o <<= 1) key = max(key, (unsigned)__shfl_xor((int)key, o, 64));
                    } else {
                        // lanes l, l+4, l+8, l+12 of a 16-lane row share q: two row rotations give every lane its row's maximum
                        key = max(key, (unsigned)__builtin_amdgcn_update_dpp(0, (int)key, 0x124, 0xF, 0xF, false));   // row_ror:4
                        key = max(key, (unsigned)__builtin_amdgcn_update_dpp(0, (int)key, 0x128, 0xF, 0xF, false));   // row_ror:8
                        const unsigned k0 = __builtin_amdgcn_readlane((int)key, qq), k1 = __builtin_amdgcn_readlane((int)key, 16 + qq);
                        const unsigned k2 = __builtin_amdgcn_readlane((int)key, 32 + qq), k3 = __builtin_amdgcn_readlane((int)key, 48 + qq);
                        key = max(max(k0, k1), max(k2, k3));
                    }
                    if ((tid & 63) == 0) wkey[col & 1][wave] = key;
                    if (SEARCH == 3) {
                        // the wave's own best row is published speculatively, with its reciprocal: ONE barrier per column
                        const int rw = 127 - (int)(key & 127u);
                        if (key != 0u && r == rw) {
                            double* c = cand[col & 1][wave];
#pragma unroll
                            for (int jj = 0; jj < TQ; ++jj) c[q * TQ + jj] = a[jj];
                            if (q == qq) c[LD] = 1.0 / a[j];
                        }
                    }
                }
                double arc = a[j];
                if (!(ABL & 8)) arc = __shfl(a[j], (tid & 60) | qq, 64);
                if (SEARCH == 3) {
                    __syncthreads();
                    const unsigned* wk = wkey[col & 1];
                    unsigned best = wk[0]; int bw = 0;
#pragma unroll
                    for (int w = 1; w < 5; ++w) if (wk[w] > best) { best = wk[w]; bw = w; }
                    const int piv = 127 - (int)(best & 127u);
                    const double* c = cand[col & 1][bw];
                    if (r == piv) { if (q == qq) diag[r] = a[j]; used = true; mycol = col; }
                    const double f = arc * c[LD];
                    if (rowok && r != piv) {
#pragma unroll
                        for (int jj = 0; jj < TQ; ++jj) a[jj] -= f * c[q * TQ + jj];
                    }
                    continue;
                }
                if (!(ABL & 16)) __syncthreads();
                int piv;
                if (SEARCH >= 1) {
                    const unsigned* wk = wkey[col & 1];
                    piv = 127 - (int)(max(max(max(wk[0], wk[1]), max(wk[2], wk[3])), wk[4]) & 127u);
                } else piv = (ABL & 1) ? col : 127 - (int)(pkey[col] & 127ull);
                if (r == piv) {
                    if (!(ABL & 4)) {
#pragma unroll
                        for (int jj = 0; jj < TQ; ++jj) prow[q * TQ + jj] = a[jj];
                    }
                    if (q == qq) { diag[r] = a[j]; s_pinv = (ABL & 2) ? a[j] : 1.0 / a[j]; }
                    used = true;
                    mycol = col;
                }
                if (!(ABL & 16)) __syncthreads();
                const double f = arc * s_pinv;
                if (!(ABL & 32)) {
                    if (rowok && r != piv) {
#pragma unroll
                        for (int jj = 0; jj < TQ; ++jj) a[jj] -= f * ((ABL & 4) ? (double)(jj + q) : prow[q * TQ + jj]);
                    }
                } else a[0] += f;
            }
        }
    }
    __syncthreads();
    if (rowok && q == 3) out[blockIdx.x * NT + mycol] = a[NT - 3 * TQ] / diag[r] + a[0];
}

template <int ABL, int SEARCH = 0>
static void run(const double* A, double* out, int n, const char* what) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe<ABL, SEARCH>), dim3(n), dim3(320), 0, 0, A, out);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((probe<ABL, SEARCH>), dim3(n), dim3(320), 0, 0, A, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-46s n=%2d: %6.1f us per launch = %5.0f clocks per column at 2.3 GHz\n", what, n, ms * 1e3 / 20, ms * 1e3 / 20 / 66 * 2300);
}

int main() {
    const int n = 64;
    double* hA = new double[n * NT * LD];
    unsigned s = 12345u;
    for (int i = 0; i < n * NT * LD; ++i) { s = s * 1664525u + 1013904223u; hA[i] = (double)(s >> 8) / (1 << 24) - 0.5; }
    for (int b = 0; b < n; ++b) for (int r = 0; r < NT; ++r) hA[(b * NT + r) * LD + r] += 8.0;
    double *A, *out;
    hipMalloc(&A, sizeof(double) * n * NT * LD); hipMalloc(&out, sizeof(double) * n * NT);
    hipMemcpy(A, hA, sizeof(double) * n * NT * LD, hipMemcpyHostToDevice);
    for (int nn : {2, 64}) {
        run<0>(A, out, nn, "full step");
        run<0, 1>(A, out, nn, "pivot search: 32-bit key, 4 shuffle rounds");
        run<0, 2>(A, out, nn, "pivot search: 32-bit key, 2 DPP + 4 readlane");
        run<0, 3>(A, out, nn, "... + speculative candidate rows, ONE barrier");
        run<1>(A, out, nn, "no pivot search (atomic max)");
        run<2>(A, out, nn, "no fp64 division");
        run<4>(A, out, nn, "no pivot-row broadcast through LDS");
        run<8>(A, out, nn, "no row-group shuffle");
        run<16>(A, out, nn, "no barriers");
        run<32>(A, out, nn, "no fp64 FMAs");
        run<1 | 2 | 8>(A, out, nn, "no search, division, shuffle");
        run<1 | 2 | 4 | 8>(A, out, nn, "... and no broadcast");
        run<1 | 2 | 4 | 8 | 16>(A, out, nn, "... and no barriers");
        run<63>(A, out, nn, "empty loop");
    }
    return 0;
}
