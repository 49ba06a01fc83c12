// Micro-benchmark: what does a wave that is NOT in an MFMA stream get to issue while the other wave of its SIMD runs dense
// v_mfma_f32_32x32x2_f32?  Workgroup = 8 waves (2 per SIMD): waves 0-3 stream MFMAs, waves 4-7 run `iters` x 64 instructions of
// one kind (independent v_fma chains, one dependent v_fma chain, ds_read_b128, v_max3) and report clocks per instruction
// (s_memtime), with the MFMA waves running or idle, at priority 0 or 3.
//   hipcc --offload-arch=gfx950 -O3 mfma_neighbor.hip -o mfma_neighbor
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND, int PRIO>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* clk, int iters, int mfma_on, float seed) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    lds[threadIdx.x] = seed + lane; lds[threadIdx.x + 512] = seed;
    __syncthreads();
    if (wave < 4) {
        if (!mfma_on) return;
        f32x16 a0 = {}, a1 = {};
        float x = seed + lane, y = seed * 0.5f;
        for (int i = 0; i < iters * 3; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
            }
        }
        float s = 0.f;
        for (int r = 0; r < 16; ++r) s += a0[r] + a1[r];
        if (s == 1234.5f) out[threadIdx.x] = s;
        return;
    }
    if (PRIO) __builtin_amdgcn_s_setprio(3);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = seed + j + lane;
    f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {                      // 8 independent chains of v_fma_f32
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = __builtin_fmaf(v[j], 1.0001f, seed);
        } else if (KIND == 1) {               // one dependent chain
#pragma unroll
            for (int u = 0; u < 64; ++u) v[0] = __builtin_fmaf(v[0], 1.0001f, seed);
        } else if (KIND == 2) {               // ds_read_b128, independent
#pragma unroll
            for (int u = 0; u < 64; ++u) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(lds + ((lane * 4 + u * 16) & 1023));
                acc4 += t;
            }
        } else {                              // v_max3_f32, 8 independent chains
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = __builtin_fmaxf(__builtin_fmaxf(v[j], seed), v[(j + 1) & 7]);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = acc4[0] + acc4[1] + acc4[2] + acc4[3];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j];
    if (s == 1234.5f) out[threadIdx.x] = s;
    if (lane == 0 && wave == 4) clk[blockIdx.x] = t1 - t0;
}

template <int KIND, int PRIO>
void run(const char* what, float* d, unsigned long long* c) {
    const int iters = 200, blocks = 256;
    for (int on = 0; on < 2; ++on) {
        hipLaunchKernelGGL((k<KIND, PRIO>), dim3(blocks), dim3(512), 0, 0, d, c, iters, on, 1.0f);
        hipDeviceSynchronize();
        unsigned long long h[256];
        hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
        double s = 0;
        for (int i = 0; i < blocks; ++i) s += (double)h[i];
        const double per = s / blocks / (iters * 64.0) / (KIND == 2 ? 1.0 : (KIND == 3 ? 2.0 : 1.0));
        printf("%-34s prio %d, MFMA neighbour %-3s: %7.1f clocks per instruction\n", what, PRIO ? 3 : 0, on ? "ON" : "off", per);
    }
}

int main() {
    float* d; unsigned long long* c;
    hipMalloc(&d, 4096); hipMalloc(&c, 256 * 8);
    run<0, 0>("v_fma_f32, 8 independent chains", d, c); run<0, 1>("v_fma_f32, 8 independent chains", d, c);
    run<1, 0>("v_fma_f32, one dependent chain", d, c);  run<1, 1>("v_fma_f32, one dependent chain", d, c);
    run<2, 0>("ds_read_b128 + 4 v_add", d, c);          run<2, 1>("ds_read_b128 + 4 v_add", d, c);
    run<3, 0>("v_max3 (2 v_max per step)", d, c);       run<3, 1>("v_max3 (2 v_max per step)", d, c);
    return 0;
}
