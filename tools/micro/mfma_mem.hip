// Micro-benchmark: MFMA-pipe time stolen by LDS / VMEM instructions issued between fp32 MFMAs (gfx950).
// Per iteration: 16 dependent MFMAs + N instructions of one kind.  6 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// KIND 0: none, 1: ds_read_b128, 2: ds_write_b128, 3: global float4 load (L2/L1 hit), 4: s_barrier
template <int KIND, int N>
__global__ __launch_bounds__(256) void k(float* out, const float4* __restrict__ src, int iters, float a0) {
    __shared__ __attribute__((aligned(16))) float lds[256 * 36 + 64];
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int tid = threadIdx.x;
    for (int i = tid; i < 256 * 36; i += 256) lds[i] = 1.f;
    __syncthreads();
    float a = a0 + tid * 1e-9f, b = a0;
    f32x4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (f32x4){a, b, a, b};
    const float* rp = lds + (tid & 63) * 36;
    float* wp = lds + tid * 36;
    const float4* gp = src + tid;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v[u & 3][u >> 2 & 3], b, acc, 0, 0, 0);
            if (u < N) {
                if (KIND == 1) v[u & 3] = *reinterpret_cast<const f32x4*>(rp + ((u * 4) & 31));
                if (KIND == 2) *reinterpret_cast<f32x4*>(wp + ((u * 4) & 31)) = v[u & 3];
                if (KIND == 3) { float4 g = gp[(u & 7) * 256]; v[u & 3] = (f32x4){g.x, g.y, g.z, g.w}; }
                if (KIND == 4) __syncthreads();
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[r];
    if (s == 12345.678f) out[tid] = s + lds[tid];
}

template <int KIND, int N>
void run(float* d, float4* src) {
    int iters = 2000, wps = 4;
    dim3 grid(256 * wps), block(256);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, N>), grid, block, 0, 0, d, src, 50, 1.f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, N>), grid, block, 0, 0, d, src, iters, 1.f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid.x * 4 * iters * 16 * 4096.0;
    static double base = 0; double cyc_per_iter = ms * 1e-3 * 2.4e9 / iters / wps;      // MFMA-pipe cycles per wave-iteration (1024 ideal)
    if (KIND == 0) base = cyc_per_iter;
    printf("kind=%d N=%2d: %.3f ms  %.1f TFLOP/s  pipe cycles/iter %.0f  (+%.1f per instr)\n", KIND, N, ms,
           flops / ms / 1e9, cyc_per_iter, N ? (cyc_per_iter - base) / N : 0.0);
}

int main() {
    float* d; (void)hipMalloc(&d, 4096);
    float4* src; (void)hipMalloc(&src, 256 * 8 * 16 + 4096); (void)hipMemset(src, 0, 256 * 8 * 16 + 4096);
    run<0, 0>(d, src);
    run<1, 8>(d, src); run<1, 16>(d, src);
    run<2, 4>(d, src); run<2, 8>(d, src);
    run<3, 4>(d, src); run<3, 8>(d, src);
    run<4, 2>(d, src); run<4, 4>(d, src);
    return 0;
}
