// Micro-benchmark: the conv K-tile instruction mix without address arithmetic (gfx950).
// Per K tile: 8 ds_read_b128 feeding 16 dependent MFMAs, 4 global 16-byte loads, 4 ds_write_b128, NBAR barriers.
// 256-thread blocks, 18 KB LDS (8 blocks per CU like the conv engine), grid = one resident round.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NBAR, int LOADS, int WRITES, int READS>
__global__ __launch_bounds__(256) void k(float* out, const float4* __restrict__ src, int iters, int span) {
    __shared__ __attribute__((aligned(16))) float lds[128 * 36];
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 128 * 36; i += 256) lds[i] = 1.f;
    __syncthreads();
    const float* ap = lds + ((wave >> 1) * 32 + (lane & 31)) * 36 + (lane >> 5) * 16;
    const float* bp = lds + (64 + (wave & 1) * 32 + (lane & 31)) * 36 + (lane >> 5) * 16;
    float* wp = lds + (tid >> 3) * 36 + (tid & 7) * 4;
    const float4* gp = src + (size_t)blockIdx.x * 64 + tid;
    float4 g[4];
    for (int it = 0; it < iters; ++it) {
        if (LOADS) {
#pragma unroll
            for (int j = 0; j < 4; ++j) g[j] = gp[(size_t)((it * 4 + j) & (span - 1)) * 65536];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            f32x4 a, b;
            if (READS) {
                a = *reinterpret_cast<const f32x4*>(ap + c * 4);
                b = *reinterpret_cast<const f32x4*>(bp + c * 4);
            } else {
                a = (f32x4){1.f, 1.f, 1.f, 1.f}; b = a;
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
        }
        if (NBAR >= 2) __syncthreads();
        if (WRITES) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 v = LOADS ? (f32x4){g[j].x + 1.f, g[j].y + 1.f, g[j].z + 1.f, g[j].w + 1.f} : (f32x4){1.f, 1.f, 1.f, 1.f};
                *reinterpret_cast<f32x4*>(wp + (j & 3) * 32 * 36) = v;
            }
        }
        if (NBAR >= 1) __syncthreads();
    }
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[r];
    if (s == 12345.678f) out[tid] = s + lds[tid];
}

template <int NBAR, int LOADS, int WRITES, int READS>
void run(float* d, float4* src, int span, const char* what) {
    int iters = 1000, wps = 8;
    dim3 grid(256 * wps), block(256);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NBAR, LOADS, WRITES, READS>), grid, block, 0, 0, d, src, 50, span);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NBAR, LOADS, WRITES, READS>), grid, block, 0, 0, d, src, iters, span);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid.x * 4 * iters * 16 * 4096.0;
    printf("%-46s %.3f ms  %.1f TFLOP/s\n", what, ms, flops / ms / 1e9);
}

int main() {
    float* d; (void)hipMalloc(&d, 4096);
    size_t n4 = (size_t)65536 * 64 + 2048 * 64 + 4096;      // span up to 64 x 1 MiB strides
    float4* src; (void)hipMalloc(&src, n4 * 16); (void)hipMemset(src, 0, n4 * 16);
    run<0, 0, 0, 0>(d, src, 1, "mfma only");
    run<0, 0, 0, 1>(d, src, 1, "+ 8 ds_read_b128");
    run<2, 0, 0, 1>(d, src, 1, "+ 8 ds_read + 2 barriers");
    run<2, 0, 1, 1>(d, src, 1, "+ 8 ds_read + 4 ds_write + 2 barriers");
    run<2, 1, 1, 1>(d, src, 1, "+ loads (same 1 KB per block: L1/L2 hits)");
    run<2, 1, 1, 1>(d, src, 64, "+ loads (64 MiB footprint: L2/HBM stream)");
    run<1, 1, 1, 1>(d, src, 64, "same with 1 barrier");
    run<2, 1, 0, 1>(d, src, 64, "loads, no ds_write");
    return 0;
}
