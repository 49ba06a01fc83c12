// Micro-benchmark: how should the next K tile travel global -> LDS beside fp32 MFMAs (gfx950)?
// Per K tile: 8 ds_read_b128 -> 16 dependent MFMAs, 16 bytes x 4 per thread global -> LDS by MODE:
//   0 none (LDS constant)      1 buffer_load -> VGPR -> ds_write_b128 (loads at the top of the tile)
//   2 same, loads issued one tile ahead (2 register sets)   3 global_load_lds_dwordx4 (double-buffered LDS)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk_rsrc(const void* base, unsigned bytes) {
    unsigned long long a = (unsigned long long)base;
    unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    void* ub = (void*)(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(ub, 0, (int)bytes, 0x00020000);
}

template <int MODE>
__global__ __launch_bounds__(256) void stage_kernel(float* out, const float* __restrict__ src, unsigned bytes, int iters, unsigned share) {
    constexpr int NB = MODE == 3 ? 2 : 1;
    constexpr int LD = MODE == 3 ? 32 : 36;
    __shared__ __attribute__((aligned(16))) float lds[NB][128 * LD];
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < NB * 128 * LD; i += 256) (&lds[0][0])[i] = 1.f;
    __syncthreads();
    const int arow = (wave >> 1) * 32 + (lane & 31), brow = 64 + (wave & 1) * 32 + (lane & 31);
    const int half = lane >> 5;
    const __amdgpu_buffer_rsrc_t rs = mk_rsrc(src, bytes);
    // thread's 4 rows: tid>>3 + 32 j, 16-byte column tid&7
    const unsigned bsrc = blockIdx.x % share;
    unsigned goff = (bsrc * 128u + (tid >> 3)) * 4096u + (tid & 7) * 16u;     // row pitch 4 KB in the source
    u32x4 g[2][4];
    if (MODE == 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) g[0][j] = __builtin_amdgcn_raw_buffer_load_b128(rs, goff + j * 32u * 4096u, 0, 0);
        goff += 128u;
    }
    for (int it = 0; it < iters; ++it) {
        const int buf = MODE == 3 ? (it & 1) : 0;
        const int cur = MODE == 2 ? (it & 1) : 0;
        if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) g[0][j] = __builtin_amdgcn_raw_buffer_load_b128(rs, goff + j * 32u * 4096u, 0, 0);
        }
        if (MODE == 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) g[cur ^ 1][j] = __builtin_amdgcn_raw_buffer_load_b128(rs, goff + j * 32u * 4096u, 0, 0);
        }
        if (MODE == 3) {
            // wave w stages rows [32w, 32w+32): 4 instructions of 8 rows x 128 B; lane -> row (lane>>3), chunk (lane&7)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = wave * 32 + j * 8 + (lane >> 3);
                const int chunk = (lane & 7) ^ ((row >> 1) & 7);                      // source-side XOR swizzle
                const float* gp = src + ((size_t)(bsrc * 128u + row) * 1024u + (it & 31) * 32u + chunk * 4);
#if defined(__HIP_DEVICE_COMPILE__)
                __builtin_amdgcn_global_load_lds(gp, &lds[buf ^ 1][(wave * 32 + j * 8) * 32], 16, 0, 0);
#endif
            }
        }
        goff = (goff & ~4095u) | ((goff + 128u) & 4095u);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            f32x4 a, b;
            if (MODE == 3) {
                const int ca = ((half * 4 + c) ^ ((arow >> 1) & 7)) * 4, cb = ((half * 4 + c) ^ ((brow >> 1) & 7)) * 4;
                a = *reinterpret_cast<const f32x4*>(&lds[buf][arow * 32 + ca]);
                b = *reinterpret_cast<const f32x4*>(&lds[buf][brow * 32 + cb]);
            } else {
                a = *reinterpret_cast<const f32x4*>(&lds[0][arow * 36 + half * 16 + c * 4]);
                b = *reinterpret_cast<const f32x4*>(&lds[0][brow * 36 + half * 16 + c * 4]);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
        }
        if (MODE == 1 || MODE == 2) {
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<u32x4*>(&lds[0][((tid >> 3) + 32 * j) * 36 + (tid & 7) * 4]) = g[cur][j];
        }
        __syncthreads();
    }
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[r];
    if (s == 12345.678f) out[tid] = s + lds[0][tid];
}

template <int MODE>
void run(float* d, float* src, unsigned bytes, const char* what, unsigned share) {
    int iters = 1000;
    dim3 grid(256 * (MODE == 3 ? 5 : 8)), block(256);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((stage_kernel<MODE>), grid, block, 0, 0, d, src, bytes, 50, share);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((stage_kernel<MODE>), grid, block, 0, 0, d, src, bytes, iters, share);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid.x * 4 * iters * 16 * 4096.0;
    printf("share=%4u %-64s %.3f ms  %.1f TFLOP/s\n", share, what, ms, flops / ms / 1e9);
}

int main() {
    float* d; (void)hipMalloc(&d, 4096);
    size_t bytes = (size_t)2048 * 128 * 4096;      // 1 GiB: 2048 blocks x 128 rows x 4 KB
    float* src; (void)hipMalloc(&src, bytes + 65536); (void)hipMemset(src, 0, bytes + 65536);
    for (unsigned share : {8u, 64u, 2048u}) {
        run<0>(d, src, (unsigned)bytes, "no staging", share);
        run<1>(d, src, (unsigned)bytes, "buffer_load -> VGPR -> ds_write_b128, loads at tile top", share);
        run<2>(d, src, (unsigned)bytes, "same, loads one tile ahead", share);
        run<3>(d, src, (unsigned)bytes, "global_load_lds_dwordx4, 2 LDS buffers, XOR-swizzled source", share);
    }
    return 0;
}
