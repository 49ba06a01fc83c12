// Shared helpers for libstabstitch_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/stabstitch_hip.h"

#define SS_GRID_H 6
#define SS_GRID_W 8
#define SS_NV 63   // (SS_GRID_H+1)*(SS_GRID_W+1) control points
#define SS_NT 66   // SS_NV + 3 TPS coefficients per coordinate

static inline int ss_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SS_OK : SS_ERR_LAUNCH;
}

static inline int ss_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// exact n / d for n < 65536, 1 <= d < 65536, branch free:  q = umulhi(n, mul) + (n & mask)
//   d > 1: mul = ceil(2^32 / d), mask = 0;   d == 1: mul = 0, mask = ~0 (identity)
struct SsFastDiv {
    uint32_t mul, mask;
};
static inline SsFastDiv ss_fastdiv_make(uint32_t d) {
    SsFastDiv f;
    if (d <= 1) { f.mul = 0u; f.mask = 0xFFFFFFFFu; }
    else { f.mul = (uint32_t)(((1ull << 32) + d - 1) / d); f.mask = 0u; }
    return f;
}
__device__ __forceinline__ uint32_t ss_fastdiv(uint32_t n, SsFastDiv f) { return __umulhi(n, f.mul) + (n & f.mask); }

// exact n / d for every 32-bit n (Granlund-Montgomery round-up method): t = umulhi(n, mul); q = (t + ((n - t) >> sh1)) >> sh2
struct SsDiv32 {
    uint32_t mul, sh1, sh2;
};
static inline SsDiv32 ss_div32_make(uint32_t d) {
    SsDiv32 f;
    if (d <= 1) { f.mul = 0u; f.sh1 = 0u; f.sh2 = 0u; return f; }
    const uint32_t l = 32u - (uint32_t)__builtin_clz(d - 1u);          // ceil(log2 d)
    f.mul = (uint32_t)(((((1ull << l) - d) << 32) / d) + 1ull);
    f.sh1 = 1u;
    f.sh2 = l - 1u;
    return f;
}
__device__ __forceinline__ uint32_t ss_div32(uint32_t n, SsDiv32 f) {
    const uint32_t t = __umulhi(n, f.mul);
    return (t + ((n - t) >> f.sh1)) >> f.sh2;
}

__device__ __forceinline__ float ss_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float ss_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float ss_wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}

#ifdef SS_TUNING
// tools/ build only: buffer for per-workgroup s_memtime stamps (set through ss_debug_ptr, conv.hip)
extern unsigned long long* ss_tuning_dbg;
#endif
