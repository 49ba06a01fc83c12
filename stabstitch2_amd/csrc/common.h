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

// overflow watcher of a streaming canvas (geom.hip: ss_canvas_watch has the state's layout); used by geom.hip and render.hip
__device__ __forceinline__ void canvas_watch_update(float xmin, float xmax, float ymin, float ymax, bool bad, float guard, int* wi,
                                                    float* wf) {
    xmin = ss_wave_min(xmin); xmax = ss_wave_max(xmax); ymin = ss_wave_min(ymin); ymax = ss_wave_max(ymax);
    // fminf / fmaxf drop a NaN operand: a NaN control point would vanish from the extremes, so it is carried as a flag of its own
    const bool anybad = __builtin_amdgcn_ballot_w64(bad) != 0ull;
    if (threadIdx.x == 0) {
        const float lo = fminf(xmin, ymin), hi = fmaxf(xmax, ymax);
        const int seen = wi[0];
        // (half a pixel of a 4096-wide canvas: the canvas is the first window's OWN bbox when margin = 0, its extremes sit on +-1;
        // `near` gets the same slack when the guard is smaller than it, else fp32 rounding alone would ask for a growth)
        const float slack = 2.5e-4f;
        const bool out = anybad || lo < -1.0f - slack || hi > 1.0f + slack;
        const float g = guard > slack ? guard : -slack;
        const bool near = out || lo < -1.0f + g || hi > 1.0f - g;
        if (out) { wi[1] += 1; if (wi[2] < 0) wi[2] = seen; }
        if (near) wi[3] += 1;
        wi[0] = seen + 1;
        wf[0] = fminf(wf[0], xmin); wf[1] = fmaxf(wf[1], xmax); wf[2] = fminf(wf[2], ymin); wf[3] = fmaxf(wf[3], ymax);
    }
}


#ifdef SS_TUNING
// tools/ build only: buffer for per-workgroup s_memtime stamps (set through ss_debug_ptr, conv.hip)
extern unsigned long long* ss_tuning_dbg;
#endif
