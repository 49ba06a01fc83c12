// Cost of the instruction kinds a single-wave fp64 elimination step is made of (gfx950), by s_memtime around unrolled runs.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/wave_ops_cost tools/micro/wave_ops_cost.hip && /tmp/wave_ops_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#define PIN() do { _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) { asm volatile("" : "+v"(a[i_])); asm volatile("" : "+v"(b[i_])); } asm volatile("" : "+v"(f)); } while (0)
#define STAMP(t) do { PIN(); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_nop 0" ::: "memory"); t = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); PIN(); } while (0)

__device__ __forceinline__ double rl(double v, int l) {
    long long b = __double_as_longlong(v);
    unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), l);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

__global__ __launch_bounds__(256) void probe(double* out, long long* ticks, int lane_sel, int nwaves_active) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= nwaves_active) return;
    double a[16], b[16];
    for (int i = 0; i < 16; ++i) { a[i] = 1.0 + lane * 0.001 + i; b[i] = 2.0 + i * 0.5 + lane; }
    double f = 0.5 + lane * 1e-3;
    const int l = __builtin_amdgcn_readfirstlane(lane_sel);
    unsigned long long t0, t1;
    long long res[12];
    // (0) empty
    STAMP(t0); STAMP(t1); res[0] = t1 - t0;
    // (1) 16 x (2 readlane + fma with the SGPR pair)  = the update of 16 columns, one row per lane
    STAMP(t0);
#pragma unroll
    for (int i = 0; i < 16; ++i) { const double p = rl(a[i], l); a[i] = fma(-f, p, a[i]); }
    STAMP(t1); res[1] = t1 - t0;
    // (2) the same with two rows per lane
    STAMP(t0);
#pragma unroll
    for (int i = 0; i < 16; ++i) { const double p = rl(a[i], l); a[i] = fma(-f, p, a[i]); b[i] = fma(-f, p, b[i]); }
    STAMP(t1); res[2] = t1 - t0;
    // (3) 32 readlanes alone (results summed on the scalar side so they stay)
    STAMP(t0);
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) { long long bb = __double_as_longlong(a[i]); acc ^= (unsigned)__builtin_amdgcn_readlane((int)(unsigned)bb, l) ^ (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(bb >> 32), l); }
    asm volatile("" : "+v"(acc));
    STAMP(t1); res[3] = t1 - t0;
    // (4) 32 independent fma_f64, VGPR operands
    STAMP(t0);
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = fma(-f, b[i], a[i]); }
#pragma unroll
    for (int i = 0; i < 16; ++i) { b[i] = fma(f, a[(i + 5) & 15], b[i]); }
    STAMP(t1); res[4] = t1 - t0;
    // (5) 16 dependent fma_f64
    STAMP(t0);
    double c = a[0];
#pragma unroll
    for (int i = 0; i < 16; ++i) c = fma(c, f, b[i]);
    asm volatile("" : "+v"(c));
    STAMP(t1); res[5] = t1 - t0;
    // (6) one fp64 division
    STAMP(t0);
    double d = 1.0 / c;
    asm volatile("" : "+v"(d));
    STAMP(t1); res[6] = t1 - t0;
    // (7) 16 x (ds_bpermute pair + fma): the broadcast through the LDS crossbar instead of readlane
    STAMP(t0);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        long long bb = __double_as_longlong(a[i]);
        unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute(l * 4, (int)(unsigned)bb), hi = (unsigned)__builtin_amdgcn_ds_bpermute(l * 4, (int)(unsigned)(bb >> 32));
        const double p = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
        a[i] = fma(-f, p, a[i]);
    }
    STAMP(t1); res[7] = t1 - t0;
    // (8) key reduction: 4 DPP max + 4 readlanes + scalar max
    STAMP(t0);
    unsigned key = (unsigned)(__double_as_longlong(a[3]) >> 38);
    key = max(key, (unsigned)__builtin_amdgcn_update_dpp(0, (int)key, 0x121, 0xF, 0xF, false));
    key = max(key, (unsigned)__builtin_amdgcn_update_dpp(0, (int)key, 0x122, 0xF, 0xF, false));
    key = max(key, (unsigned)__builtin_amdgcn_update_dpp(0, (int)key, 0x124, 0xF, 0xF, false));
    key = max(key, (unsigned)__builtin_amdgcn_update_dpp(0, (int)key, 0x128, 0xF, 0xF, false));
    unsigned km = max(max((unsigned)__builtin_amdgcn_readlane((int)key, 0), (unsigned)__builtin_amdgcn_readlane((int)key, 16)),
                      max((unsigned)__builtin_amdgcn_readlane((int)key, 32), (unsigned)__builtin_amdgcn_readlane((int)key, 48)));
    asm volatile("" :: "s"(km));
    STAMP(t1); res[8] = t1 - t0;
    // (9) LDS: one lane writes 16 doubles, all lanes read them back (broadcast), then 16 fma
    __shared__ double row[4][16];
    STAMP(t0);
    if (lane == l) {
#pragma unroll
        for (int i = 0; i < 16; ++i) row[wave][i] = a[i];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) { const double p = row[wave][i]; a[i] = fma(-f, p, a[i]); }
    STAMP(t1); res[9] = t1 - t0;
    // (10) barrier alone
    STAMP(t0);
    __syncthreads();
    STAMP(t1); res[10] = t1 - t0;
    // (11) LDS write + barrier + LDS read of one value (the publish / consume round trip)
    __shared__ double cell[64];
    STAMP(t0);
    cell[lane] = c;
    __syncthreads();
    double e = cell[(lane + 1) & 63];
    asm volatile("" : "+v"(e));
    STAMP(t1); res[11] = t1 - t0;
    double s = c + d + e + (double)acc + (double)km;
    for (int i = 0; i < 16; ++i) s += a[i] + b[i];
    out[threadIdx.x] = s;
    if (lane == 0) for (int i = 0; i < 12; ++i) ticks[wave * 12 + i] = res[i];
}

int main() {
    double* out; long long* ticks;
    hipMalloc(&out, 256 * 8); hipMalloc(&ticks, 4 * 12 * 8);
    const char* names[12] = {"empty (stamp overhead)", "16 x (readlane pair + fma f64 with SGPR operand)", "16 x (readlane pair + 2 fma)", "32 readlanes",
                             "32 independent fma f64 (VGPR)", "16 dependent fma f64", "1 fp64 division", "16 x (ds_bpermute pair + fma)",
                             "key reduction (4 DPP max, 4 readlane)", "LDS row broadcast (1 lane writes 16, all read) + 16 fma", "barrier", "LDS write + barrier + LDS read"};
    for (int nw = 1; nw <= 4; nw *= 4) {
        long long h[48];
        for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, 0, out, ticks, 37, nw); hipDeviceSynchronize(); }
        hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
        printf("%d wave(s) active; s_memtime ticks, wave 0 (stamp overhead subtracted):\n", nw);
        for (int i = 0; i < 12; ++i) printf("  %-60s %6lld\n", names[i], h[i] - (i ? h[0] : 0));
    }
    // tick rate: a long kernel timed by events
    return 0;
}
