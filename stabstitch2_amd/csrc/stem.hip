// The trunk's stem in ONE kernel: Conv2d(3, 64, 7, stride 2, pad 3) + folded BatchNorm + ReLU + MaxPool2d(3, 2, 1)
// (spatial_network.py:127-130, temporal_network.py:47-50), for `groups` filter banks that read the same frames (the
// SpatialNet and TemporalNet stems of the shared-stem path) -- fp32 MFMA, gfx950.
//
// Why: as two kernels (ss_conv_stem3 + ss_maxpool_nhwc) the conv writes its 180 x 240 x 128-channel output (1.4 GB per 64
// images) and the pool reads it back: 19 % of all HBM bytes of a clip for 0.47 ms of pool launches, on top of a conv whose
// implicit-GEMM workgroups walk only six K tiles each (prologue + epilogue = 35 % of a workgroup's life, MFMA pipe 57 % busy).
// Here a workgroup owns a 2-D tile of POOLED pixels, computes the conv outputs that feed it (one ring of halo) and pools them
// in its epilogue: the un-pooled map never exists in memory.
//
//   workgroup = 256 threads = 4 waves; tile = 9 x 12 pooled pixels x 64 channels of ONE image and ONE filter bank;
//   conv region = 19 x 25 pixels (rows 2 py0 - 1 .. 2 py0 + 17, columns 2 px0 - 1 .. 2 px0 + 23) = 475 GEMM rows, padded to
//   16 row tiles of 32 (the map 90 x 120 is 10 x 10 tiles exactly; 512 computed rows per 432 conv pixels a non-overlapping
//   tiling would need: 1.185x the MFMA work, the price of never writing the un-pooled map);
//   K = 7 filter rows x 22 (21 = 7 taps x 3 channels + 1 zero weight) = 154 in 21 groups (8 + 8 + 6 k per row: 77 MFMA steps per
//   tile instead of the 84 of a 24-padded row), on the row-packed 3-channel
//   frames of ss_nchw_to_nhwc3_padded (a filter row is 24 contiguous floats of an image row);
//   wave w owns row tiles 4 w .. 4 w + 3 x both 32-channel halves: 8 accumulator tiles of v_mfma_f32_32x32x2_f32 = 128 registers.
//   * the 43 x 168-float input patch is staged once into LDS (coalesced 16-byte buffer loads; rows outside the image and the
//     one column left of it come back as zeros / harmless neighbours through the descriptor's bounds check -- they only feed conv
//     pixels outside the map, which the pool skips);
//   * K loop: per group of 8 k a lane reads its 4 A values per row tile straight from the patch (k is permuted so that lane half
//     h takes k = 8 g + 4 h + s: one 16-byte LDS read per row tile feeds 8 MFMAs; pixels are 24 bytes apart: conflict free) and
//     its 4 B values per channel half from filters PRE-PACKED in that register layout (ss_stem_pool_pack; 1 KB coalesced loads,
//     L2 resident): 10 loads and no VALU instruction per 32 MFMAs;
//   * epilogue, per 32-channel half: accumulators -> LDS [conv pixel][32] (aliases the patch), then every thread pools 3 x 3
//     windows of (pixel, 4 channels) with 16-byte LDS reads, adds the folded-BN bias, applies ReLU and stores 16 bytes.
//     max(relu(x + b)) == relu(max(x) + b) bit for bit (rounding is monotone), so bias / ReLU run on the pooled values.
// Results equal conv + ReLU + pool of the two-kernel path up to the summation order of the K axis (fp32 rounding).
#include "common.h"
#include <type_traits>

typedef float sp_f32x16 __attribute__((ext_vector_type(16)));
typedef float sp_f32x4 __attribute__((ext_vector_type(4)));
typedef float sp_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned sp_u32x4 __attribute__((ext_vector_type(4)));

#define SP_PR 9                          // pooled rows / columns of a workgroup tile
#define SP_PC 12
#define SP_CR (2 * SP_PR + 1)            // conv rows / columns feeding it
#define SP_CC (2 * SP_PC + 1)
#define SP_M (SP_CR * SP_CC)             // 475 GEMM rows
#define SP_PROWS (2 * SP_CR + 5)         // 43 input rows
#define SP_PFLOATS (6 * (SP_CC - 1) + 24)    // 168 floats of an input row
#define SP_PITCH 172                     // LDS pitch of a patch row (floats; 16-byte multiple)
#define SP_NG 21                         // K groups of 8
#ifndef SP_SPLIT_DEFAULT
#define SP_SPLIT_DEFAULT true            // which of the two kernels ss_stem_pool launches (measured: see stem_pool_kernel_half)
#endif

struct StemP {
    const float* in;             // [n][h][w + 8][3]
    const float* packed;         // [groups][2][21][64][4]
    const float* bias;           // [groups][64] or nullptr
    float* out;                  // [groups][n][hp][wp][64]
    int n, h, w, ho, wo, hp, wp, groups;
    unsigned ntx, nty, ntiles;
    SsDiv32 divTx, divTy, divG, divG2;
    long long out_gs;
    unsigned in_bytes, pk_bytes;
#ifdef SS_TUNING
    unsigned long long* dbg;     // per-workgroup phase stamps (tools/diag_stem.py)
    int stagger;
#endif
};

#ifdef SS_TUNING
extern int g_wino_lds_pad;
extern int g_wino_knob[4];
#define SP_STAMP(i) do { if (p.dbg) ts[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SP_STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ __amdgpu_buffer_rsrc_t sp_rsrc(const float* base, unsigned bytes) {
    unsigned long long a = (unsigned long long)base;
    unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    void* ub = (void*)(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(ub, 0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

// Staged accumulators [conv pixel][32 channels]: a row is 32 dwords = half of the 64 LDS banks, so in the pool's 16-byte reads rows of
// equal parity collide (round 4: 0.41 of this kernel's LDS cycles were bank conflicts, all of them there: every lane group read the
// same 16 dwords of two even rows).  With row m stored at m ^ bit 2 of m (rows swap inside pairs when bit 2 is set) two rows exactly 4
// apart always differ in parity; the pool's lane map pairs such rows in every lane group.
__device__ __forceinline__ int sp_row(int m) { return m ^ ((m >> 2) & 1); }

__global__ __launch_bounds__(256, 2) void stem_pool_kernel(StemP p) {
    // the input patch [43][172] during the K loop; the staged accumulators [512][32] in the epilogue
    __shared__ __attribute__((aligned(16))) float smem[512 * 32];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h2 = lane >> 5, ln = lane & 31;
#ifdef SS_TUNING
    unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    SP_STAMP(0);
    // prologue and epilogue are VALU / LDS work that shares its SIMD with the OTHER resident workgroup's dense MFMA stream: at
    // equal priority each of their instructions waits for an MFMA boundary (64 clocks).  Raised priority lets them through.
    __builtin_amdgcn_s_setprio(3);
#ifdef SS_TUNING
    // experiment (ss_debug_set key 16, units of 1024 clocks): the second resident workgroup of a CU (LDS base != 0) sleeps in the
    // FIRST round so that the two run in anti-phase (all workgroups have the same length: left alone they stay in lock-step)
    if (p.stagger > 0 && blockIdx.x < 512u && (__builtin_amdgcn_s_getreg((31 << 11) | 6) & 0x1FFu) != 0u) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)p.stagger * 1024ull) __builtin_amdgcn_s_sleep(32);
    }
#endif

    // consecutive workgroups = the filter banks of one tile (they read the same patch), then the next tile of the image
    const unsigned lin = blockIdx.x;
    const unsigned tile = ss_div32(lin, p.divG);
    const unsigned grp = lin - tile * (unsigned)p.groups;
    const unsigned t1 = ss_div32(tile, p.divTx);
    const int tx = (int)(tile - t1 * p.ntx);
    const unsigned img = ss_div32(t1, p.divTy);
    const int ty = (int)(t1 - img * p.nty);
    const int py0 = ty * SP_PR, px0 = tx * SP_PC;
    const int cy0 = 2 * py0 - 1, cx0 = 2 * px0 - 1;

    const __amdgpu_buffer_rsrc_t rin = sp_rsrc(p.in, p.in_bytes);
    const __amdgpu_buffer_rsrc_t rpk = sp_rsrc(p.packed, p.pk_bytes);

    // ---- stage the input patch: rows 2 cy0 - 3 .. + 42 of the image, floats 6 cx0 .. + 167 of the padded 3-channel row.
    // Thread = (row slot, float4 of the row): 6 rows x 42 float4 per pass (252 threads), 8 passes; one division per thread, the
    // rest is adds (prologue / epilogue instructions are paid for at the neighbouring workgroup's MFMA cadence: few of them).
    {
        constexpr int Q = SP_PFLOATS / 4;                   // 42 float4 per row
        const int slot = tid / Q, q = tid - slot * Q;       // slot 0..5 (6: idle threads 252..255)
        const long long rowf = (long long)(p.w + 8) * 3;
        const int iy0 = 2 * cy0 - 3 + slot;
        long long off = ((long long)img * p.h + iy0) * rowf + 6 * cx0 + 4 * q;      // floats; < 0 left of the very first row
        float* dst = smem + slot * SP_PITCH + 4 * q;
#pragma unroll
        for (int i = 0; i < (SP_PROWS + 5) / 6; ++i) {
            const int pr = slot + 6 * i, iy = iy0 + 6 * i;
            const bool ok = slot < 6 && pr < SP_PROWS && (unsigned)iy < (unsigned)p.h && off >= 0;
            const sp_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rin, ok ? (unsigned)(off * 4) : 0xFFFFFFFFu, 0, 0);
            if (slot < 6 && pr < SP_PROWS) *reinterpret_cast<sp_u32x4*>(dst) = v;
            off += 6 * rowf;
            dst += 6 * SP_PITCH;
        }
    }

    // this lane's GEMM rows: row tile mt of the wave -> conv pixel (r, c) of the region -> first float of its window in the patch
    // (two bases: the 8-k groups read 4 floats at + 4 h, the 6-k group of a filter row 3 floats at 16 + 3 h)
    int abase[4], abase3[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        int m = 32 * (4 * wave + mt) + ln;
        m = m < SP_M ? m : SP_M - 1;                        // rows 475 .. 511: idle copies of the last pixel
        const int r = m / SP_CC, c = m - r * SP_CC;
        abase[mt] = (2 * r) * SP_PITCH + 6 * c + 4 * h2;
        abase3[mt] = (2 * r) * SP_PITCH + 6 * c + 16 + 3 * h2;
    }
    const unsigned pk_lane = (unsigned)lane * 16u;
    const unsigned pk_grp = grp * (2u * SP_NG * 1024u);

    sp_f32x16 acc[4][2];
    const sp_f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    __syncthreads();
    SP_STAMP(1);
    __builtin_amdgcn_s_setprio(0);

    // ---- K loop: 21 groups of 8 k; group g = filter row g / 3, floats 8 (g % 3) .. + 7 of its 24
    sp_f32x4 a_cur[4], b_cur[2], a_nxt[4], b_nxt[2];
    auto load = [&](int g, sp_f32x4 (&a)[4], sp_f32x4 (&b)[2]) {
        if (g % 3 == 2) {           // the 21 real floats of a filter row end with a group of SIX k: three MFMA steps (k = 16 + 3 h + s)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const float* ap = smem + abase3[mt] + (g / 3) * SP_PITCH;
                a[mt] = (sp_f32x4){ap[0], ap[1], ap[2], 0.f};
            }
        } else {
            const int koff = (g / 3) * SP_PITCH + (g % 3) * 8;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const float* ap = smem + abase[mt] + koff;
                const sp_f32x2 lo = *reinterpret_cast<const sp_f32x2*>(ap);
                const sp_f32x2 hi = *reinterpret_cast<const sp_f32x2*>(ap + 2);
                a[mt] = (sp_f32x4){lo[0], lo[1], hi[0], hi[1]};
            }
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int so = (int)__builtin_amdgcn_readfirstlane(pk_grp + (unsigned)(nt * SP_NG + g) * 1024u);
            b[nt] = __builtin_bit_cast(sp_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rpk, pk_lane, so, 0));
        }
    };
    load(0, a_cur, b_cur);
#pragma unroll
    for (int g = 0; g < SP_NG; ++g) {
        // the next group's operands are requested BEFORE this group's 32 MFMAs (pinned: left alone, hipcc sinks the loads to
        // the end of the group, right in front of their first use, and every group pays an L2 round trip)
        if (g + 1 < SP_NG) load(g + 1, a_nxt, b_nxt);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < (g % 3 == 2 ? 3 : 4); ++s)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    // the very first product of a tile starts from the instruction's inline 0 (no 128 v_mov to clear the tiles)
                    const sp_f32x16 c0 = (g == 0 && s == 0) ? zero16 : acc[mt][nt];
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[mt][s], b_cur[nt][s], c0, 0, 0, 0);
                }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) a_cur[mt] = a_nxt[mt];
        b_cur[0] = b_nxt[0];
        b_cur[1] = b_nxt[1];
    }

    SP_STAMP(2);
    __builtin_amdgcn_s_setprio(3);
    // ---- epilogue: per 32-channel half stage the accumulators as [conv pixel][32], pool 3 x 3 / stride 2, bias, ReLU, store
    __syncthreads();                                        // every wave is done with the patch (the stage aliases it)
    SP_STAMP(3);
    float* __restrict__ out = p.out + (long long)grp * p.out_gs;
    // a tile is INTERIOR when every tap of every pooled pixel lies inside the conv map (81 of the 100 tiles of a 90 x 120 map):
    // no padding logic at all.  Border tiles replace the taps outside the map (row / column -1 of the image, or beyond a partial
    // last tile) by -inf, MaxPool2d's padding.  Either way branch free per tap: all nine taps lie inside the staged region, the
    // nine 16-byte LDS reads of an item go out back to back (with a branch per tap they went one at a time: 59k clocks of epilogue
    // beside 45k of K loop), and the 3 x 3 maximum is four v_max3_f32 per channel.
    const bool interior = cy0 >= 0 && cx0 >= 0 && cy0 + SP_CR <= p.ho && cx0 + SP_CC <= p.wo;
    // a thread's items all have the same channel quad (item & 7 = tid & 7): its two bias quads are loaded once
    sp_f32x4 bias4[2] = {(sp_f32x4){0.f, 0.f, 0.f, 0.f}, (sp_f32x4){0.f, 0.f, 0.f, 0.f}};
    if (p.bias) {
        bias4[0] = *reinterpret_cast<const sp_f32x4*>(p.bias + grp * 64 + 4 * (tid & 7));
        bias4[1] = *reinterpret_cast<const sp_f32x4*>(p.bias + grp * 64 + 32 + 4 * (tid & 7));
    }
    auto pool = [&](int nt, auto masked) {
        constexpr bool MASKED = decltype(masked)::value;
#pragma unroll
        for (int it = 0; it < (SP_PR * SP_PC * 8 + 255) / 256; ++it) {
            const int item = tid + 256 * it;
            const int q = item & 7;
            int pp = item >> 3;
            // a ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}: channel quads 0-3 of the pixels
            // of lanes 0-7 and 24-31 with quads 4-7 of those of lanes 8-15 and 16-23 (and vice versa).  Pixels b, b + 1, b + 3, b + 2
            // in that lane order (the last two of every four swapped; SP_PC = 12 is a multiple of 4) pair conv pixels exactly 4
            // apart -- with the staged rows' parity swizzle (sp_row) each pair lands on different bank halves, all nine taps alike
            pp ^= (pp >> 1) & 1;
            pp = pp < SP_PR * SP_PC ? pp : SP_PR * SP_PC - 1;
            const int a = pp / SP_PC, b = pp - a * SP_PC;
            const int py = py0 + a, px = px0 + b;
            const int m0 = (2 * a) * SP_CC + 2 * b;
            sp_f32x4 v[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) v[t] = *reinterpret_cast<const sp_f32x4*>(smem + sp_row(m0 + (t / 3) * SP_CC + (t % 3)) * 32 + 4 * q);
            if constexpr (MASKED) {
                bool rok[3], cok[3];
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    rok[d] = (unsigned)(cy0 + 2 * a + d) < (unsigned)p.ho;
                    cok[d] = (unsigned)(cx0 + 2 * b + d) < (unsigned)p.wo;
                }
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const bool ok = rok[t / 3] && cok[t % 3];
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[t][k] = ok ? v[t][k] : -INFINITY;
                }
            }
            const sp_f32x4 bb = bias4[nt];
            sp_f32x4 mx;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float r0 = fmaxf(fmaxf(v[0][k], v[1][k]), v[2][k]);
                const float r1 = fmaxf(fmaxf(v[3][k], v[4][k]), v[5][k]);
                const float r2 = fmaxf(fmaxf(v[6][k], v[7][k]), v[8][k]);
                mx[k] = fmaxf(fmaxf(fmaxf(r0, r1), r2) + bb[k], 0.f);
            }
            if (item < SP_PR * SP_PC * 8 && py < p.hp && px < p.wp)
                *reinterpret_cast<sp_f32x4*>(out + (((long long)img * p.hp + py) * p.wp + px) * 64 + 32 * nt + 4 * q) = mx;
        }
    };
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * (4 * wave + mt) + 8 * (r >> 2) + (r & 3) + 4 * h2;
                smem[sp_row(row) * 32 + ln] = acc[mt][nt][r];
            }
        __syncthreads();
        if (nt == 0) SP_STAMP(4); else SP_STAMP(6);
        if (interior) pool(nt, std::false_type{}); else pool(nt, std::true_type{});
        if (nt == 0) SP_STAMP(5);
        if (nt == 0) __syncthreads();                       // the second half overwrites the stage
    }
#ifdef SS_TUNING
    if (p.dbg && tid == 0) {
        unsigned long long* d = p.dbg + (size_t)blockIdx.x * 10;
        for (int i = 0; i < 7; ++i) d[i] = ts[i];
        d[8] = __builtin_amdgcn_s_memtime();
    }
#endif
}

// The same stem with ONE 32-channel half per workgroup (round 5; the kernel ss_stem_pool launches: 1065 against 1088 us for 64 images
// x 2 banks, tools/ab_stem_split.py; the two-halves kernel above stays for the tuning build's A/B): 4 accumulator tiles per wave instead of 8
// (~120 registers: four waves per SIMD), a 32 KB stage (two 16-channel passes): four to five workgroups per CU instead of two, so
// that a workgroup's prologue (patch staging) and epilogue (pool) overlap THREE neighbours' K loops.  Price: every patch is staged by
// two workgroups (the halves of a tile are consecutive workgroups: the second finds it in L2) and an A operand read from LDS feeds one
// MFMA instead of two.  Same arithmetic per output: results equal stem_pool_kernel's bit for bit.
__global__ __launch_bounds__(256, 4) void stem_pool_kernel_half(StemP p) {
    __shared__ __attribute__((aligned(16))) float smem[512 * 16];           // the patch [43][172] (7396 floats), then the stage [512][16]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h2 = lane >> 5, ln = lane & 31;
    __builtin_amdgcn_s_setprio(3);
    // Workgroups are dealt round-robin to the 8 XCDs (each with its own L2).  The 2 x groups workgroups that read ONE patch (its
    // filter banks x channel halves) are consecutive workgroups OF ONE XCD: the first stages the patch from HBM, the others find it in
    // that XCD's L2 (dealt out in plain order they would sit on 2 x groups different XCDs and each fetch it again).
    const unsigned xcd = blockIdx.x & 7u, jx = blockIdx.x >> 3;
    const unsigned tq = ss_div32(jx, p.divG2), sub = jx - tq * 2u * (unsigned)p.groups;
    const unsigned tile = tq * 8u + xcd;
    if (tile >= p.ntiles) return;
    const unsigned nt0 = sub & 1u;                         // this workgroup's 32-channel half
    const unsigned grp = sub >> 1;
    const unsigned t1 = ss_div32(tile, p.divTx);
    const int tx = (int)(tile - t1 * p.ntx);
    const unsigned img = ss_div32(t1, p.divTy);
    const int ty = (int)(t1 - img * p.nty);
    const int py0 = ty * SP_PR, px0 = tx * SP_PC;
    const int cy0 = 2 * py0 - 1, cx0 = 2 * px0 - 1;
    const __amdgpu_buffer_rsrc_t rin = sp_rsrc(p.in, p.in_bytes);
    const __amdgpu_buffer_rsrc_t rpk = sp_rsrc(p.packed, p.pk_bytes);
    {
        constexpr int Q = SP_PFLOATS / 4;
        const int slot = tid / Q, q = tid - slot * Q;
        const long long rowf = (long long)(p.w + 8) * 3;
        const int iy0 = 2 * cy0 - 3 + slot;
        long long off = ((long long)img * p.h + iy0) * rowf + 6 * cx0 + 4 * q;
        float* dst = smem + slot * SP_PITCH + 4 * q;
#pragma unroll
        for (int i = 0; i < (SP_PROWS + 5) / 6; ++i) {
            const int pr = slot + 6 * i, iy = iy0 + 6 * i;
            const bool ok = slot < 6 && pr < SP_PROWS && (unsigned)iy < (unsigned)p.h && off >= 0;
            const sp_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rin, ok ? (unsigned)(off * 4) : 0xFFFFFFFFu, 0, 0);
            if (slot < 6 && pr < SP_PROWS) *reinterpret_cast<sp_u32x4*>(dst) = v;
            off += 6 * rowf;
            dst += 6 * SP_PITCH;
        }
    }
    // (the 6-k group of a filter row starts at abase + 16 - h2 = row base + 16 + 3 h2: derived where it is used -- four registers
    // less; at 128 registers for four workgroups per CU the allocator had parked seven values in scratch)
    int abase[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        int m = 32 * (4 * wave + mt) + ln;
        m = m < SP_M ? m : SP_M - 1;
        const int r = m / SP_CC, c = m - r * SP_CC;
        abase[mt] = (2 * r) * SP_PITCH + 6 * c + 4 * h2;
    }
    const unsigned pk_lane = (unsigned)lane * 16u;
    const unsigned pk_grp = grp * (2u * SP_NG * 1024u) + nt0 * (SP_NG * 1024u);
    sp_f32x16 acc[4];
    const sp_f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    __builtin_amdgcn_s_setprio(0);
    sp_f32x4 a_cur[4], b_cur, a_nxt[4], b_nxt;
    auto load = [&](int g, sp_f32x4 (&a)[4], sp_f32x4& b) {
        if (g % 3 == 2) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const float* ap = smem + abase[mt] + (16 - h2) + (g / 3) * SP_PITCH;
                a[mt] = (sp_f32x4){ap[0], ap[1], ap[2], 0.f};
            }
        } else {
            const int koff = (g / 3) * SP_PITCH + (g % 3) * 8;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const float* ap = smem + abase[mt] + koff;
                const sp_f32x2 lo = *reinterpret_cast<const sp_f32x2*>(ap);
                const sp_f32x2 hi = *reinterpret_cast<const sp_f32x2*>(ap + 2);
                a[mt] = (sp_f32x4){lo[0], lo[1], hi[0], hi[1]};
            }
        }
        const int so = (int)__builtin_amdgcn_readfirstlane(pk_grp + (unsigned)g * 1024u);
        b = __builtin_bit_cast(sp_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rpk, pk_lane, so, 0));
    };
    load(0, a_cur, b_cur);
#pragma unroll
    for (int g = 0; g < SP_NG; ++g) {
        if (g + 1 < SP_NG) load(g + 1, a_nxt, b_nxt);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < (g % 3 == 2 ? 3 : 4); ++s)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const sp_f32x16 c0 = (g == 0 && s == 0) ? zero16 : acc[mt];
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[mt][s], b_cur[s], c0, 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) a_cur[mt] = a_nxt[mt];
        b_cur = b_nxt;
    }
    __builtin_amdgcn_s_setprio(3);
    __syncthreads();
    // this image's pooled map as a buffer: the store address is one 32-bit offset per item (a 64-bit pointer per item was one of
    // the values parked in scratch), stores outside the map are dropped by the descriptor
    const __amdgpu_buffer_rsrc_t rout = sp_rsrc(p.out + (long long)grp * p.out_gs + (long long)img * p.hp * p.wp * 64,
                                                (unsigned)(p.hp * p.wp) * 256u);
    const bool interior = cy0 >= 0 && cx0 >= 0 && cy0 + SP_CR <= p.ho && cx0 + SP_CC <= p.wo;
    // The thread's identity is re-derived here (lane = v_mbcnt, wave from its SGPR) instead of being kept from the kernel's entry:
    // tid, lane & 31 and the row offsets derived from them were the values the allocator carried across the K loop in SCRATCH
    // (7 registers, 24 bytes per lane, a scratch_load inside the unrolled MFMA stream).
    const int elane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int etid = wave * 64 + elane;
    const int eh2 = elane >> 5, eln = elane & 31;
#pragma unroll
    for (int qp = 0; qp < 2; ++qp) {                       // channels 32 nt0 + 16 qp .. + 15
        if ((eln >> 4) == qp) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = 32 * (4 * wave + mt) + 8 * (r >> 2) + (r & 3) + 4 * eh2;
                    smem[row * 16 + (eln & 15)] = acc[mt][r];
                }
        }
        __syncthreads();
        const int cbase = 32 * (int)nt0 + 16 * qp;
#pragma unroll
        for (int it = 0; it < (SP_PR * SP_PC * 4 + 255) / 256; ++it) {
            const int item = etid + 256 * it;
            const int q = item & 3;
            int pp = item >> 2;
            pp = pp < SP_PR * SP_PC ? pp : SP_PR * SP_PC - 1;
            const int a = pp / SP_PC, b = pp - a * SP_PC;
            const int py = py0 + a, px = px0 + b;
            const int m0 = (2 * a) * SP_CC + 2 * b;
            // one window row at a time (three 16-byte reads in flight, running maximum): 12 registers of taps instead of 36 -- the
            // second half's 64 accumulators are still live here and the kernel must stay within 128 registers
            sp_f32x4 mx = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
            for (int tr = 0; tr < 3; ++tr) {
                sp_f32x4 v[3];
#pragma unroll
                for (int tc = 0; tc < 3; ++tc) v[tc] = *reinterpret_cast<const sp_f32x4*>(smem + (m0 + tr * SP_CC + tc) * 16 + 4 * q);
                if (!interior) {
                    const bool rok = (unsigned)(cy0 + 2 * a + tr) < (unsigned)p.ho;
#pragma unroll
                    for (int tc = 0; tc < 3; ++tc) {
                        const bool ok = rok && (unsigned)(cx0 + 2 * b + tc) < (unsigned)p.wo;
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[tc][k] = ok ? v[tc][k] : -INFINITY;
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) mx[k] = fmaxf(mx[k], fmaxf(fmaxf(v[0][k], v[1][k]), v[2][k]));
            }
            sp_f32x4 bb = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) bb = *reinterpret_cast<const sp_f32x4*>(p.bias + grp * 64 + cbase + 4 * q);
#pragma unroll
            for (int k = 0; k < 4; ++k) mx[k] = fmaxf(mx[k] + bb[k], 0.f);
            const bool ok = item < SP_PR * SP_PC * 4 && py < p.hp && px < p.wp;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(sp_u32x4, mx), rout,
                                                   ok ? (unsigned)(((py * p.wp + px) * 64 + cbase + 4 * q) * 4) : 0xFFFFFFFFu, 0, 0);
        }
        if (qp == 0) __syncthreads();
    }
}

// filters [groups][64][7][24] (layers.pack_stem3: w[co][dh][3 dw + c], entries 21..23 of a row zero, BN folded) ->
// packed [groups][2][21][64 lanes][4]: lane (n = lane & 31, h = lane >> 5), float s = w[32 nt + n][k]: filter row g / 3, within
// it k = 8 (g % 3) + 4 h + s for the two 8-k groups and 16 + 3 h + s (s < 3) for the 6-k group
__global__ void stem_pool_pack_kernel(const float* __restrict__ w, float* __restrict__ packed, int groups) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;           // (grp, nt, g, lane)
    const int total = groups * 2 * SP_NG * 64;
    if (idx >= total) return;
    const int lane = idx & 63;
    int r = idx >> 6;
    const int g = r % SP_NG;
    r /= SP_NG;
    const int nt = r & 1, grp = r >> 1;
    const int co = 32 * nt + (lane & 31);
    const int hh = lane >> 5;
    const float* row = w + ((long long)grp * 64 + co) * 168 + 24 * (g / 3);
    float4 v;
    if (g % 3 == 2) v = make_float4(row[16 + 3 * hh], row[17 + 3 * hh], row[18 + 3 * hh], 0.f);      // k = 16 + 3 h + s, three steps
    else { const float* src = row + 8 * (g % 3) + 4 * hh; v = make_float4(src[0], src[1], src[2], src[3]); }
    reinterpret_cast<float4*>(packed)[idx] = v;
}

extern "C" long long ss_stem_pool_packed_floats(int groups) { return groups > 0 ? (long long)groups * 2 * SP_NG * 64 * 4 : 0; }

extern "C" int ss_stem_pool_pack(const float* wgt, float* packed, int groups, void* stream) {
    if (!wgt || !packed || groups <= 0) return SS_ERR_ARG;
    const int total = groups * 2 * SP_NG * 64;
    hipLaunchKernelGGL(stem_pool_pack_kernel, dim3(ss_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, wgt, packed, groups);
    return ss_launch_status();
}

extern "C" int ss_stem_pool(const float* in_padded, const float* packed, const float* bias, float* out, int n, int h, int w,
                            int groups, long long out_gs, void* stream) {
    if (!in_padded || !packed || !out || n <= 0 || h <= 0 || w <= 0 || groups <= 0) return SS_ERR_ARG;
    StemP p;
    p.in = in_padded; p.packed = packed; p.bias = bias; p.out = out;
    p.n = n; p.h = h; p.w = w; p.groups = groups;
    p.ho = (h - 1) / 2 + 1; p.wo = (w - 1) / 2 + 1;          // Conv2d(k 7, s 2, p 3)
    p.hp = (p.ho - 1) / 2 + 1; p.wp = (p.wo - 1) / 2 + 1;    // MaxPool2d(k 3, s 2, p 1)
    p.nty = (unsigned)ss_cdiv(p.hp, SP_PR);
    p.ntx = (unsigned)ss_cdiv(p.wp, SP_PC);
    p.divTx = ss_div32_make(p.ntx);
    p.divTy = ss_div32_make(p.nty);
    p.divG = ss_div32_make((unsigned)groups);
    p.divG2 = ss_div32_make(2u * (unsigned)groups);
    p.out_gs = out_gs;
    const long long in_bytes = (long long)n * h * (w + 8) * 12;
    const long long wgs = (long long)n * p.nty * p.ntx * groups;
    if (in_bytes >= (1ll << 32) || wgs >= (1ll << 31)) return SS_ERR_UNSUPPORTED;
    if (groups > 1 && out_gs < (long long)n * p.hp * p.wp * 64) return SS_ERR_ARG;
    p.in_bytes = (unsigned)in_bytes;
    p.pk_bytes = (unsigned)(ss_stem_pool_packed_floats(groups) * 4);
#ifdef SS_TUNING
    p.dbg = ss_tuning_dbg;
    p.stagger = g_wino_knob[0];
    const unsigned dyn = (unsigned)g_wino_lds_pad;          // ss_debug_set key 20: extra LDS -> one workgroup per CU (experiments)
#else
    const unsigned dyn = 0u;
#endif
#ifdef SS_TUNING
    const bool split = g_wino_knob[2] != 2;                  // ss_debug_set(18, 2): the two-halves-per-workgroup kernel
#else
    const bool split = SP_SPLIT_DEFAULT;
#endif
    if (split) {
        const long long tiles = (long long)n * p.nty * p.ntx;
        const long long grid = 8ll * ((tiles + 7) / 8) * 2 * groups;         // (tile slots past the last tile exit at once)
        if (grid >= (1ll << 31)) return SS_ERR_UNSUPPORTED;
        p.ntiles = (unsigned)tiles;
        hipLaunchKernelGGL(stem_pool_kernel_half, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, p);
    } else {
        hipLaunchKernelGGL(stem_pool_kernel, dim3((unsigned)wgs), dim3(256), dyn, (hipStream_t)stream, p);
    }
    return ss_launch_status();
}
