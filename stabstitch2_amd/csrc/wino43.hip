// 3x3 / stride-1 / pad-1 convolution as FUSED Winograd F(4x4,3x3) on the fp32 matrix cores (gfx950).
//
// F(4x4,3x3) computes a 4x4 output tile from a 6x6 input tile with 36 multiplications per (cin, cout) pair where
// F(2x2,3x3) (wino.hip) spends 64 and the direct form 144:  Y = A^T [ (G g G^T) .* (B^T d B) ] A  (Lavin & Gray 2015),
// 36 independent GEMMs over cin, 1.78x fewer MFMA flops than wino.hip.  The price is the input transform: B^T has
// entries 4, -5, 2 instead of +-1 and every VALU instruction takes ~4 clocks of the SIMD the matrix pipe shares
// (tools/micro/mfma_valu: at ANY occupancy), so the transform is organised to run ONCE per workgroup and value:
//
//   workgroup = 512 threads = 8 waves, ONE per CU (144 KB of LDS); tile = 32 output tiles of 4x4 pixels -- GEO 0: 2 x 15 tiles = 8 rows x
//   60 columns of ONE image (the trunk's maps are 60 / 120 wide; slots 15 and 31 of the MFMA's 32 rows idle), GEO 1: 4 x 8 tiles = 16 x 32
//   pixels for maps up to 31 columns wide (layer3's 23 x 30) -- x 64 output channels x all 36 positions.  Wave (a, b, blk) owns the 3 x 3
//   block of positions (3a .. 3a+2, 3b .. 3b+2) for the 32-channel block `blk`: 9 accumulator tiles of v_mfma_f32_32x32x2_f32 = 144 registers.
//   K loop over cin in chunks of 16 = 8 channel PAIRS; both transforms run on packed fp32 (v_pk_fma_f32 / v_pk_add_f32) over the two
//   channels of a pair (inline asm: hipcc unpacks packed fp32 between MFMAs):
//     stage 1 (cooperative, once per workgroup): thread (pixel column x, tile row ty, channel quad) loads its 6 raw rows (16-byte buffer
//       loads, halo / image border through the descriptor's bounds check) one chunk ahead, applies the ROW transform B^T d (12 packed
//       ops per pair) and writes V1[pair][ty][i][x][2] to LDS with 8-byte stores (two buffers, one barrier per chunk);
//     stage 2 (per wave and pair step = 2 MFMA steps x 3 positions): lane (tile, k half) reads its window row of V1 as three 16-byte
//       entries (A[tx], B[tx], A[tx+1]; laid out for the ds_read_b128 lane groups, see XG) and forms the 3 A operands of its column block
//       for both channels in 6 packed ops -- .x / .y of a result pair feed the two MFMA steps;
//     72 MFMAs per wave and chunk against 124 VALU: B operand = transformed filters PRE-PACKED in the instruction's register layout
//       (global -> VGPR, 1 KB coalesced loads, two groups of 12 MFMAs ahead).
//   The K loop exists in two compile-time copies (column block B = 0 / 1 of the wave); the prologue (first rows, first filters, chunk 0's
//   row transform) runs INSIDE each copy: hipcc lays the wave-uniform branch out as "copy 0, then maybe copy 1", and whatever a shared
//   prologue left in registers for copy 1 was spilled around copy 0 (47 registers, 114 KB of scratch stores per workgroup in round 4).
//   epilogue in two phases of 16 tiles: all eight waves stage their accumulators ([position][tile][64 cout], 144 KB), then a thread owns a
//   (tile, cout pair): 36 8-byte LDS reads, A^T m A on packed fp32, bias (folded BatchNorm), residual (requested inside the last chunk /
//   while the first phase stores), ReLU, 8-byte stores (a wave writes 256-byte runs of a pixel's channels).
// Results differ from wino.hip / the direct form by fp32 rounding of the larger transform constants; gated in tests/ against fp64, the
// reference goldens (G8 / G9 forced everywhere, G14 under trained-like weights in three dispatch modes) and the oracle at 720p.
#include "common.h"
#include <atomic>
#include <thread>
#include <type_traits>

typedef float x_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned x_u32x4 __attribute__((ext_vector_type(4)));
typedef float x_f32x16 __attribute__((ext_vector_type(16)));
typedef float x_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned x_u32x2 __attribute__((ext_vector_type(2)));

struct W43P {
    const float* in;
    const float* U;          // packed transformed filters, see wino43_pack_kernel
    const float* bias;
    const float* res;
    float* out;
    int N, H, W, C, Co;      // C = input channels (multiple of 16), Co = output channels (multiple of 64)
    int nchunk;              // C / 16
    int relu, out_cs;
    SsDiv32 divBx, divBy, divNcb;
    unsigned nbx, nby, ncb;  // tile blocks per image row / column; 64-channel output blocks
    long long in_gs, u_gs, out_gs;
    unsigned in_bytes, out_bytes, u_bytes;
#ifdef SS_TUNING
    unsigned long long* dbg;            // per-workgroup phase stamps (tools/diag_wino43.py)
    int stagger;                        // experiment (ss_debug_set key 17): first-round workgroups start slot * stagger clocks late
#endif
};

#ifdef SS_TUNING
#define X_STAMP(i) do { if (p.dbg) ts[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define X_STAMP(i) do { } while (0)
#endif

#ifndef X_SCHED
#define X_SCHED 1
#endif

namespace {
// Two block geometries of the 32 tile slots a workgroup's MFMA rows hold (template parameter GEO):
//   GEO 0: 2 tile rows x 16 slots (15 carry pixels) = 8 x 60 output pixels -- the 60 / 120-wide maps of layer1 / layer2;
//   GEO 1: 4 tile rows x 8 slots = 16 x 32 output pixels, for maps up to 31 columns wide (layer3's 23 x 30: one block column; the
//          512 stage-1 threads cover raw columns -1 .. 30, the window's last two columns lie past every such map and are zeros
//          written once) -- on a 30-wide map GEO 0 would leave half of every tile row idle.
template <int GEO> struct XG;
// V1 (row-transformed input of one chunk) in LDS: [channel PAIR 8][tile row][transform row 6][ROWP], the two channels of a pair
// interleaved so that both transforms run on v_pk_*_f32 over aligned register pairs.  A row holds its columns as 16-byte entries
// (x, x + 1) x (c0, c1): entries of x = 0, 1 (mod 4) -- "A", from dword 0 -- and of x = 2, 3 (mod 4) -- "B", from dword BOFF.
// Tile tx's window x = 4 tx .. 4 tx + 5 is A[tx], B[tx], A[tx + 1]: three 16-byte reads; a 16-lane group of a read (GEO 0: 16 tiles of
// a tile row; GEO 1: 8 tiles of two tile rows, 6 ROWP = 32 (mod 64) dwords apart) covers 64 consecutive banks -- conflict-free.
// GEO 0 stage-1 writes (8 bytes per lane: 8 columns x 4 channel quads per half wave) hit all 64 banks once: quads are 2 planes =
// 16 banks apart (PL = 8 mod 32), B sits 8 banks behind A for the columns a half wave covers (they start at x = 0 mod 8 in tile
// row 0, at x = 2 mod 8 in tile row 1: hence the two B offsets).
// ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+ 32 for the upper half) -- not 16 consecutive
// lanes (MI355X_MICROARCH.md, LDS): a group holds complementary tile columns of different tile rows.  GEO 0: 8 tiles of tile row 0 and
// the other 8 of tile row 1 -- tile rows an exact multiple of 64 dwords apart (tyoff) and ONE B offset make the group cover 64
// consecutive banks.  GEO 1: four quarter rows of tile rows 0 .. 3: rows 0 / 1 a multiple of 64 apart, rows 2 / 3 shifted by 32 banks.
// (Round 5's first packed layout had tile rows 840 = 8 (mod 64) apart and B offsets 72 / 76: 0.37 of the kernel's LDS cycles were
// conflicts, profiles/r05_pmc_lds_before_relayout.json.)
template <> struct XG<0> {
    static constexpr int TYN = 2, TXS = 16, TXU = 15;           // tile rows, tile slots per row, slots that carry pixels
    static constexpr int BH = 4 * TYN, BW = 4 * TXU;            // output pixels of a block
    static constexpr int RW = BW + 2;                           // raw columns staged by threads (62; threads 496..511 idle)
    static constexpr int ROWP = 140;                            // A: 17 entries (68 dwords), B from 72: 16 entries
    static constexpr int TYS = 896;                             // tile-row stride: 6 * 140 = 840 rounded up to a multiple of 64
    static constexpr int PL = TYN * TYS + 8;                    // channel-pair plane (1800 = 8 mod 32)
    __device__ static constexpr int boff(int) { return 72; }
    __device__ static constexpr int tyoff(int ty) { return ty * TYS; }
};
template <> struct XG<1> {
    static constexpr int TYN = 4, TXS = 8, TXU = 8;
    static constexpr int BH = 4 * TYN, BW = 4 * TXU;            // 16 x 32
    static constexpr int RW = 32;                               // raw columns -1 .. 30 of the block (4 tile rows x 32 x 4 quads = 512 items)
    static constexpr int ROWP = 80;                             // A: 9 entries (36 dwords), B from 40: 8 entries
    static constexpr int TYS = 512;                             // 6 * 80 = 480 rounded up to a multiple of 64
    static constexpr int PL = TYN * TYS + 32 + 8;               // 2088 = 8 (mod 32)
    __device__ static constexpr int boff(int) { return 40; }
    __device__ static constexpr int tyoff(int ty) { return ty * TYS + ((ty >> 1) & 1) * 32; }
};
constexpr int X_V1F_MAX = 8 * XG<0>::PL > 8 * XG<1>::PL ? 8 * XG<0>::PL : 8 * XG<1>::PL;
constexpr int X_DUMPF = 36 * 32 * 32;        // epilogue stage: [position][tile][cout of one 32-channel block]
constexpr int X_SMEMF = X_DUMPF > 2 * X_V1F_MAX ? X_DUMPF : 2 * X_V1F_MAX;
constexpr unsigned X_UPOS = 2048u;           // bytes of one packed position (2 halves x 64 lanes x 16 B)
constexpr unsigned X_UCHUNK = 36u * X_UPOS;  // bytes per (32-cout block, chunk)
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t x_rsrc(const float* base, unsigned bytes) {
    unsigned long long a = (unsigned long long)base;
    unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    void* ub = (void*)(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(ub, 0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

// Packed fp32 on register PAIRS (the two channels of a pair), forced by inline asm: hipcc's gfx950 model unpacks v_pk_* between MFMAs
// ("packed fp32 does not co-issue with the matrix pipe"), but beside a dense v_mfma_f32_32x32x2_f32 stream a v_pk_fma_f32 costs the
// same issue slot as a v_fma_f32 (tools/micro/mfma_valu kind 3 vs 2: 117 vs 121 TF/s at 4 per MFMA, 95 vs 83 at 8) -- half the
// transform instructions for the same arithmetic, bit for bit.  k = a wave-uniform constant pair in SGPRs (one constant-bus operand).
__device__ __forceinline__ x_f32x2 x_pk_k(x_f32x2 k, x_f32x2 x, x_f32x2 c) {        // k * x + c
    x_f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "s"(k), "v"(x), "v"(c));
    return d;
}
__device__ __forceinline__ x_f32x2 x_pk_nk(x_f32x2 k, x_f32x2 x, x_f32x2 c) {       // (-k) * x + c
    x_f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(d) : "s"(k), "v"(x), "v"(c));
    return d;
}
__device__ __forceinline__ x_f32x2 x_pk_add(x_f32x2 a, x_f32x2 b) {
    x_f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ x_f32x2 x_pk_sub(x_f32x2 a, x_f32x2 b) {                  // a - b
    x_f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
#define X_LO(v) __builtin_shufflevector(v, v, 0, 1)
#define X_HI(v) __builtin_shufflevector(v, v, 2, 3)

// conv_wino43_kernel: ONE workgroup per tile block, with the phase stamps and compile-time ablations the lab tools use
// (tools/diag_wino43.py, tools/ab_wino43_stagger.py).  TUNING BUILD ONLY since round 6: the product library carries
// conv_wino43p_kernel below (same arithmetic, bit-identical; launched with one workgroup per block it IS this kernel's schedule).
#ifdef SS_TUNING
// ABL (tuning build, timing only -- wrong results): compile-time ablations of the K loop: 1 no filter loads, 2 no stage 1 (raw
// loads, row transform, LDS writes), 4 no stage-2 arithmetic, 8 no LDS reads, 16 no barrier
// ONE: cin == 16, the single chunk is the last one (its own instantiation: with the K loop's zero-trip case in the same code hipcc
// kept the prologue's prefetches alive on a second path around the loop and spilled 47 registers to scratch for it)
template <bool RES, int ABL = 0, bool ONE = false, int GEO = 0>
__global__ __launch_bounds__(512, 1) void conv_wino43_kernel(W43P p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using Gm = XG<GEO>;
    constexpr int X_ROWP = Gm::ROWP, X_PL = Gm::PL, X_V1F = 8 * Gm::PL, X_RW = Gm::RW, X_TXU = Gm::TXU;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int blk = wave & 1;                              // 32-channel half of the workgroup's 64 output channels
    const int pa = wave >> 2, pbb = (wave >> 1) & 1;        // position block: rows 3 pa .., columns 3 pbb ..
#ifdef SS_TUNING
    unsigned long long ts[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    X_STAMP(0);
#ifdef SS_TUNING
    // experiment: all 256 CUs start their first workgroup together and, with workgroups of equal length, stay in lock-step: every
    // round's prologue is one HBM burst of the whole chip.  Spread the first round over `stagger` clocks per CU slot of the XCD.
    if (p.stagger > 0 && blockIdx.x < 256u) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        const unsigned long long wait = (unsigned long long)((blockIdx.x >> 3) & 31u) * (unsigned)p.stagger;
        while (__builtin_amdgcn_s_memtime() - t0 < wait) __builtin_amdgcn_s_sleep(8);
    }
#endif

    // XCD-aware block order (as wino.hip): every XCD gets one contiguous run of tile blocks
    unsigned lin = blockIdx.x;
    {
        const unsigned nwg = gridDim.x;
        if (nwg >= 16) {
            const unsigned q = nwg / 8, r = nwg % 8, xcd = lin % 8, idx = lin / 8;
            lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        }
    }
    const unsigned mb = ss_div32(lin, p.divNcb);
    const unsigned cbk = lin - mb * p.ncb;                  // 64-channel output block
    const unsigned t1 = ss_div32(mb, p.divBx);
    const int bx = (int)(mb - t1 * p.nbx);
    const unsigned img = ss_div32(t1, p.divBy);
    const int by = (int)(t1 - img * p.nby);
    const int oy0 = by * Gm::BH, ox0 = bx * Gm::BW;
    const int grp = blockIdx.z;

    const __amdgpu_buffer_rsrc_t rin = x_rsrc(p.in + (long long)grp * p.in_gs, p.in_bytes);
    const __amdgpu_buffer_rsrc_t ru = x_rsrc(p.U + (long long)grp * p.u_gs, p.u_bytes);

    // ---- stage 1 item of this thread: (channel quad q, raw column xx, tile row ty); threads 496..511 carry no pixel: their
    // loads are out of range (zeros) and their writes land in the two pad columns of V1
    const int s1_q = tid & 3;
    const int s1_pix = tid >> 2;
    const bool s1_real = GEO == 1 || s1_pix < 2 * X_RW;
    const int s1_ty = GEO == 1 ? (s1_pix >> 5) : (s1_real ? (s1_pix >= X_RW ? 1 : 0) : ((s1_pix >> 1) & 1));
    const int s1_xx = GEO == 1 ? (s1_pix & 31) : (s1_real ? s1_pix - s1_ty * X_RW : X_RW + (s1_pix & 1));
    const int s1_iy0 = oy0 - 1 + 4 * s1_ty, s1_ix = ox0 - 1 + s1_xx;
    const unsigned rowstep = (unsigned)p.W * (unsigned)p.C * 4u;
    const unsigned rbase = ((((unsigned)img * p.H + (unsigned)s1_iy0) * p.W + (unsigned)s1_ix) * (unsigned)p.C + 4u * s1_q) * 4u;
    // rows of the item that lie inside the image (bit r); v_bfe_i32 turns a bit into the all-ones "out of range" mask
    int rmask = 0;
#pragma unroll
    for (int r = 0; r < 6; ++r)
        rmask |= (s1_real && (unsigned)(s1_iy0 + r) < (unsigned)p.H && (unsigned)s1_ix < (unsigned)p.W) ? 0 : (1 << r);
    const int s1_lds = (2 * s1_q) * X_PL + Gm::tyoff(s1_ty) + ((s1_xx & 2) ? Gm::boff(s1_ty) : 0) + (s1_xx >> 2) * 4 + (s1_xx & 1) * 2;
    if constexpr (GEO == 1) {
        // window columns 32, 33 (image columns 31, 32: past every map this geometry takes) = entry A[8] of every V1 row of both
        // buffers: zeros, written once -- 2 buffers x 8 pairs x 4 tile rows x 6 rows = 384 entries (made visible by the prologue's barrier)
        if (tid < 384) {
            const int bufi = tid / 192, r = tid - bufi * 192;          // r = (pair, tile row, row)
            *reinterpret_cast<x_f32x4*>(smem + bufi * X_V1F + (r / 24) * X_PL + Gm::tyoff((r % 24) / 6) + ((r % 24) % 6) * X_ROWP + 4 * Gm::TXS) = (x_f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }

    // ---- this lane in the GEMMs (v_mfma_f32_32x32x2_f32: A[i = lane & 31][k = lane >> 5]): tile lane & 31, channels
    // 8 kh .. 8 kh + 7 of the chunk = pairs 4 kh .. 4 kh + 3
    const int kh = lane >> 5;
    const int m_tile = lane & 31;
    const int m_ty = GEO == 1 ? (m_tile >> 3) : (m_tile >> 4), m_tx = GEO == 1 ? (m_tile & 7) : (m_tile & 15);
    const int t_srcA = (4 * kh) * X_PL + Gm::tyoff(m_ty) + (3 * pa) * X_ROWP + 4 * m_tx;
    const int t_srcB = t_srcA + Gm::boff(m_ty);
    const x_f32x2 k2 = {2.f, 2.f}, k4 = {4.f, 4.f}, k5 = {5.f, 5.f};

    const unsigned u_lane = (unsigned)lane * 16u;
    // packed filters: [cout/32][chunk][pos 36][half][lane][4]
    const unsigned u_wave = (cbk * 2u + (unsigned)blk) * (unsigned)p.nchunk * X_UCHUNK + (unsigned)((3 * pa) * 6 + 3 * pbb) * X_UPOS;

    x_f32x16 acc[3][3];
    auto acc_zero = [&]() {
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[g][s][r] = 0.f;
    };

    x_f32x4 rr[6];
    if constexpr (ABL & 2) {
#pragma unroll
        for (int r = 0; r < 6; ++r) rr[r] = (x_f32x4){0.f, 0.f, 0.f, 0.f};
    }
    auto raw_issue = [&](int c) {
        if constexpr (ABL & 2) return;
        const unsigned coff = (unsigned)c * 64u;
        int rm = rmask;
        asm volatile("" : "+v"(rm));         // (left alone, hipcc hoists the six masks out of the K loop and spills them)
#pragma unroll
        for (int r = 0; r < 6; ++r)
            rr[r] = __builtin_bit_cast(x_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (rbase + (unsigned)r * rowstep + coff) | (unsigned)__builtin_amdgcn_sbfe(rm, r, 1), 0, 0));
    };
    // row transform B^T d of channel pair pr of the quad (rows 0..2 when half == 0, rows 3..5 when half == 1), packed over the pair
    auto s1_piece = [&](float* buf, int pr, int half) {
        if constexpr (ABL & 2) return;
        auto D = [&](int r) { return pr ? X_HI(rr[r]) : X_LO(rr[r]); };
        float* w = buf + s1_lds + pr * X_PL;
        auto put = [&](int i, x_f32x2 v) { *reinterpret_cast<x_f32x2*>(w + i * X_ROWP) = v; };
        if (half == 0) {
            const x_f32x2 t1 = x_pk_nk(k4, D(2), D(4));
            const x_f32x2 t2 = x_pk_nk(k4, D(1), D(3));
            put(0, x_pk_k(k4, D(0), x_pk_nk(k5, D(2), D(4))));
            put(1, x_pk_add(t1, t2));
            put(2, x_pk_sub(t1, t2));
        } else {
            const x_f32x2 t3 = x_pk_sub(D(4), D(2));
            const x_f32x2 t4 = x_pk_sub(D(3), D(1));
            put(3, x_pk_k(k2, t4, t3));
            put(4, x_pk_nk(k2, t4, t3));
            put(5, x_pk_k(k4, D(1), x_pk_nk(k5, D(3), D(5))));
        }
    };
    auto lds_barrier = [&]() {       // __syncthreads() minus its global-memory fence (it would drain every prefetch in flight)
        if constexpr (ABL & 16) return;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    };

    // filters of one GROUP = (chunk, half h, position row g): three positions x 16 bytes (MFMA steps 4 h .. 4 h + 3)
    x_f32x4 u[3][3];
    if constexpr (ABL & 1) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) u[i][j] = (x_f32x4){(float)lane, 1.f, 2.f, (float)(i + j)};
    }
    auto u_issue = [&](int set, int c, int G) {
        if constexpr (ABL & 1) return;
        const int h = G / 3, g = G % 3;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int so = (int)__builtin_amdgcn_readfirstlane(u_wave + (unsigned)c * X_UCHUNK + (unsigned)(g * 6 + s) * X_UPOS + (unsigned)h * 1024u);
            u[set][s] = __builtin_bit_cast(x_f32x4, __builtin_amdgcn_raw_buffer_load_b128(ru, u_lane, so, 0));
        }
    };

    // ---- prologue: chunk 0 row-transformed in LDS, chunk 1's rows and the first two filter groups in flight.  Runs INSIDE each of
    // the two column-block copies of the K loop: hipcc structurizes the (wave-uniform) branch between them as "copy 0, then maybe
    // copy 1", so whatever the prologue leaves in registers for copy 1 is live across all of copy 0 -- it spilled 47 registers (12
    // prefetched vectors) to scratch around copy 0 for that: 114 KB of scratch stores per workgroup, as much as the output tile
    auto prologue = [&]() {
        acc_zero();
        X_STAMP(7);
        raw_issue(0);
        u_issue(0, 0, 0);
        u_issue(1, 0, 1);
#pragma unroll
        for (int k = 0; k < 4; ++k) s1_piece(smem, k >> 1, k & 1);
        X_STAMP(8);
        lds_barrier();
        X_STAMP(1);
        if constexpr (!ONE) raw_issue(1);
    };

    // ---- epilogue addressing and the residual: set up inside the LAST chunk (live ranges do not cross the K loop), the residual
    // of the first phase requested there too -- behind the chunk's last filter loads, so no wait of the stream covers it
    float* __restrict__ out = p.out + (long long)grp * p.out_gs;
    const __amdgpu_buffer_rsrc_t rout = x_rsrc(out, p.out_bytes);
    const __amdgpu_buffer_rsrc_t rres = x_rsrc(RES ? p.res + (long long)grp * p.out_gs : out, p.out_bytes);
    int e_n, tx, kh_e, lane_e;
    unsigned pixb, rowb, base;
    unsigned roff[4], cinv[4];
    x_f32x2 rv[4][4];
    auto epi_setup = [&]() {
        // (thread indices re-derived behind an opaque copy: hipcc otherwise computes these addresses in front of the K loop and
        // spills them across it)
        int tid_e = threadIdx.x;
        asm volatile("" : "+v"(tid_e));
        e_n = tid_e & 31;
        tx = tid_e >> 5;
        lane_e = tid_e & 63;
        kh_e = lane_e >> 5;
        // pixel offsets: row part per a -- 0xFFFF0000 (past every buffer the launcher admits) for rows outside the image -- plus
        // y * pixel pitch, or-ed with the column's out-of-range mask (columns past the image, the idle tile slot)
        // tile `tx` of a phase's 16: GEO 0 -- column tx of tile row `phase`; GEO 1 -- column tx & 7 of tile row 2 phase + (tx >> 3)
        const int txc = GEO == 1 ? (tx & 7) : tx;
        const int oxb = ox0 + 4 * txc;
        pixb = (unsigned)p.out_cs * 4u;
        rowb = pixb * (unsigned)p.W;
        base = ((((unsigned)img * p.H + oy0) * p.W + oxb) * (unsigned)p.out_cs + cbk * 64 + 2 * e_n) * 4u;
#pragma unroll
        for (int y = 0; y < 4; ++y) cinv[y] = (txc < X_TXU && oxb + y < p.W) ? 0u : 0xFFFFFFFFu;
    };
    auto row_offsets = [&](int ty) {
        const int trow = GEO == 1 ? 2 * ty + (tx >> 3) : ty;        // tile row of this thread's tile in phase ty
#pragma unroll
        for (int a = 0; a < 4; ++a) roff[a] = oy0 + 4 * trow + a < p.H ? base + (unsigned)(4 * trow + a) * rowb : 0xFFFF0000u;
    };
    // residual of output column y of the current phase's rows (roff)
    auto res_col = [&](int y) {
        if constexpr (RES) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
                rv[y][a] = __builtin_bit_cast(x_f32x2, __builtin_amdgcn_raw_buffer_load_b64(rres, (roff[a] + (unsigned)y * pixb) | cinv[y], 0, 0));
        }
    };

    auto kloop = [&](auto bc) {
        constexpr int B = decltype(bc)::value;          // column block of this wave (compile time: the column transform differs)
        prologue();
        x_f32x4 rd[3];                                   // the window row of one pair step: A[tx], B[tx], A[tx + 1]
        x_f32x2 av[2][3];
        if constexpr (ABL & (4 | 8)) {
#pragma unroll
            for (int i = 0; i < 3; ++i) rd[i] = (x_f32x4){(float)lane, 1.f, 2.f, 3.f};
#pragma unroll
            for (int i = 0; i < 2; ++i) av[i][0] = av[i][1] = av[i][2] = (x_f32x2){(float)(lane + i), 1.f};
        }
        // pair step M of a chunk = (half h = M / 6, position row g = (M / 2) % 3, channel pair e2 = M % 2 of the half's four channels):
        // two MFMA steps (channels 2 e2, 2 e2 + 1) x three positions
        auto rd_issue = [&](const float* buf, int M) {
            if constexpr (ABL & 8) return;
            const int h = M / 6, g = (M / 2) % 3, e2 = M % 2;
            const int off = (2 * h + e2) * X_PL + g * X_ROWP;
            rd[0] = *reinterpret_cast<const x_f32x4*>(buf + t_srcA + off);
            rd[1] = *reinterpret_cast<const x_f32x4*>(buf + t_srcB + off);
            rd[2] = *reinterpret_cast<const x_f32x4*>(buf + t_srcA + off + 4);
            // (all 16 bytes are "used": left alone hipcc shortens the half-used entry of a column block to a ds_read_b64, whose
            // 32-lane groups put the tiles of both tile rows on the same banks)
            asm volatile("" : "+v"(rd[0]), "+v"(rd[2]));
        };
        // column transform of the window row in rd -> the three A operands of this wave's column block, both channels of the pair
        auto xf = [&](int slot) {
            if constexpr (ABL & 4) return;
            const x_f32x2 x0 = X_LO(rd[0]), x1 = X_HI(rd[0]), x2 = X_LO(rd[1]), x3 = X_HI(rd[1]), x4 = X_LO(rd[2]), x5 = X_HI(rd[2]);
            if constexpr (B == 0) {
                const x_f32x2 s1 = x_pk_nk(k4, x2, x4);
                const x_f32x2 s2 = x_pk_nk(k4, x1, x3);
                av[slot][0] = x_pk_k(k4, x0, x_pk_nk(k5, x2, x4));
                av[slot][1] = x_pk_add(s1, s2);
                av[slot][2] = x_pk_sub(s1, s2);
            } else {
                const x_f32x2 s3 = x_pk_sub(x4, x2), s4 = x_pk_sub(x3, x1);
                av[slot][0] = x_pk_k(k2, s4, s3);
                av[slot][1] = x_pk_nk(k2, s4, s3);
                av[slot][2] = x_pk_k(k4, x1, x_pk_nk(k5, x3, x5));
            }
        };
        rd_issue(smem, 0);
        xf(0);
        rd_issue(smem, 1);
        // one chunk = 12 pair steps.  MORE (compile time): a chunk follows -- its row transform, its first two steps' operands and its
        // filters are produced inside this one.  The last chunk is a second copy without them: branches inside the stream cost
        // more than the code (tried: +19 % K-loop time), and nothing stays in flight in front of the epilogue's barrier.
        auto chunk = [&](int c, auto more_c) {
            constexpr bool MORE = decltype(more_c)::value;
            float* bc_ = smem + (c & 1) * X_V1F;                // V1 of chunk c
            float* bn = smem + ((c + 1) & 1) * X_V1F;           // chunk c + 1 (written during this chunk)
            const int c2 = c + 2 < p.nchunk ? c + 2 : c + 1;    // (the last but one chunk re-requests its successor's rows: unused)
#pragma unroll
            for (int M = 0; M < 12; ++M) {
                const int G = M / 2, g = G % 3, e2 = M % 2;
                __builtin_amdgcn_sched_barrier(0);
                // operands of the next pair step from the row in rd (requested a step ago), then the request of the row after it
                if (M + 1 < 12 || MORE) xf((M + 1) & 1);
                if (M + 2 < 12) rd_issue(bc_, M + 2);
                else if (MORE) rd_issue(bn, M + 2 - 12);
                if (e2 == 0) {                                   // first step of a group: the filters of group G + 2
                    if (G + 2 < 6) u_issue((G + 2) % 3, c, G + 2);
                    else if (MORE) u_issue((G + 2) % 3, c + 1, G + 2 - 6);
                }
#if X_SCHED == 1
                __builtin_amdgcn_sched_barrier(0);               // the requests go out in front of the step's six MFMAs
#endif
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int s = 0; s < 3; ++s)
                        acc[g][s] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[M & 1][s][e], u[G % 3][s][2 * e2 + e], acc[g][s], 0, 0, 0);
                if constexpr (!MORE) {
                    if (M == 6) { epi_setup(); row_offsets(0); }
                    if (M == 7) res_col(0);
                    if (M == 8) res_col(1);
                    if (M == 9) res_col(2);
                    if (M == 10) res_col(3);
                }
                if constexpr (MORE) {
                    // stage 1 of chunk c + 1 between the MFMAs of steps 4..7 (its rows were requested a chunk ago)
                    if (M == 4) s1_piece(bn, 0, 0);
                    if (M == 5) s1_piece(bn, 0, 1);
                    if (M == 6) s1_piece(bn, 1, 0);
                    if (M == 7) s1_piece(bn, 1, 1);
                    // behind step 8's filter loads: buffer loads return in order, the next filter wait (4 steps on) covers these too
                    if (M == 8) raw_issue(c2);
                    // every V1 read of chunk c has been issued (step 11's, two steps ahead); behind the barrier chunk c + 1 is read
                    if (M == 9) { __builtin_amdgcn_sched_barrier(0); lds_barrier(); }
                }
            }
        };
        if constexpr (!ONE) {
            int c = 0;
            do { chunk(c, std::true_type{}); } while (++c + 1 < p.nchunk);
        }
#ifdef SS_TUNING
        // experiment (ss_debug_set(17, -1 / -2)): the two waves of a SIMD (w, w + 4) drift apart inside a chunk (the older one wins
        // the matrix pipe); behind the last chunk nothing re-aligns them but the epilogue's barrier.  Priority to the younger wave
        // (-1: in the last chunk, -2: in every chunk's second half) -- does the skew ("wait" in tools/diag_wino43.py) shrink?
        if (p.stagger == -1 && wave >= 4) __builtin_amdgcn_s_setprio(2);
#endif
        chunk(p.nchunk - 1, std::false_type{});
#ifdef SS_TUNING
        if (p.stagger == -1) __builtin_amdgcn_s_setprio(0);
#endif
    };
    if (pbb == 0) kloop(std::integral_constant<int, 0>{});
    else kloop(std::integral_constant<int, 1>{});
    X_STAMP(2);

    // ---------------------------------------------------------------- epilogue: Y = A^T M A, bias, residual, ReLU
    const float relu_lo = p.relu ? 0.f : -__builtin_inff();
    // Two phases, one per tile row ty: ALL eight waves stage their accumulators of that row's 16 tiles ([position][tile][64 couts],
    // 144 KB), then every thread owns the tile (ty, tx) for the output channels 2 e_n and 2 e_n + 1 -- the two halves of packed
    // registers (nothing multiplies here: v_pk_add_f32 / v_pk_fma_f32 do both channels' work per issue slot) and of 8-byte LDS
    // reads, residual loads and stores (a wave moves 256-byte runs of a pixel's channels).
    const float* s0 = smem + tx * 64 + 2 * e_n;
    float bias0 = 0.f, bias1 = 0.f;
    if (p.bias) {
        bias0 = p.bias[(long long)grp * p.Co + cbk * 64 + 2 * e_n];
        bias1 = p.bias[(long long)grp * p.Co + cbk * 64 + 2 * e_n + 1];
    }
    const x_f32x2 c2 = {2.f, 2.f}, c4 = {4.f, 4.f}, c8 = {8.f, 8.f}, bias2 = {bias0, bias1};
#pragma unroll
    for (int ty = 0; ty < 2; ++ty) {
        lds_barrier();                                      // V1 (phase 0) / the previous phase's stage is free; (not
                                                            // __syncthreads(): its vmcnt(0) would wait for the residual here)
        if (ty == 0) X_STAMP(5);
        {
            float* d = smem + ((3 * pa) * 6 + 3 * pbb) * 1024 + blk * 32 + (lane_e & 31);
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int s = 0; s < 3; ++s)
#pragma unroll
                    for (int r8 = 0; r8 < 8; ++r8) {
                        const int t16 = (r8 & 3) + 8 * (r8 >> 2) + 4 * kh_e;   // accumulator 8 ty + r8 = tile 16 ty + t16
                        d[(g * 6 + s) * 1024 + t16 * 64] = acc[g][s][8 * ty + r8];
                    }
        }
        if (ty == 0) X_STAMP(6);
        lds_barrier();
        X_STAMP(3 + ty);
        // this phase's store offsets; roff then moves on to the next phase: its residual columns are requested as soon as this
        // phase has consumed theirs
        unsigned roff_s[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) roff_s[a] = roff[a];
        if (ty == 0) row_offsets(1);
        // rows of M -> T[i][y] = sum_j M[i][j] A[j][y]
        x_f32x2 t[6][4];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            x_f32x2 m[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) m[j] = *reinterpret_cast<const x_f32x2*>(s0 + (i * 6 + j) * 1024);
            const x_f32x2 p12 = m[1] + m[2], q12 = m[1] - m[2], p34 = m[3] + m[4], q34 = m[3] - m[4];
            t[i][0] = (m[0] + p12) + p34;
            t[i][1] = __builtin_elementwise_fma(c2, q34, q12);
            t[i][2] = __builtin_elementwise_fma(c4, p34, p12);
            t[i][3] = __builtin_elementwise_fma(c8, q34, q12) + m[5];
        }
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            const x_f32x2 p12 = t[1][y] + t[2][y], q12 = t[1][y] - t[2][y], p34 = t[3][y] + t[4][y], q34 = t[3][y] - t[4][y];
            x_f32x2 o[4];
            o[0] = (t[0][y] + p12) + p34;
            o[1] = __builtin_elementwise_fma(c2, q34, q12);
            o[2] = __builtin_elementwise_fma(c4, p34, p12);
            o[3] = __builtin_elementwise_fma(c8, q34, q12) + t[5][y];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                x_f32x2 v = o[a] + bias2;
                if (RES) v = v + rv[y][a];
                const unsigned off = (roff_s[a] + (unsigned)y * pixb) | cinv[y];
                v[0] = fmaxf(v[0], relu_lo);
                v[1] = fmaxf(v[1], relu_lo);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(x_u32x2, v), rout, off, 0, 0);
            }
            if (ty == 0) res_col(y);
        }
    }
#ifdef SS_TUNING
    if (p.dbg && tid == 0) {
        unsigned long long* d = p.dbg + ((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 12;
        for (int i = 0; i < 5; ++i) d[i] = ts[i];
        d[5] = __builtin_amdgcn_s_memtime();
        d[6] = ts[5];
        d[7] = ts[6];
        d[8] = ts[7];
        d[9] = ts[8];
        d[10] = __builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);   // HW_ID, XCC_ID
        d[11] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

#endif      // SS_TUNING

// The same kernel with PERSISTENT workgroups (round 6): one workgroup per CU walks its share of the launch's tile
// blocks and requests block n + 1's chunk-0 rows and first two filter groups in front of block n's epilogue.  A workgroup is alone
// on its CU (144 KB of LDS), so nothing else can hide the prologue's loaded HBM round trip; the registers it needs (rr: 24, u: 24)
// are free once the last MFMA of a block has issued.  Arithmetic per output identical to conv_wino43_kernel: bit-identical results.
template <bool RES, bool ONE = false, int GEO = 0>
__global__ __launch_bounds__(512, 1) void conv_wino43p_kernel(W43P p) {
    constexpr int ABL = 0;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using Gm = XG<GEO>;
    constexpr int X_ROWP = Gm::ROWP, X_PL = Gm::PL, X_V1F = 8 * Gm::PL, X_RW = Gm::RW, X_TXU = Gm::TXU;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int blk = wave & 1;                              // 32-channel half of the workgroup's 64 output channels
    const int pa = wave >> 2, pbb = (wave >> 1) & 1;        // position block: rows 3 pa .., columns 3 pbb ..

    // ---- the workgroup is PERSISTENT: it walks tile blocks blockIdx.x, blockIdx.x + gridDim.x, ... of the launch (XCD-aware order,
    // as wino.hip: every XCD gets one contiguous run of blocks; gridDim.x is a multiple of 8, so a workgroup's blocks stay on its
    // XCD), and requests the NEXT block's raw rows and first filters before the current block's epilogue: the prologue's loaded HBM
    // round trip (5k of a layer1 workgroup's 66k clocks, r05_wino43_ablations.txt) hides behind the dump / combine phases.
    const unsigned nblocks = p.N * p.nbx * p.nby * p.ncb;
    unsigned cbk = 0, img = 0;
    int oy0 = 0, ox0 = 0;
    auto decode = [&](unsigned it, unsigned& cbk_, unsigned& img_, int& oy_, int& ox_) __attribute__((always_inline)) {
        unsigned lin = it;
        if (nblocks >= 16) {
            const unsigned q = nblocks / 8, r = nblocks % 8, xcd = lin % 8, idx = lin / 8;
            lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        }
        const unsigned mb = ss_div32(lin, p.divNcb);
        cbk_ = lin - mb * p.ncb;                            // 64-channel output block
        const unsigned t1 = ss_div32(mb, p.divBx);
        const int bx = (int)(mb - t1 * p.nbx);
        img_ = ss_div32(t1, p.divBy);
        const int by = (int)(t1 - img_ * p.nby);
        oy_ = by * Gm::BH;
        ox_ = bx * Gm::BW;
    };
    const int grp = blockIdx.z;

    const __amdgpu_buffer_rsrc_t rin = x_rsrc(p.in + (long long)grp * p.in_gs, p.in_bytes);
    const __amdgpu_buffer_rsrc_t ru = x_rsrc(p.U + (long long)grp * p.u_gs, p.u_bytes);

    // ---- stage 1 item of this thread: (channel quad q, raw column xx, tile row ty); threads 496..511 carry no pixel: their
    // loads are out of range (zeros) and their writes land in the two pad columns of V1
    const int s1_q = tid & 3;
    const int s1_pix = tid >> 2;
    const bool s1_real = GEO == 1 || s1_pix < 2 * X_RW;
    const int s1_ty = GEO == 1 ? (s1_pix >> 5) : (s1_real ? (s1_pix >= X_RW ? 1 : 0) : ((s1_pix >> 1) & 1));
    const int s1_xx = GEO == 1 ? (s1_pix & 31) : (s1_real ? s1_pix - s1_ty * X_RW : X_RW + (s1_pix & 1));
    const unsigned rowstep = (unsigned)p.W * (unsigned)p.C * 4u;
    unsigned rbase = 0u;
    // rows of the item that lie inside the image (bit r); v_bfe_i32 turns a bit into the all-ones "out of range" mask
    int rmask = 0;
    auto s1_block = [&](unsigned img_, int oy_, int ox_) __attribute__((always_inline)) {       // the item's addressing in block (img_, oy_, ox_)
        // (the item's indices re-derived from the thread id behind an opaque copy, as epi_setup does: kept from the kernel's entry
        // they were five more registers live across the K loop, where the allocator has none to spare)
        int t_ = threadIdx.x;
        asm volatile("" : "+v"(t_));
        const int q_ = t_ & 3, pix_ = t_ >> 2;
        const bool real_ = GEO == 1 || pix_ < 2 * X_RW;
        const int ty_ = GEO == 1 ? (pix_ >> 5) : (real_ ? (pix_ >= X_RW ? 1 : 0) : ((pix_ >> 1) & 1));
        const int xx_ = GEO == 1 ? (pix_ & 31) : (real_ ? pix_ - ty_ * X_RW : X_RW + (pix_ & 1));
        const int s1_iy0 = oy_ - 1 + 4 * ty_, s1_ix = ox_ - 1 + xx_;
        rbase = ((((unsigned)img_ * p.H + (unsigned)s1_iy0) * p.W + (unsigned)s1_ix) * (unsigned)p.C + 4u * q_) * 4u;
        rmask = 0;
#pragma unroll
        for (int r = 0; r < 6; ++r)
            rmask |= (real_ && (unsigned)(s1_iy0 + r) < (unsigned)p.H && (unsigned)s1_ix < (unsigned)p.W) ? 0 : (1 << r);
    };
    const int s1_lds = (2 * s1_q) * X_PL + Gm::tyoff(s1_ty) + ((s1_xx & 2) ? Gm::boff(s1_ty) : 0) + (s1_xx >> 2) * 4 + (s1_xx & 1) * 2;
    auto zero_cols = [&]() __attribute__((always_inline)) {
        if constexpr (GEO == 1) {
            // window columns 32, 33 (image columns 31, 32: past every map this geometry takes) = entry A[8] of every V1 row of both
            // buffers: zeros, written per block (the epilogue's stage overwrites them) -- 2 buffers x 8 pairs x 4 tile rows x 6 rows =
            // 384 entries (made visible by the prologue's barrier)
            if (tid < 384) {
                const int bufi = tid / 192, r = tid - bufi * 192;          // r = (pair, tile row, row)
                *reinterpret_cast<x_f32x4*>(smem + bufi * X_V1F + (r / 24) * X_PL + Gm::tyoff((r % 24) / 6) + ((r % 24) % 6) * X_ROWP + 4 * Gm::TXS) = (x_f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
    };

    // ---- this lane in the GEMMs (v_mfma_f32_32x32x2_f32: A[i = lane & 31][k = lane >> 5]): tile lane & 31, channels
    // 8 kh .. 8 kh + 7 of the chunk = pairs 4 kh .. 4 kh + 3
    const int kh = lane >> 5;
    const int m_tile = lane & 31;
    const int m_ty = GEO == 1 ? (m_tile >> 3) : (m_tile >> 4), m_tx = GEO == 1 ? (m_tile & 7) : (m_tile & 15);
    const int t_srcA = (4 * kh) * X_PL + Gm::tyoff(m_ty) + (3 * pa) * X_ROWP + 4 * m_tx;
    const int t_srcB = t_srcA + Gm::boff(m_ty);
    const x_f32x2 k2 = {2.f, 2.f}, k4 = {4.f, 4.f}, k5 = {5.f, 5.f};

    const unsigned u_lane = (unsigned)lane * 16u;
    // packed filters: [cout/32][chunk][pos 36][half][lane][4]
    unsigned u_wave = 0u;
    auto u_block = [&](unsigned cbk_) __attribute__((always_inline)) {
        u_wave = (cbk_ * 2u + (unsigned)blk) * (unsigned)p.nchunk * X_UCHUNK + (unsigned)((3 * pa) * 6 + 3 * pbb) * X_UPOS;
    };

    x_f32x16 acc[3][3];
    auto acc_zero = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[g][s][r] = 0.f;
    };

    x_f32x4 rr[6];
    if constexpr (ABL & 2) {
#pragma unroll
        for (int r = 0; r < 6; ++r) rr[r] = (x_f32x4){0.f, 0.f, 0.f, 0.f};
    }
    auto raw_issue = [&](int c) __attribute__((always_inline)) {
        if constexpr (ABL & 2) return;
        const unsigned coff = (unsigned)c * 64u;
        int rm = rmask;
        asm volatile("" : "+v"(rm));         // (left alone, hipcc hoists the six masks out of the K loop and spills them)
#pragma unroll
        for (int r = 0; r < 6; ++r)
            rr[r] = __builtin_bit_cast(x_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (rbase + (unsigned)r * rowstep + coff) | (unsigned)__builtin_amdgcn_sbfe(rm, r, 1), 0, 0));
    };
    // row transform B^T d of channel pair pr of the quad (rows 0..2 when half == 0, rows 3..5 when half == 1), packed over the pair
    auto s1_piece = [&](float* buf, int pr, int half) __attribute__((always_inline)) {
        if constexpr (ABL & 2) return;
        auto D = [&](int r) __attribute__((always_inline)) { return pr ? X_HI(rr[r]) : X_LO(rr[r]); };
        float* w = buf + s1_lds + pr * X_PL;
        auto put = [&](int i, x_f32x2 v) __attribute__((always_inline)) { *reinterpret_cast<x_f32x2*>(w + i * X_ROWP) = v; };
        if (half == 0) {
            const x_f32x2 t1 = x_pk_nk(k4, D(2), D(4));
            const x_f32x2 t2 = x_pk_nk(k4, D(1), D(3));
            put(0, x_pk_k(k4, D(0), x_pk_nk(k5, D(2), D(4))));
            put(1, x_pk_add(t1, t2));
            put(2, x_pk_sub(t1, t2));
        } else {
            const x_f32x2 t3 = x_pk_sub(D(4), D(2));
            const x_f32x2 t4 = x_pk_sub(D(3), D(1));
            put(3, x_pk_k(k2, t4, t3));
            put(4, x_pk_nk(k2, t4, t3));
            put(5, x_pk_k(k4, D(1), x_pk_nk(k5, D(3), D(5))));
        }
    };
    auto lds_barrier = [&]() __attribute__((always_inline)) {       // __syncthreads() minus its global-memory fence (it would drain every prefetch in flight)
        if constexpr (ABL & 16) return;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    };

    // filters of one GROUP = (chunk, half h, position row g): three positions x 16 bytes (MFMA steps 4 h .. 4 h + 3)
    x_f32x4 u[3][3];
    if constexpr (ABL & 1) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) u[i][j] = (x_f32x4){(float)lane, 1.f, 2.f, (float)(i + j)};
    }
    auto u_issue = [&](int set, int c, int G) __attribute__((always_inline)) {
        if constexpr (ABL & 1) return;
        const int h = G / 3, g = G % 3;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int so = (int)__builtin_amdgcn_readfirstlane(u_wave + (unsigned)c * X_UCHUNK + (unsigned)(g * 6 + s) * X_UPOS + (unsigned)h * 1024u);
            u[set][s] = __builtin_bit_cast(x_f32x4, __builtin_amdgcn_raw_buffer_load_b128(ru, u_lane, so, 0));
        }
    };

    // ---- prologue: chunk 0 row-transformed in LDS, chunk 1's rows and the first two filter groups in flight.  Runs INSIDE each of
    // the two column-block copies of the K loop: hipcc structurizes the (wave-uniform) branch between them as "copy 0, then maybe
    // copy 1", so whatever the prologue leaves in registers for copy 1 is live across all of copy 0 -- it spilled 47 registers (12
    // prefetched vectors) to scratch around copy 0 for that: 114 KB of scratch stores per workgroup, as much as the output tile
    // `pre`: this block's chunk-0 rows and first two filter groups were requested in front of the previous block's epilogue
    auto prologue = [&](bool pre) __attribute__((always_inline)) {
        acc_zero();
        if (!pre) raw_issue(0);
        u_issue(0, 0, 0);           // (filters come from L2 -- every workgroup reads the same ones: not worth 24 registers across the epilogue)
        u_issue(1, 0, 1);
        zero_cols();
#pragma unroll
        for (int k = 0; k < 4; ++k) s1_piece(smem, k >> 1, k & 1);
        lds_barrier();
        if constexpr (!ONE) raw_issue(1);
    };

    // ---- epilogue addressing and the residual: set up inside the LAST chunk (live ranges do not cross the K loop), the residual
    // of the first phase requested there too -- behind the chunk's last filter loads, so no wait of the stream covers it
    float* __restrict__ out = p.out + (long long)grp * p.out_gs;
    const __amdgpu_buffer_rsrc_t rout = x_rsrc(out, p.out_bytes);
    const __amdgpu_buffer_rsrc_t rres = x_rsrc(RES ? p.res + (long long)grp * p.out_gs : out, p.out_bytes);
    int e_n, tx, kh_e, lane_e;
    unsigned pixb, rowb, base;
    unsigned roff[4], cinv[4];
    x_f32x2 rv[4][4];
    auto epi_setup = [&]() __attribute__((always_inline)) {
        // (thread indices re-derived behind an opaque copy: hipcc otherwise computes these addresses in front of the K loop and
        // spills them across it)
        int tid_e = threadIdx.x;
        asm volatile("" : "+v"(tid_e));
        e_n = tid_e & 31;
        tx = tid_e >> 5;
        lane_e = tid_e & 63;
        kh_e = lane_e >> 5;
        // pixel offsets: row part per a -- 0xFFFF0000 (past every buffer the launcher admits) for rows outside the image -- plus
        // y * pixel pitch, or-ed with the column's out-of-range mask (columns past the image, the idle tile slot)
        // tile `tx` of a phase's 16: GEO 0 -- column tx of tile row `phase`; GEO 1 -- column tx & 7 of tile row 2 phase + (tx >> 3)
        const int txc = GEO == 1 ? (tx & 7) : tx;
        const int oxb = ox0 + 4 * txc;
        pixb = (unsigned)p.out_cs * 4u;
        rowb = pixb * (unsigned)p.W;
        base = ((((unsigned)img * p.H + oy0) * p.W + oxb) * (unsigned)p.out_cs + cbk * 64 + 2 * e_n) * 4u;
#pragma unroll
        for (int y = 0; y < 4; ++y) cinv[y] = (txc < X_TXU && oxb + y < p.W) ? 0u : 0xFFFFFFFFu;
    };
    auto row_offsets = [&](int ty) __attribute__((always_inline)) {
        const int trow = GEO == 1 ? 2 * ty + (tx >> 3) : ty;        // tile row of this thread's tile in phase ty
#pragma unroll
        for (int a = 0; a < 4; ++a) roff[a] = oy0 + 4 * trow + a < p.H ? base + (unsigned)(4 * trow + a) * rowb : 0xFFFF0000u;
    };
    // residual of output column y of the current phase's rows (roff)
    auto res_col = [&](int y) __attribute__((always_inline)) {
        if constexpr (RES) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
                rv[y][a] = __builtin_bit_cast(x_f32x2, __builtin_amdgcn_raw_buffer_load_b64(rres, (roff[a] + (unsigned)y * pixb) | cinv[y], 0, 0));
        }
    };

    auto kloop = [&](auto bc, bool pre) __attribute__((always_inline)) {
        constexpr int B = decltype(bc)::value;          // column block of this wave (compile time: the column transform differs)
        prologue(pre);
        x_f32x4 rd[3];                                   // the window row of one pair step: A[tx], B[tx], A[tx + 1]
        x_f32x2 av[2][3];
        if constexpr (ABL & (4 | 8)) {
#pragma unroll
            for (int i = 0; i < 3; ++i) rd[i] = (x_f32x4){(float)lane, 1.f, 2.f, 3.f};
#pragma unroll
            for (int i = 0; i < 2; ++i) av[i][0] = av[i][1] = av[i][2] = (x_f32x2){(float)(lane + i), 1.f};
        }
        // pair step M of a chunk = (half h = M / 6, position row g = (M / 2) % 3, channel pair e2 = M % 2 of the half's four channels):
        // two MFMA steps (channels 2 e2, 2 e2 + 1) x three positions
        auto rd_issue = [&](const float* buf, int M) __attribute__((always_inline)) {
            if constexpr (ABL & 8) return;
            const int h = M / 6, g = (M / 2) % 3, e2 = M % 2;
            const int off = (2 * h + e2) * X_PL + g * X_ROWP;
            rd[0] = *reinterpret_cast<const x_f32x4*>(buf + t_srcA + off);
            rd[1] = *reinterpret_cast<const x_f32x4*>(buf + t_srcB + off);
            rd[2] = *reinterpret_cast<const x_f32x4*>(buf + t_srcA + off + 4);
            // (all 16 bytes are "used": left alone hipcc shortens the half-used entry of a column block to a ds_read_b64, whose
            // 32-lane groups put the tiles of both tile rows on the same banks)
            asm volatile("" : "+v"(rd[0]), "+v"(rd[2]));
        };
        // column transform of the window row in rd -> the three A operands of this wave's column block, both channels of the pair
        auto xf = [&](int slot) __attribute__((always_inline)) {
            if constexpr (ABL & 4) return;
            const x_f32x2 x0 = X_LO(rd[0]), x1 = X_HI(rd[0]), x2 = X_LO(rd[1]), x3 = X_HI(rd[1]), x4 = X_LO(rd[2]), x5 = X_HI(rd[2]);
            if constexpr (B == 0) {
                const x_f32x2 s1 = x_pk_nk(k4, x2, x4);
                const x_f32x2 s2 = x_pk_nk(k4, x1, x3);
                av[slot][0] = x_pk_k(k4, x0, x_pk_nk(k5, x2, x4));
                av[slot][1] = x_pk_add(s1, s2);
                av[slot][2] = x_pk_sub(s1, s2);
            } else {
                const x_f32x2 s3 = x_pk_sub(x4, x2), s4 = x_pk_sub(x3, x1);
                av[slot][0] = x_pk_k(k2, s4, s3);
                av[slot][1] = x_pk_nk(k2, s4, s3);
                av[slot][2] = x_pk_k(k4, x1, x_pk_nk(k5, x3, x5));
            }
        };
        rd_issue(smem, 0);
        xf(0);
        rd_issue(smem, 1);
        // one chunk = 12 pair steps.  MORE (compile time): a chunk follows -- its row transform, its first two steps' operands and its
        // filters are produced inside this one.  The last chunk is a second copy without them: branches inside the stream cost
        // more than the code (tried: +19 % K-loop time), and nothing stays in flight in front of the epilogue's barrier.
        auto chunk = [&](int c, auto more_c) __attribute__((always_inline)) {
            constexpr bool MORE = decltype(more_c)::value;
            float* bc_ = smem + (c & 1) * X_V1F;                // V1 of chunk c
            float* bn = smem + ((c + 1) & 1) * X_V1F;           // chunk c + 1 (written during this chunk)
            const int c2 = c + 2 < p.nchunk ? c + 2 : c + 1;    // (the last but one chunk re-requests its successor's rows: unused)
#pragma unroll
            for (int M = 0; M < 12; ++M) {
                const int G = M / 2, g = G % 3, e2 = M % 2;
                __builtin_amdgcn_sched_barrier(0);
                // operands of the next pair step from the row in rd (requested a step ago), then the request of the row after it
                if (M + 1 < 12 || MORE) xf((M + 1) & 1);
                if (M + 2 < 12) rd_issue(bc_, M + 2);
                else if (MORE) rd_issue(bn, M + 2 - 12);
                if (e2 == 0) {                                   // first step of a group: the filters of group G + 2
                    if (G + 2 < 6) u_issue((G + 2) % 3, c, G + 2);
                    else if (MORE) u_issue((G + 2) % 3, c + 1, G + 2 - 6);
                }
#if X_SCHED == 1
                __builtin_amdgcn_sched_barrier(0);               // the requests go out in front of the step's six MFMAs
#endif
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int s = 0; s < 3; ++s)
                        acc[g][s] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[M & 1][s][e], u[G % 3][s][2 * e2 + e], acc[g][s], 0, 0, 0);
                if constexpr (!MORE) {
                    if (M == 6) { epi_setup(); row_offsets(0); }
                    if (M == 7) res_col(0);
                    if (M == 8) res_col(1);
                    if (M == 9) res_col(2);
                    if (M == 10) res_col(3);
                }
                if constexpr (MORE) {
                    // stage 1 of chunk c + 1 between the MFMAs of steps 4..7 (its rows were requested a chunk ago)
                    if (M == 4) s1_piece(bn, 0, 0);
                    if (M == 5) s1_piece(bn, 0, 1);
                    if (M == 6) s1_piece(bn, 1, 0);
                    if (M == 7) s1_piece(bn, 1, 1);
                    // behind step 8's filter loads: buffer loads return in order, the next filter wait (4 steps on) covers these too
                    if (M == 8) raw_issue(c2);
                    // every V1 read of chunk c has been issued (step 11's, two steps ahead); behind the barrier chunk c + 1 is read
                    if (M == 9) { __builtin_amdgcn_sched_barrier(0); lds_barrier(); }
                }
            }
        };
        if constexpr (!ONE) {
            int c = 0;
            do { chunk(c, std::true_type{}); } while (++c + 1 < p.nchunk);
        }
        chunk(p.nchunk - 1, std::false_type{});
    };
    // the whole block loop exists once per column-block copy (what crosses the branch between the copies would be live across
    // copy 0, see the prologue's note): nothing but addresses does
    auto epilogue = [&](bool has_next, unsigned it_next) __attribute__((always_inline)) {

        // ---------------------------------------------------------------- epilogue: Y = A^T M A, bias, residual, ReLU
        const float relu_lo = p.relu ? 0.f : -__builtin_inff();
        // Two phases, one per tile row ty: ALL eight waves stage their accumulators of that row's 16 tiles ([position][tile][64 couts],
        // 144 KB), then every thread owns the tile (ty, tx) for the output channels 2 e_n and 2 e_n + 1 -- the two halves of packed
        // registers (nothing multiplies here: v_pk_add_f32 / v_pk_fma_f32 do both channels' work per issue slot) and of 8-byte LDS
        // reads, residual loads and stores (a wave moves 256-byte runs of a pixel's channels).
        const float* s0 = smem + tx * 64 + 2 * e_n;
        float bias0 = 0.f, bias1 = 0.f;
        if (p.bias) {
            bias0 = p.bias[(long long)grp * p.Co + cbk * 64 + 2 * e_n];
            bias1 = p.bias[(long long)grp * p.Co + cbk * 64 + 2 * e_n + 1];
        }
        const x_f32x2 c2 = {2.f, 2.f}, c4 = {4.f, 4.f}, c8 = {8.f, 8.f}, bias2 = {bias0, bias1};
    #pragma unroll
        for (int ty = 0; ty < 2; ++ty) {
            lds_barrier();                                      // V1 (phase 0) / the previous phase's stage is free; (not
                                                                // __syncthreads(): its vmcnt(0) would wait for the residual here)
            {
                float* d = smem + ((3 * pa) * 6 + 3 * pbb) * 1024 + blk * 32 + (lane_e & 31);
    #pragma unroll
                for (int g = 0; g < 3; ++g)
    #pragma unroll
                    for (int s = 0; s < 3; ++s)
    #pragma unroll
                        for (int r8 = 0; r8 < 8; ++r8) {
                            const int t16 = (r8 & 3) + 8 * (r8 >> 2) + 4 * kh_e;   // accumulator 8 ty + r8 = tile 16 ty + t16
                            d[(g * 6 + s) * 1024 + t16 * 64] = acc[g][s][8 * ty + r8];
                        }
            }
            lds_barrier();
            // this phase's store offsets; roff then moves on to the next phase: its residual columns are requested as soon as this
            // phase has consumed theirs
            unsigned roff_s[4];
    #pragma unroll
            for (int a = 0; a < 4; ++a) roff_s[a] = roff[a];
            if (ty == 0) row_offsets(1);
            // rows of M -> T[i][y] = sum_j M[i][j] A[j][y]
            x_f32x2 t[6][4];
    #pragma unroll
            for (int i = 0; i < 6; ++i) {
                x_f32x2 m[6];
    #pragma unroll
                for (int j = 0; j < 6; ++j) m[j] = *reinterpret_cast<const x_f32x2*>(s0 + (i * 6 + j) * 1024);
                const x_f32x2 p12 = m[1] + m[2], q12 = m[1] - m[2], p34 = m[3] + m[4], q34 = m[3] - m[4];
                t[i][0] = (m[0] + p12) + p34;
                t[i][1] = __builtin_elementwise_fma(c2, q34, q12);
                t[i][2] = __builtin_elementwise_fma(c4, p34, p12);
                t[i][3] = __builtin_elementwise_fma(c8, q34, q12) + m[5];
            }
    #pragma unroll
            for (int y = 0; y < 4; ++y) {
                const x_f32x2 p12 = t[1][y] + t[2][y], q12 = t[1][y] - t[2][y], p34 = t[3][y] + t[4][y], q34 = t[3][y] - t[4][y];
                x_f32x2 o[4];
                o[0] = (t[0][y] + p12) + p34;
                o[1] = __builtin_elementwise_fma(c2, q34, q12);
                o[2] = __builtin_elementwise_fma(c4, p34, p12);
                o[3] = __builtin_elementwise_fma(c8, q34, q12) + t[5][y];
    #pragma unroll
                for (int a = 0; a < 4; ++a) {
                    x_f32x2 v = o[a] + bias2;
                    if (RES) v = v + rv[y][a];
                    const unsigned off = (roff_s[a] + (unsigned)y * pixb) | cinv[y];
                    v[0] = fmaxf(v[0], relu_lo);
                    v[1] = fmaxf(v[1], relu_lo);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(x_u32x2, v), rout, off, 0, 0);
                }
                if (ty == 0) res_col(y);
            }
            // the NEXT block's chunk-0 rows, requested behind phase 0 (and behind phase 1's residual columns: buffer loads return in
            // order, the wait for a residual must not cover an HBM round trip of rows): rr is free (stage 1 ended a chunk ago) and so
            // is the half of the accumulators phase 0 staged; the rows arrive while phase 1 is dumped, combined and stored
            // (issued unconditionally -- behind the last block the row mask marks every row out of range and the descriptor returns
            // zeros without touching memory: with the loads inside a branch hipcc's vmcnt bookkeeping turns conservative and phase 1's
            // wait for its residual columns waits for these rows too)
            if (ty == 0) {
                unsigned cbk_n, img_n;
                int oy_n, ox_n;
                decode(has_next ? it_next : 0u, cbk_n, img_n, oy_n, ox_n);
                s1_block(img_n, oy_n, ox_n);
                rmask = has_next ? rmask : 0x3F;
                raw_issue(0);
            }
        }

        lds_barrier();              // the stage is read: the next block's row transform may overwrite it
    };
    auto run = [&](auto bc) __attribute__((always_inline)) {
        bool pre = false;
        for (unsigned it = blockIdx.x; it < nblocks; it += gridDim.x) {
            decode(it, cbk, img, oy0, ox0);
            if (!pre) s1_block(img, oy0, ox0);
            u_block(cbk);
            kloop(bc, pre);
            const unsigned it_next = it + gridDim.x;
            pre = it_next < nblocks;
            epilogue(pre, it_next);
        }
    };
    if (pbb == 0) run(std::integral_constant<int, 0>{});
    else run(std::integral_constant<int, 1>{});
}


// Packed transformed filters in the MFMA B-operand register layout (one 16-byte load per lane = 4 MFMA steps):
//   U[cout/32][chunk][pos 36][half][lane][e] = (G g G^T)[pos] of (cout = 32 cb + (lane & 31), cin = 16 chunk + 8 (lane >> 5) + 4 half + e)
// fp64 accumulation, rounded once.   wgt: [cout][1][3][3][cin] (BN folded).
__global__ void wino43_pack_kernel(const float* __restrict__ wgt, float* __restrict__ U, int cout, int cin, int nchunk,
                                   long long w_gs, long long u_gs) {
    const long long per = (long long)(cout / 32) * nchunk * 36 * 2 * 64;   // float4 slots per group
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= per) return;
    const int grp = blockIdx.y;
    const int lane = (int)(idx & 63);
    const int half = (int)((idx >> 6) & 1);
    const long long pc = idx >> 7;
    const int pos = (int)(pc % 36);
    const long long cc = pc / 36;
    const int chunk = (int)(cc % nchunk);
    const int cb = (int)(cc / nchunk);
    const int co = cb * 32 + (lane & 31);
    const int i = pos / 6, j = pos % 6;
    const double G[6][3] = {{1.0 / 4, 0.0, 0.0},          {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                            {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0.0, 0.0, 1.0}};
    float v[4];
    for (int e = 0; e < 4; ++e) {
        const int ci = chunk * 16 + 8 * (lane >> 5) + 4 * half + e;
        double acc = 0.0;
        if (ci < cin) {
            const float* g = wgt + (long long)grp * w_gs + (long long)co * 9 * cin + ci;
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) acc += G[i][a] * (double)g[(a * 3 + b) * cin] * G[j][b];
        }
        v[e] = (float)acc;
    }
    reinterpret_cast<float4*>(U + (long long)grp * u_gs)[idx] = make_float4(v[0], v[1], v[2], v[3]);
}

extern "C" long long ss_wino43_packed_floats(int cout, int cin) {
    if (cout <= 0 || cin <= 0 || (cout & 31) || (cin & 15)) return 0;
    return (long long)(cout / 32) * (cin / 16) * 36 * 2 * 64 * 4;
}

extern "C" int ss_wino43_pack(const float* wgt, float* packed, int cout, int cin, int groups, void* stream) {
    if (!wgt || !packed || cout <= 0 || cin <= 0 || (cout & 31) || (cin & 15) || groups <= 0) return SS_ERR_ARG;
    const int nchunk = cin / 16;
    const long long per = (long long)(cout / 32) * nchunk * 36 * 2 * 64;
    hipLaunchKernelGGL(wino43_pack_kernel, dim3(ss_cdiv(per, 256), groups), dim3(256), 0, (hipStream_t)stream, wgt, packed,
                       cout, cin, nchunk, (long long)cout * 9 * cin, per * 4);
    return ss_launch_status();
}

// Block geometry of a launch: the 16 x 32 blocks (GEO 1) for maps up to 31 columns wide, else 8 x 60 (GEO 0)
static inline int x_geo(int wo) { return wo <= 31 ? 1 : 0; }

// The engine's dispatch rule for this kernel (host/ops.py applies it; bench.py counts executed flops with it): geometry the kernel
// takes at all, then -- one workgroup per CU -- at least `min_wgs` workgroups (default 512: two rounds of the chip), enough of the
// block's tile slots on real pixels (default: 85 % of the 8 x 60 blocks; 60 % of the 16 x 32 blocks of narrow maps, where the
// alternative -- F(2x2,3x3) -- spends 1.78x the MFMA flops: layer3's 23 x 30 maps fill 67 %), K long enough to carry the
// un-overlapped prologue / epilogue (cin >= 64; tools/bench_wino43.py).
// images = images per group, groups = launch groups.  min_wgs / min_cin / min_fill_pct <= 0: the defaults; all three at 1 =
// "wherever the kernel runs at all".
extern "C" int ss_conv_uses_wino43(int kt, int kh, int kw, int stride, int cin, int cout, int ho, int wo, int images, int groups,
                                   int min_wgs, int min_cin, int min_fill_pct) {
    if (kt != 1 || kh != 3 || kw != 3 || stride != 1 || cin <= 0 || cout <= 0 || (cin & 15) || (cout & 63)) return 0;
    if (ho <= 0 || wo <= 0 || images <= 0 || groups <= 0) return 0;
    if ((long long)images * ho * wo * (cin > cout ? cin : cout) * 4 >= 0xFFFF0000ll) return 0;   // the kernel's 32-bit buffer offsets
    const int geo = x_geo(wo);
    const int bh = geo ? XG<1>::BH : XG<0>::BH, bw = geo ? XG<1>::BW : XG<0>::BW;
    if (min_wgs <= 0) min_wgs = 512;
    if (min_cin <= 0) min_cin = 64;
    if (min_fill_pct <= 0) min_fill_pct = geo ? 60 : 85;
    const long long nby = ss_cdiv(ho, bh), nbx = ss_cdiv(wo, bw);
    const double eff = (double)ho * wo / (double)(nby * bh * nbx * bw);
    return eff * 100.0 >= (double)min_fill_pct && (long long)images * nby * nbx * (cout / 64) * groups >= min_wgs && cin >= min_cin;
}

// Process-wide A/B knob (output-neutral, like ss_cost_volume_set_tile): 1 = persistent workgroups (conv_wino43p_kernel), 0 = one
// workgroup per tile block (conv_wino43_kernel).
static std::atomic<int> g_w43_persistent{1};
extern "C" int ss_wino43_set_persistent(int on) { g_w43_persistent.store(on ? 1 : 0); return SS_OK; }

#ifdef SS_TUNING
int g_w43_ablate = 0;                    // ss_debug_set key 21
extern int g_wino_knob[4];               // [1] (key 17): first-round stagger of this kernel, clocks per CU slot
#endif

extern "C" int ss_conv3x3_wino43_nhwc(const float* in, const float* packed, const float* bias, const float* res, float* out,
                                      int n, int h, int w, int cin, int cout, int relu, int out_cs, int groups,
                                      long long in_gs, long long u_gs, long long out_gs, void* stream) {
    if (!in || !packed || !out || n <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout <= 0 || groups <= 0 || out_cs < cout)
        return SS_ERR_ARG;
    if ((cin & 15) || (cout & 63)) return SS_ERR_UNSUPPORTED;
    const long long in_elems = (long long)n * h * w * cin;
    const long long out_elems = (long long)n * h * w * out_cs;
    const long long u_floats = ss_wino43_packed_floats(cout, cin);
    if (in_elems * 4 >= (1ll << 32) || out_elems * 4 >= 0xFFFF0000ll || (long long)out_cs * 16 >= 65536 || u_floats * 4 >= (1ll << 31))
        return SS_ERR_UNSUPPORTED;
    W43P p;
    p.in = in; p.U = packed; p.bias = bias; p.res = res; p.out = out;
    p.N = n; p.H = h; p.W = w; p.C = cin; p.Co = cout;
    p.nchunk = cin / 16;
    p.relu = relu; p.out_cs = out_cs;
    const int geo = x_geo(w);
    p.nbx = (unsigned)ss_cdiv(w, geo ? XG<1>::BW : XG<0>::BW);
    p.nby = (unsigned)ss_cdiv(h, geo ? XG<1>::BH : XG<0>::BH);
    p.ncb = (unsigned)(cout / 64);
    p.divBx = ss_div32_make(p.nbx);
    p.divBy = ss_div32_make(p.nby);
    p.divNcb = ss_div32_make(p.ncb);
    p.in_gs = in_gs; p.u_gs = u_gs; p.out_gs = out_gs;
    p.in_bytes = (unsigned)(in_elems * 4);
    p.out_bytes = (unsigned)(out_elems * 4);
    p.u_bytes = (unsigned)(u_floats * 4);
#ifdef SS_TUNING
    p.dbg = ss_tuning_dbg;
    p.stagger = g_wino_knob[1];
#endif
    const long long wgs = (long long)n * p.nbx * p.nby * p.ncb;
    if (wgs >= (1ll << 31)) return SS_ERR_UNSUPPORTED;
    constexpr unsigned lds = (unsigned)X_SMEMF * 4u;
    // more than the 64 KB a kernel gets by default: a per-DEVICE function attribute (a process may drive several GPUs), set once per
    // device whichever host thread launches first: 0 = not yet, 1 = a thread is setting it, 2 = set, 3 = the device refused
    static std::atomic<int> attr_state[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return SS_ERR_LAUNCH;
    int st8 = attr_state[dev].load(std::memory_order_acquire);
    if (st8 != 2) {
        int expect = 0;
        if (st8 == 0 && attr_state[dev].compare_exchange_strong(expect, 1, std::memory_order_acq_rel)) {
            const bool ok =
#ifdef SS_TUNING
                hipFuncSetAttribute((const void*)conv_wino43_kernel<true, 0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess &&
                hipFuncSetAttribute((const void*)conv_wino43_kernel<false, 0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess &&
                hipFuncSetAttribute((const void*)conv_wino43_kernel<true, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess &&
                hipFuncSetAttribute((const void*)conv_wino43_kernel<false, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess &&
                hipFuncSetAttribute((const void*)conv_wino43_kernel<true, 0, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess &&
                hipFuncSetAttribute((const void*)conv_wino43_kernel<false, 0, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess &&
                hipFuncSetAttribute((const void*)conv_wino43_kernel<true, 0, true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess &&
                hipFuncSetAttribute((const void*)conv_wino43_kernel<false, 0, true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess &&
#endif
                hipFuncSetAttribute((const void*)conv_wino43p_kernel<true, false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess &&
                hipFuncSetAttribute((const void*)conv_wino43p_kernel<false, false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess &&
                hipFuncSetAttribute((const void*)conv_wino43p_kernel<true, true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess &&
                hipFuncSetAttribute((const void*)conv_wino43p_kernel<false, true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess &&
                hipFuncSetAttribute((const void*)conv_wino43p_kernel<true, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess &&
                hipFuncSetAttribute((const void*)conv_wino43p_kernel<false, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess &&
                hipFuncSetAttribute((const void*)conv_wino43p_kernel<true, true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess &&
                hipFuncSetAttribute((const void*)conv_wino43p_kernel<false, true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
            if (!ok) (void)hipGetLastError();
            attr_state[dev].store(ok ? 2 : 3, std::memory_order_release);
        }
        while ((st8 = attr_state[dev].load(std::memory_order_acquire)) == 1) std::this_thread::yield();
        if (st8 != 2) return SS_ERR_DEVICE;            // the device cannot give a workgroup 144 KB of LDS (not gfx950)
    }
    dim3 g((unsigned)wgs, 1, groups);
    hipStream_t st = (hipStream_t)stream;
#ifdef SS_TUNING
    const bool use_p = g_w43_persistent.load(std::memory_order_relaxed) != 0;
#else
    const bool use_p = true;       // (product: ss_wino43_set_persistent(0) launches the same kernel with one workgroup per block)
#endif
    if (use_p) {
        // persistent workgroups: one per CU (a multiple of 8, so that a workgroup's blocks stay on its XCD), never more than blocks
        static std::atomic<int> cus[64];
        int ncu = cus[dev].load(std::memory_order_relaxed);
        if (ncu == 0) {
            hipDeviceProp_t prop;
            ncu = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
            ncu -= ncu % 8;
            if (ncu < 8) ncu = 8;
            cus[dev].store(ncu, std::memory_order_relaxed);
        }
        const unsigned per_group = (unsigned)((wgs < ncu || !g_w43_persistent.load(std::memory_order_relaxed)) ? wgs : ncu);
        dim3 gp(per_group, 1, groups);
#define X_LAUNCHP(RES_, ONE_, GEO_) hipLaunchKernelGGL((conv_wino43p_kernel<RES_, ONE_, GEO_>), gp, dim3(512), lds, st, p)
        const bool onep = p.nchunk == 1;
        if (geo) {
            if (onep) { if (res) X_LAUNCHP(true, true, 1); else X_LAUNCHP(false, true, 1); }
            else { if (res) X_LAUNCHP(true, false, 1); else X_LAUNCHP(false, false, 1); }
        } else {
            if (onep) { if (res) X_LAUNCHP(true, true, 0); else X_LAUNCHP(false, true, 0); }
            else { if (res) X_LAUNCHP(true, false, 0); else X_LAUNCHP(false, false, 0); }
        }
#undef X_LAUNCHP
        return ss_launch_status();
    }
#ifdef SS_TUNING
    // (tuning build, ss_wino43_set_persistent(0): the instrumented one-workgroup-per-block kernel)
    if (g_w43_ablate && !res && !geo) {         // tools/diag_wino43.py <layers> <ablation masks>
        switch (g_w43_ablate) {
#define X_ABL_CASE(m) case m: (void)hipFuncSetAttribute((const void*)conv_wino43_kernel<false, m>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            hipLaunchKernelGGL((conv_wino43_kernel<false, m>), g, dim3(512), lds, st, p); return ss_launch_status();
            X_ABL_CASE(1) X_ABL_CASE(2) X_ABL_CASE(4) X_ABL_CASE(8) X_ABL_CASE(12) X_ABL_CASE(14) X_ABL_CASE(15) X_ABL_CASE(16) X_ABL_CASE(31)
#undef X_ABL_CASE
            default: break;
        }
    }
#define X_LAUNCH(RES_, ONE_, GEO_) hipLaunchKernelGGL((conv_wino43_kernel<RES_, 0, ONE_, GEO_>), g, dim3(512), lds, st, p)
    const bool one = p.nchunk == 1;
    if (geo) {
        if (one) { if (res) X_LAUNCH(true, true, 1); else X_LAUNCH(false, true, 1); }
        else { if (res) X_LAUNCH(true, false, 1); else X_LAUNCH(false, false, 1); }
    } else {
        if (one) { if (res) X_LAUNCH(true, true, 0); else X_LAUNCH(false, true, 0); }
        else { if (res) X_LAUNCH(true, false, 0); else X_LAUNCH(false, false, 0); }
    }
#undef X_LAUNCH
#endif
    return ss_launch_status();
}
