// Convolution / pooling / FC engine for gfx950 (CDNA4).
//
// conv: fp32 implicit GEMM on v_mfma_f32_32x32x2_f32 (exact fp32 fma chain, 157 TF peak).
//   GEMM view  D[m][co] = sum_k A[m][k] * Wt[co][k]
//     m  = (n, to, ho, wo)  output position          (rows, M = n*to*ho*wo)
//     k  = (dt, dh, dw, ci) filter tap x in-channel  (ci fastest, cin % 4 == 0)
//   Block = 256 threads = WGM x WGN waves (=4), each wave owns WM x WN tiles of 32x32 (16 accumulator
//   registers each); block tile BM x BN = (WGM*WM*32) x (WGN*WN*32).
//   On a saturated fp32 MFMA pipe every non-MFMA instruction is paid for (a VALU instruction between a wave's own
//   MFMAs costs ~2.3 pipe cycles, one issued by a wave outside its K loop waits ~60 cycles behind the other
//   workgroups' MFMAs; LDS / buffer / barrier instructions are free -- tools/micro/), so everything around the MFMAs
//   is built to be cheap:
//     * K is walked 32 at a time; one 16-byte buffer_load per thread per 32 tile rows, issued ahead of the tile's
//       MFMAs.  Halo taps, the M tail and the K tail are handled by the buffer descriptor's bounds check: an invalid
//       lane gets byte offset 0xFFFFFFFF and the hardware returns zeros -- no branches, no selects on the data;
//     * addresses come from tables, not arithmetic: per workgroup an LDS table {tap byte offset, tap bit index} per
//       4 k, per tile row (decomposed once, by one thread, with multiply-shift divisions, shared through LDS) a base
//       offset and a <= 64-bit mask of the taps that fall outside the input.  Per K tile: one ds_read_b64, then per
//       row v_bfe_i32 + add + or -> 12 VALU instructions per 16 MFMAs (filters with > 64 taps fall back to the
//       arithmetic path: three multiply-high divisions per K tile);
//     * LDS holds rows k-contiguous ([rows][32+4]); the K index inside a tile is permuted so that lane half
//       h of the MFMA takes k = 16h + j: every lane reads its 16 operands as four ds_read_b128 (row stride
//       36 dwords = 4*9 keeps each 16-lane group on 16 distinct 16-byte slots: conflict free), and the
//       staging store is one ds_write_b128;
//     * ONE LDS buffer (18 KB) and two barriers per K tile: 6-7 workgroups resident per CU (measured: throughput is
//       flat from 3 to 7 resident workgroups, tools/ab_occupancy.py);
//     * s_setprio 3 outside the K loop so that prologue / epilogue instructions win the issue slot.
//   Epilogue: bias (folded BatchNorm), residual, ReLU; for a fixed accumulator register the 32 lanes of a
//   half-wave hold 32 consecutive output channels -> 128-byte coalesced NHWC accesses, branch free through buffer
//   instructions (out-of-range offsets for rows >= M), all residual loads in flight before the first use.
//   Small problems (few tiles, long K) are split along K into `splits` partial sums written to a caller
//   workspace and combined (deterministically) by splitk_reduce_kernel.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct ConvP {
    const float* in;
    const float* wgt;
    const float* bias;
    const float* res;
    float* out;
    float* partial;          // split-K workspace [groups*splits][M][Co] or nullptr
    int T, H, W, C;
    int px_b, row_b;                 // bytes between x-neighbours / rows of the input (C * 4, W * C * 4 for plain NHWC)
    int To, Ho, Wo, Co;
    int kt, kh, kw, s;
    int pt, ph, pw;
    int K, M;
    SsFastDiv divC, divKw, divKh;
    SsDiv32 divWo, divHo, divTo;     // output-row decomposition m -> (n, to, ho, wo)
    SsDiv32 divNt, divSplits;        // workgroup index decomposition (wave-uniform: scalar multiplies)
    int relu, out_cs;
    int pool2;                       // split-K launches only: the reduce kernel also takes MaxPool2d(2, 2) (out is the pooled map)
    int splits, tiles_per_split;
    unsigned ntiles;                 // Cout tiles (grid.x = M tiles * ntiles)
    long long in_gs, w_gs, out_gs;
    unsigned in_bytes, w_bytes;      // per-group extents for the buffer descriptors
#ifdef SS_TUNING                     // tools/ build only (csrc/build.sh tuning): never in the shipped library
    unsigned long long* dbg;         // ss_debug_ptr: per-workgroup phase timestamps, or nullptr
    int ablate;                      // ss_debug_set key 1: 8 = row-major tile order instead of XCD-aware, 32 = no setprio, 64 = no epilogue
#endif
};

#ifdef SS_TUNING
#define SS_ABLATE(p, bit) ((p).ablate & (bit))
#define SS_STAMP(p, var) do { if ((p).dbg) var = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SS_ABLATE(p, bit) 0
#define SS_STAMP(p, var) do { } while (0)
#endif


__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const float* base, unsigned bytes) {
    unsigned long long a = (unsigned long long)base;
    unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    void* ub = (void*)(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(ub, 0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

template <int WGM, int WGN, int WM, int WN, int NBUF, int MINW = 1, int BK = 32, bool TAIL = false, int SCHED = 0,
          int AMODE = 0>
__global__ __launch_bounds__(256, MINW) void conv_igemm_kernel(ConvP p) {
    constexpr int BM = WGM * WM * 32;
    constexpr int BN = WGN * WN * 32;
    constexpr int LDK = BK + 4;          // row stride in dwords: 4 * odd -> conflict-free ds_read_b128
    constexpr int TPR = BK / 4;          // threads (float4s) per tile row
    constexpr int RPP = 256 / TPR;       // rows staged per pass
    constexpr int RA = BM / RPP;         // A rows staged per thread
    constexpr int RB = BN / RPP;
    constexpr int NCH = BK / 8;          // 4-wide operand chunks per lane half
    __shared__ __attribute__((aligned(16))) float As[NBUF][BM * LDK];
    __shared__ __attribute__((aligned(16))) float Bs[NBUF][BN * LDK];
    extern __shared__ __attribute__((aligned(16))) uint2 tap_tab[];   // AMODE > 0: one entry per 4 k of this block's K range

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave / WGN, wc = wave % WGN;
    // XCD-aware tile order: the dispatcher deals consecutive workgroups round-robin to the 8 XCDs (private L2s).
    // Give every XCD one contiguous run of M tiles (all Cout tiles of an M tile back to back) so that the 3x3 halo
    // rows and the Cout-tile re-reads of an input tile hit that XCD's L2 instead of being fetched 8 times.
    int mt, nt;
    {
        const unsigned nwg = gridDim.x, b = blockIdx.x;
        unsigned lin = b;
        if (!SS_ABLATE(p, 8) && nwg >= 16) {
            const unsigned q = nwg / 8, r = nwg % 8, xcd = b % 8, idx = b / 8;
            lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        }
        mt = (int)ss_div32(lin, p.divNt);
        nt = (int)(lin - (unsigned)mt * p.ntiles);
    }
    // Waves outside the K loop (row setup, epilogue) issue only VALU / memory instructions; beside 6 waves that keep the
    // MFMA pipe busy they would get one issue slot per 64-cycle MFMA and take longer than the K loop itself (measured:
    // 31k + 44k cycles around a 75k-cycle loop).  Raised priority lets them through.
    if (!SS_ABLATE(p, 32)) __builtin_amdgcn_s_setprio(3);
#ifdef SS_TUNING
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, tsa = 0, tsb = 0, tsc = 0;
#endif
    SS_STAMP(p, ts0);
    const int m0 = mt * BM;
    const int n0 = nt * BN;
    const int grp = (int)ss_div32(blockIdx.z, p.divSplits);
    const int split = blockIdx.z - grp * p.splits;

    // descriptor inputs through readfirstlane so that the compiler can prove the SRD wave-uniform (otherwise
    // every buffer_load is wrapped in a waterfall loop)
    const __amdgpu_buffer_rsrc_t rin = uniform_rsrc(p.in + (long long)grp * p.in_gs, p.in_bytes);
    const __amdgpu_buffer_rsrc_t rw = uniform_rsrc(p.wgt + (long long)grp * p.w_gs, p.w_bytes);

    const int lrow = tid / TPR;
    const int kq = tid % TPR;    // which float4 of the BK-wide K tile

    // Row bookkeeping (fixed over the K loop): byte offset of tap (0,0,0) and tap-validity masks of each tile row.
    // Instructions issued outside the K loop are expensive (they queue behind the other workgroups' MFMAs, ~60 cycles
    // each), so every row is decomposed ONCE, by thread `row` with multiply-shift divisions, and shared through LDS
    // instead of being recomputed by the 8 threads that stage it; the remaining waves build the tap table meanwhile.
    __shared__ uint4 rowinfo[BM];      // {a_off, a_msk, a_inv_lo, a_inv_hi}
    if (tid < BM) {
        const int m = m0 + tid;
        const bool ok = m < p.M;
        const uint32_t mm = ok ? (uint32_t)m : 0u;
        const uint32_t t1 = ss_div32(mm, p.divWo);
        const int wo = (int)(mm - t1 * (uint32_t)p.Wo);
        const uint32_t t2 = ss_div32(t1, p.divHo);
        const int ho = (int)(t1 - t2 * (uint32_t)p.Ho);
        const uint32_t n = ss_div32(t2, p.divTo);
        const int to = (int)(t2 - n * (uint32_t)p.To);
        const int ti0 = to - p.pt, hi0 = ho * p.s - p.ph, wi0 = wo * p.s - p.pw;
        const int aoff = (((int)n * p.T + ti0) * p.H + hi0) * p.row_b + wi0 * p.px_b;     // bytes
        // valid taps along each axis form one interval [lo, hi): bit mask = (1 << hi) - (1 << lo)
        auto span = [](int x0, int k, int size) -> unsigned {
            int lo = max(0, -x0), hi = min(k, size - x0);
            return hi > lo ? (1u << hi) - (1u << lo) : 0u;
        };
        unsigned msk = span(wi0, p.kw, p.W) | (span(hi0, p.kh, p.H) << 8) | (span(ti0, p.kt, p.T) << 16);
        msk = ok ? msk : 0u;           // bits 0..7: valid dw, 8..15: valid dh, 16..23: valid dt
        unsigned inv_lo = 0xFFFFFFFFu, inv_hi = 0xFFFFFFFFu;
        if (AMODE == 1 || AMODE == 3) {
            // table mode, <= 32 taps: bit (dt*kh + dh)*kw + dw SET where the tap falls outside the input
            unsigned valid = 0u;
            const unsigned mw = msk & 0xFFu;
            if (p.kw == 1 && p.kt == 1) {
                // one tap column (the row-packed stem, 1x1 convs): the tap bits are the valid-dh bits themselves
                valid = (mw & (msk >> 16) & 1u) ? ((msk >> 8) & 0xFFu) : 0u;
            } else {
                for (int dt = 0; dt < p.kt; ++dt)
                    for (int dh = 0; dh < p.kh; ++dh) {
                        const unsigned sel = 0u - ((msk >> (8 + dh)) & (msk >> (16 + dt)) & 1u);
                        valid |= (mw << ((dt * p.kh + dh) * p.kw)) & sel;
                    }
            }
            inv_lo = ~valid;
        } else if (AMODE == 2) {
            unsigned long long valid = 0ull;
            const unsigned long long mw = msk & 0xFFu;
            for (int dt = 0; dt < p.kt; ++dt)
                for (int dh = 0; dh < p.kh; ++dh) {
                    const unsigned long long sel = 0ull - (unsigned long long)((msk >> (8 + dh)) & (msk >> (16 + dt)) & 1u);
                    valid |= (mw << ((dt * p.kh + dh) * p.kw)) & sel;
                }
            inv_lo = ~(unsigned)valid;
            inv_hi = ~(unsigned)(valid >> 32);
        }
        rowinfo[tid] = make_uint4((unsigned)aoff, msk, inv_lo, inv_hi);
    }
    SS_STAMP(p, tsa);
    unsigned w_off[RB], w_bad[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        int co = n0 + lrow + i * RPP;
        w_off[i] = (unsigned)co * (unsigned)p.K * 4u;
        w_bad[i] = co < p.Co ? 0u : 0xFFFFFFFFu;
    }

    const int nk_all = (p.K + BK - 1) / BK;   // == host-side nk for this BK
    const int kt0 = split * p.tiles_per_split;
    const int kt1 = min(nk_all, kt0 + p.tiles_per_split);

    if (AMODE > 0) {
        // per-block tap table: entry e describes k = kt0*BK + 4e: {byte offset of the tap relative to tap (0,0,0),
        // tap bit index, or 64 for k >= K}.  Replaces three divisions per K tile per thread by one ds_read_b64.
        constexpr int T0 = BM < 256 ? BM : 0;         // built by the threads that own no row
        const int tab_n = (kt1 - kt0) * TPR;
        if (tid >= T0) {
            for (int e = tid - T0; e < tab_n; e += 256 - T0) {
                int k = kt0 * BK + e * 4;
                uint32_t tap = ss_fastdiv((uint32_t)k, p.divC);
                int ci = k - (int)tap * p.C;
                uint32_t t2 = ss_fastdiv(tap, p.divKw);
                int dw = (int)tap - (int)t2 * p.kw;
                uint32_t dt = ss_fastdiv(t2, p.divKh);
                int dh = (int)t2 - (int)dt * p.kh;
                int tapoff = ((int)dt * p.H + dh) * p.row_b + dw * p.px_b + ci * 4;
                tap_tab[e] = k < p.K ? make_uint2((unsigned)tapoff, tap) : make_uint2(0u, 64u);
            }
        }
    }
    __syncthreads();
    int a_off[RA];
    unsigned a_inv_lo[RA], a_inv_hi[RA];
    unsigned a_msk[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const uint4 ri = rowinfo[lrow + i * RPP];
        a_off[i] = (int)ri.x;
        a_msk[i] = ri.y;
        a_inv_lo[i] = ri.z;
        a_inv_hi[i] = ri.w;
    }
    SS_STAMP(p, tsb);
    unsigned wk[RB];             // table mode: running byte offset of each filter row's next K tile
#pragma unroll
    for (int i = 0; i < RB; ++i) wk[i] = w_off[i] + (unsigned)(kt0 * BK + kq * 4) * 4u;

    u32x4 ra[RA], rb[RB];
    unsigned oa[RA], ob[RB];     // byte offsets of the next tile's loads (0xFFFFFFFF = out of range -> zeros)
    unsigned bsoff = 0u;         // AMODE 3: wave-uniform byte offset of the K tile inside a filter row
    auto gaddr = [&](int kt) {
        if (AMODE == 3) {
            // aligned problems (Cout % BN == 0, K % BK == 0, <= 32 taps): no K / Cout masks, and the filter rows advance
            // through the scalar offset of the buffer instruction -> 7 VALU instructions per K tile
            const uint2 e = tap_tab[(kt - kt0) * TPR + kq];
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                const unsigned inv = (unsigned)__builtin_amdgcn_sbfe((int)a_inv_lo[i], e.y, 1u);
                oa[i] = ((unsigned)a_off[i] + e.x) | inv;
            }
            bsoff = (unsigned)kt * (BK * 4u);
            return;
        }
        if (AMODE > 0) {
            const uint2 e = tap_tab[(kt - kt0) * TPR + kq];
            const unsigned kinv = (unsigned)__builtin_amdgcn_sbfe((int)e.y, 6u, 1u);      // all ones for k >= K
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                unsigned word = a_inv_lo[i];
                if (AMODE == 2) word = (e.y & 32u) ? a_inv_hi[i] : word;
                const unsigned inv = (unsigned)__builtin_amdgcn_sbfe((int)word, e.y, 1u);   // v_bfe_i32 uses idx[4:0]
                oa[i] = ((unsigned)a_off[i] + e.x) | inv | kinv;
            }
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                ob[i] = wk[i] | w_bad[i] | kinv;
                wk[i] += BK * 4u;
            }
            return;
        }
        int k = kt * BK + kq * 4;
        bool kok = k < p.K;
        uint32_t tap = ss_fastdiv((uint32_t)k, p.divC);
        int ci = k - (int)tap * p.C;
        uint32_t t2 = ss_fastdiv(tap, p.divKw);
        int dw = (int)tap - (int)t2 * p.kw;
        uint32_t dt = ss_fastdiv(t2, p.divKh);
        int dh = (int)t2 - (int)dt * p.kh;
        int tapoff = ((int)dt * p.H + dh) * p.row_b + dw * p.px_b + ci * 4;     // bytes
        unsigned sw = (unsigned)dw, sh = 8u + (unsigned)dh, st = 16u + dt;
        // branch-free: an invalid lane ORs its offset up to 0xFFFFFFFF and the descriptor returns zeros
        const unsigned kmask = kok ? 0u : 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            unsigned m = a_msk[i];
            unsigned v = (m >> sw) & (m >> sh) & (m >> st) & 1u;
            oa[i] = ((unsigned)(a_off[i] + tapoff)) | (v - 1u) | kmask;
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) ob[i] = (w_off[i] + (unsigned)k * 4u) | w_bad[i] | kmask;
    };
    auto gissue = [&]() {
#pragma unroll
        for (int i = 0; i < RA; ++i) ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rin, oa[i], 0, 0);
#pragma unroll
        for (int i = 0; i < RB; ++i)
            rb[i] = AMODE == 3 ? __builtin_amdgcn_raw_buffer_load_b128(rw, w_off[i] + (unsigned)kq * 16u, (int)bsoff, 0)
                               : __builtin_amdgcn_raw_buffer_load_b128(rw, ob[i], 0, 0);
    };
    auto gload = [&](int kt) {
        gaddr(kt);
        gissue();
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < RA; ++i)
            *reinterpret_cast<u32x4*>(&As[buf][(lrow + i * RPP) * LDK + kq * 4]) = ra[i];
#pragma unroll
        for (int i = 0; i < RB; ++i)
            *reinterpret_cast<u32x4*>(&Bs[buf][(lrow + i * RPP) * LDK + kq * 4]) = rb[i];
    };

    f32x16 acc[WM][WN];
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int b = 0; b < WN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int li = lane & 31, lh = lane >> 5;
    const int aidx = (wr * WM * 32 + li) * LDK + lh * (BK / 2);
    const int bidx = (wc * WN * 32 + li) * LDK + lh * (BK / 2);

    auto mma_chunk = [&](int buf, int c) {
        f32x4 a[WM], b[WN];
#pragma unroll
        for (int x = 0; x < WM; ++x) a[x] = *reinterpret_cast<const f32x4*>(&As[buf][aidx + x * 32 * LDK + c * 4]);
#pragma unroll
        for (int y = 0; y < WN; ++y) b[y] = *reinterpret_cast<const f32x4*>(&Bs[buf][bidx + y * 32 * LDK + c * 4]);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int x = 0; x < WM; ++x)
#pragma unroll
                for (int y = 0; y < WN; ++y)
                    acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[x][s], b[y][s], acc[x][y], 0, 0, 0);
    };

    if (kt0 < kt1) {
        gload(kt0);
        SS_STAMP(p, tsc);
        lstore(0);
    }
    __syncthreads();
    SS_STAMP(p, ts1);
    if (!SS_ABLATE(p, 32)) __builtin_amdgcn_s_setprio(0);
    // main loop: straight-line body (no branches) so that hipcc can interleave the next tile's address arithmetic and
    // loads with the MFMAs; the last tile is peeled (nothing to prefetch, and only it can be a partial K tile)
    int kt = kt0;
    for (; kt + 1 < kt1; ++kt) {
        const int buf = NBUF == 2 ? ((kt - kt0) & 1) : 0;
        gload(kt + 1);
        if (SCHED == 1) __builtin_amdgcn_sched_barrier(0);     // keep the loads ahead of the MFMAs
#pragma unroll
        for (int c = 0; c < NCH; ++c) mma_chunk(buf, c);
        if (NBUF == 1) __syncthreads();      // single LDS buffer: everyone done reading before it is overwritten
        lstore(NBUF == 2 ? (buf ^ 1) : 0);
        __syncthreads();
    }
    if (kt < kt1) {
        const int buf = NBUF == 2 ? ((kt - kt0) & 1) : 0;
        // K tail: with the k = (BK/2)h + j permutation chunk c holds k in {4c..4c+3} u {BK/2+4c..}; chunks past the end
        // of K only multiply zeros (conv1: K = 196 -> the 7th tile needs 1 chunk of 4).  TAIL is a template flag set
        // by the host only when K % BK < BK/2.
        if (!TAIL || p.K - kt * BK >= BK / 2) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) mma_chunk(buf, c);
        } else {
            const int cmax = (p.K - kt * BK + 3) >> 2;
            for (int c = 0; c < cmax; ++c) mma_chunk(buf, c);
        }
    }

    SS_STAMP(p, ts2);
    if (!SS_ABLATE(p, 32)) __builtin_amdgcn_s_setprio(3);
    // epilogue: bias (folded BatchNorm), residual, ReLU.  Branch free and without waits between the 16 rows a lane
    // owns: offsets of rows past M / channels past Cout are forced out of range (buffer loads return 0, stores are
    // dropped), all residual loads are issued before the first use and all stores after the last.  (The first
    // version tested `m < M` per row and the compiler serialised load -> wait -> store 16 times: 44k cycles per
    // workgroup beside a 75k-cycle K loop.)
    if (SS_ABLATE(p, 64)) return;                 // tuning build: no epilogue at all (wrong results; what would a free epilogue buy?)
    const bool direct = p.splits == 1;
    float* __restrict__ out = direct ? p.out + (long long)grp * p.out_gs
                                     : p.partial + (long long)blockIdx.z * p.M * p.Co;
    const float* __restrict__ res = (direct && p.res) ? p.res + (long long)grp * p.out_gs : nullptr;
    const int ocs = direct ? p.out_cs : p.Co;
    const unsigned obytes = (unsigned)p.M * (unsigned)ocs * 4u;      // < 2^32 checked by the host
    const __amdgpu_buffer_rsrc_t rout = uniform_rsrc(out, obytes);
    const __amdgpu_buffer_rsrc_t rres = uniform_rsrc(res ? res : out, obytes);
#pragma unroll
    for (int y = 0; y < WN; ++y) {
        const int co = n0 + (wc * WN + y) * 32 + li;
        const bool cok = co < p.Co;
        const float bias = (direct && p.bias && cok) ? p.bias[(long long)grp * p.Co + co] : 0.f;
#pragma unroll
        for (int x = 0; x < WM; ++x) {
            unsigned off[16];
            float rv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wr * WM + x) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                off[r] = (m < p.M && cok) ? ((unsigned)m * (unsigned)ocs + (unsigned)co) * 4u : 0xFFFFFFFFu;
            }
            if (res) {
#pragma unroll
                for (int r = 0; r < 16; ++r) rv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rres, off[r], 0, 0));
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[x][y][r] + bias;
                if (res) v += rv[r];
                if (direct && p.relu) v = fmaxf(v, 0.f);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rout, off[r], 0, 0);
            }
        }
    }
#ifdef SS_TUNING
    if (p.dbg && tid == 0) {
        unsigned long long* d = p.dbg + ((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 8;
        d[5] = tsa; d[6] = tsb; d[7] = tsc;
        d[0] = ts0; d[1] = ts1; d[2] = ts2; d[3] = __builtin_amdgcn_s_memtime();
        d[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
    }
#endif
}

// out[m][co] = relu(sum_s partial[s][m][co] + bias[co] + res[m][co]); fixed summation order.
// pool2: out[n][hp][wp][co] = relu(max over the 2 x 2 window of the split sums + bias) = MaxPool2d(2, 2) of the above (bias add and
// ReLU are monotone: same bits as pooling the stored map) -- the regressors' conv, ReLU, MaxPool2d(2, 2) on the small maps
// that run split-K: one launch less per pair of layers.
__global__ void splitk_reduce_kernel(ConvP p) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long per = (long long)p.M * p.Co;
    int grp = blockIdx.y;
    if (p.pool2) {
        const int hp = p.Ho >> 1, wp = p.Wo >> 1;
        const long long nimg = (long long)p.M / ((long long)p.Ho * p.Wo);
        const long long pper = nimg * hp * wp * p.Co;
        if (idx >= pper) return;
        const int co = (int)(idx % p.Co);
        long long q = idx / p.Co;
        const int x = (int)(q % wp); q /= wp;
        const int y = (int)(q % hp);
        const long long n = q / hp;
        const float* part = p.partial + (long long)grp * p.splits * per;
        // the window's four split sums, each in the fixed order s = 0, 1, ...; the loads of eight splits x four pixels are
        // issued together (a loop that loaded and added one value at a time ran at one L2 round trip per value: the fused
        // kernel took longer than the reduce + max-pool launches it replaced)
        const long long m00 = (n * p.Ho + 2 * y) * p.Wo + 2 * x;
        const float* pq = part + m00 * p.Co + co;
        const long long rowo = (long long)p.Wo * p.Co;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        for (int s0 = 0; s0 < p.splits; s0 += 8) {
            float t[8][4];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const bool ok = s0 + k < p.splits;
                const float* qs = pq + (long long)(ok ? s0 + k : s0) * per;
                t[k][0] = ok ? qs[0] : 0.f;
                t[k][1] = ok ? qs[p.Co] : 0.f;
                t[k][2] = ok ? qs[rowo] : 0.f;
                t[k][3] = ok ? qs[rowo + p.Co] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (s0 + k < p.splits) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] += t[k][i];
                }
        }
        float best = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
        if (p.bias) best += p.bias[(long long)grp * p.Co + co];
        if (p.relu) best = fmaxf(best, 0.f);
        p.out[(long long)grp * p.out_gs + ((n * hp + y) * wp + x) * p.out_cs + co] = best;
        return;
    }
    if (idx >= per) return;
    int co = (int)(idx % p.Co);
    long long m = idx / p.Co;
    const float* part = p.partial + (long long)grp * p.splits * per + idx;
    float v = 0.f;
    for (int s0 = 0; s0 < p.splits; s0 += 8) {          // eight loads in flight, added in the fixed order s = 0, 1, ...
        float t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = s0 + k < p.splits ? part[(long long)(s0 + k) * per] : 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (s0 + k < p.splits) v += t[k];
    }
    if (p.bias) v += p.bias[(long long)grp * p.Co + co];
    long long o = (long long)grp * p.out_gs + m * p.out_cs + co;
    if (p.res) v += p.res[o];
    if (p.relu) v = fmaxf(v, 0.f);
    p.out[o] = v;
}

// Tuning knobs: compile-time constants in the shipped library; the tools/ build (-DSS_TUNING) turns them into variables
// behind ss_debug_set / ss_debug_ptr (not part of the ABI, not declared in include/stabstitch_hip.h).
#ifdef SS_TUNING
static int g_lds_pad = 0;           // key 4: extra dynamic LDS bytes per workgroup (occupancy experiments)
static int g_amode_off = 0;         // key 3: 1 = arithmetic addressing, 2 = table without the aligned fast path
static int g_force_tile = 0;        // key 0: force a tile variant
static int g_ablate = 0;            // key 1: ablation mask
static int g_split_target = 512;    // key 2: workgroups a split-K launch aims for
unsigned long long* ss_tuning_dbg = nullptr;
extern int g_wino_nb1_max_cin, g_wino_ablate, g_wino_variant;
extern int g_wino_knob[4];
extern int g_wino_lds_pad;
extern int g_w43_ablate;
#define g_dbg ss_tuning_dbg
extern "C" SS_API void ss_debug_ptr(void* ptr) { g_dbg = (unsigned long long*)ptr; }
extern "C" SS_API void ss_debug_set(int key, int value) {
    if (key == 0) g_force_tile = value;
    if (key == 1) g_ablate = value;
    if (key == 2) g_split_target = value;
    if (key == 3) g_amode_off = value;
    if (key == 4) g_lds_pad = value;
    if (key == 6) g_wino_ablate = value;           // Winograd ablations (wrong results): 1 no filter loads, 2 no transform, 4 no epilogue
    if (key == 7) g_wino_variant = value;          // Winograd kernel: 0 auto, 1 one tile block per workgroup, 2 pair kernel
    if (key == 5) g_wino_nb1_max_cin = value;      // Winograd: 32-channel blocks (3 workgroups / CU) up to this cin
    if (key == 20) g_wino_lds_pad = value;
    if (key == 21) g_w43_ablate = value;           // F(4x4,3x3) K-loop ablations (timing only)
    if (key >= 16 && key < 20) g_wino_knob[key - 16] = value;      // Winograd round-3 experiments (see WinoP::knob)
}
#else
constexpr int g_lds_pad = 0, g_amode_off = 0, g_force_tile = 0;
constexpr int g_split_target = 512;     // measured best (tools/ab_splitk.py)
#endif

template <int WGM, int WGN, int WM, int WN, int NBUF, int MINW = 1, int BK = 32, bool TAIL = false, int SCHED = 0,
          int AMODE = 0>
static void launch_conv(const ConvP& p, int groups, hipStream_t st, unsigned dyn_lds = 0) {
    constexpr int BM = WGM * WM * 32, BN = WGN * WN * 32;
    ConvP q = p;
    q.ntiles = (unsigned)ss_cdiv(p.Co, BN);
    q.divNt = ss_div32_make(q.ntiles);
    q.divSplits = ss_div32_make((uint32_t)p.splits);
    dim3 g((unsigned)ss_cdiv(p.M, BM) * q.ntiles, 1, groups * p.splits);
    hipLaunchKernelGGL((conv_igemm_kernel<WGM, WGN, WM, WN, NBUF, MINW, BK, TAIL, SCHED, AMODE>), g, dim3(256),
                       dyn_lds + (unsigned)g_lds_pad, st, q);
}

// Address mode: per-block tap table in LDS (8 bytes per 4 k of the block's K range) when the filter has <= 64 taps
// and the table stays small; otherwise the arithmetic path (three divisions per K tile per thread).  `tail` selects
// the variant whose peeled last K tile may be shorter than BK/2 (only instantiated for the 64x64 and 128x64 tiles).
template <int WGM, int WGN, int WM, int WN, int NBUF, int BK = 32, int SCHED = 0>
static void launch_auto(const ConvP& p, int groups, hipStream_t st, int taps, bool tail) {
    const unsigned tab_bytes = (unsigned)p.tiles_per_split * (unsigned)(BK / 4) * 8u;
    constexpr int BN_ = WGN * WN * 32;
    int amode = ((g_amode_off & 1) || taps > 64 || tab_bytes > 24576u) ? 0 : (taps <= 32 ? 1 : 2);
    if (amode == 1 && !(g_amode_off & 2) && p.Co % BN_ == 0 && p.K % BK == 0) amode = 3;
    constexpr bool DEF = WGM == 2 && WGN == 2 && (WM == 1 || WM == 2) && WN == 1 && NBUF == 1 && BK == 32 && SCHED == 1;
    if (DEF && tail) {
        if (amode == 1) launch_conv<WGM, WGN, WM, WN, NBUF, 1, BK, DEF, SCHED, 1>(p, groups, st, tab_bytes);
        else if (amode == 2) launch_conv<WGM, WGN, WM, WN, NBUF, 1, BK, DEF, SCHED, 2>(p, groups, st, tab_bytes);
        else launch_conv<WGM, WGN, WM, WN, NBUF, 1, BK, DEF, SCHED, 0>(p, groups, st);
    } else {
        if (amode == 3) launch_conv<WGM, WGN, WM, WN, NBUF, 1, BK, false, SCHED, 3>(p, groups, st, tab_bytes);
        else if (amode == 1) launch_conv<WGM, WGN, WM, WN, NBUF, 1, BK, false, SCHED, 1>(p, groups, st, tab_bytes);
        else if (amode == 2) launch_conv<WGM, WGN, WM, WN, NBUF, 1, BK, false, SCHED, 2>(p, groups, st, tab_bytes);
        else launch_conv<WGM, WGN, WM, WN, NBUF, 1, BK, false, SCHED, 0>(p, groups, st);
    }
}

// Split-K plan of a launch (shared by ss_conv_workspace_need and ss_conv_nhwc): small problems (few tiles, long K) are
// cut along K so that ~g_split_target workgroups exist.  -> number of splits (1 = none).
static int conv_splits(long long M, int cout, int groups, int nk) {
    long long b64 = (long long)ss_cdiv(M, 64) * ss_cdiv(cout, 64) * groups;
    int want = (int)((g_split_target + b64 - 1) / b64);
    // (round 6) a launch with a SHORT K that already has 0.6 of the target gains little from a two-way cut and pays a reduction pass
    // over its whole output for it: twin-trunk layer2 head (K = 576) on 3 / 2 views, 508 / 340 tiles: 42.6 -> 26.8 us, 33.1 ->
    // 24.4 us without the cut (tools/ab_splitk_stream.py).  Long K keeps it (SmoothNet's 3-D convolutions of a clip, K = 3456, 442
    // tiles: the clip's implicit-GEMM launches lose 12 % without); rounding the quotient down everywhere loses the same.
    if (want == 2 && nk < 32 && b64 * 5 >= (long long)g_split_target * 3) want = 1;
    int maxs = nk / 4 > 0 ? nk / 4 : 1;
    int splits = want < maxs ? want : maxs;
    if (splits <= 1) return 1;
    const int tps = ss_cdiv(nk, splits);
    return ss_cdiv(nk, tps);
}

extern "C" long long ss_conv_workspace_need(int n, int t, int h, int w, int cin, int cout, int kt, int kh, int kw,
                                            int stride, int pad_t, int pad_h, int pad_w, int groups) {
    if (n <= 0 || t <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout <= 0 || kt <= 0 || kh <= 0 || kw <= 0 || stride <= 0 ||
        groups <= 0)
        return 0;
    const int To = t + 2 * pad_t - kt + 1, Ho = (h + 2 * pad_h - kh) / stride + 1, Wo = (w + 2 * pad_w - kw) / stride + 1;
    if (To <= 0 || Ho <= 0 || Wo <= 0) return 0;
    const long long M = (long long)n * To * Ho * Wo, K = (long long)kt * kh * kw * cin;
    const int splits = conv_splits(M, cout, groups, ss_cdiv(K, 32));
    return splits > 1 ? (long long)groups * splits * M * cout : 0;
}

static int conv_dispatch(ConvP& p, long long K, long long M, int groups, int taps, float* ws, long long ws_floats,
                         hipStream_t st);

extern "C" int ss_conv_nhwc(const float* in, const float* wgt, const float* bias, const float* res, float* out,
                            int n, int t, int h, int w, int cin, int cout, int kt, int kh, int kw, int stride,
                            int pad_t, int pad_h, int pad_w, int relu, int out_cs, int groups, long long in_gs,
                            long long w_gs, long long out_gs, float* ws, long long ws_floats, void* stream) {
    if (!in || !wgt || !out || n <= 0 || t <= 0 || h <= 0 || w <= 0 || cin <= 0 || (cin & 3) || cout <= 0 ||
        kt <= 0 || kh <= 0 || kw <= 0 || kt > 8 || kh > 8 || kw > 8 || stride <= 0 || groups <= 0 || out_cs < cout)
        return SS_ERR_ARG;
    ConvP p;
    p.in = in; p.wgt = wgt; p.bias = bias; p.res = res; p.out = out; p.partial = nullptr;
    p.T = t; p.H = h; p.W = w; p.C = cin;
    p.To = t + 2 * pad_t - kt + 1;
    p.Ho = (h + 2 * pad_h - kh) / stride + 1;
    p.Wo = (w + 2 * pad_w - kw) / stride + 1;
    p.Co = cout;
    if (p.To <= 0 || p.Ho <= 0 || p.Wo <= 0) return SS_ERR_ARG;
    p.kt = kt; p.kh = kh; p.kw = kw; p.s = stride; p.pt = pad_t; p.ph = pad_h; p.pw = pad_w;
    long long K = (long long)kt * kh * kw * cin;
    long long M = (long long)n * p.To * p.Ho * p.Wo;
    long long in_elems = (long long)n * t * h * w * cin;
    long long w_elems = (long long)cout * K;
    // 32-bit byte offsets inside one group (buffer addressing); padded halo offsets may be negative but a
    // valid tap always lands inside [0, in_elems)
    if (K >= 65536 || M >= (1ll << 31) || in_elems * 4 >= (1ll << 31) || w_elems * 4 >= (1ll << 31) ||
        M * (long long)(out_cs > cout ? out_cs : cout) * 4 >= (1ll << 32))
        return SS_ERR_UNSUPPORTED;
    p.px_b = cin * 4;
    p.row_b = w * cin * 4;
    p.in_gs = in_gs; p.w_gs = w_gs; p.out_gs = out_gs;
    p.in_bytes = (unsigned)(in_elems * 4);
    p.w_bytes = (unsigned)(w_elems * 4);
    p.relu = relu; p.out_cs = out_cs; p.pool2 = 0;
    return conv_dispatch(p, K, M, groups, kt * kh * kw, ws, ws_floats, (hipStream_t)stream);
}

// conv + bias + ReLU + MaxPool2d(2, 2) for launches that run split-K (ss_conv_workspace_need(...) > 0 and a workspace of that
// size): the pool rides in the split-K reduction; out [groups][n][Ho/2][Wo/2][out_cs].  Anything else: SS_ERR_UNSUPPORTED (the
// caller then runs ss_conv_nhwc + ss_maxpool_nhwc).  2-D (t = kt = 1), no residual.
extern "C" int ss_conv_pool2_nhwc(const float* in, const float* wgt, const float* bias, float* out, int n, int h, int w,
                                  int cin, int cout, int kh, int kw, int stride, int pad_h, int pad_w, int relu, int out_cs,
                                  int groups, long long in_gs, long long w_gs, long long out_gs, float* ws, long long ws_floats,
                                  void* stream) {
    if (!in || !wgt || !out || n <= 0 || h <= 0 || w <= 0 || cin <= 0 || (cin & 3) || cout <= 0 || kh <= 0 || kw <= 0 ||
        kh > 8 || kw > 8 || stride <= 0 || groups <= 0 || out_cs < cout)
        return SS_ERR_ARG;
    const long long need = ss_conv_workspace_need(n, 1, h, w, cin, cout, 1, kh, kw, stride, 0, pad_h, pad_w, groups);
    if (need <= 0 || !ws || ws_floats < need || need >= (1ll << 31)) return SS_ERR_UNSUPPORTED;
    ConvP p;
    p.in = in; p.wgt = wgt; p.bias = bias; p.res = nullptr; p.out = out; p.partial = nullptr;
    p.T = 1; p.H = h; p.W = w; p.C = cin;
    p.To = 1;
    p.Ho = (h + 2 * pad_h - kh) / stride + 1;
    p.Wo = (w + 2 * pad_w - kw) / stride + 1;
    p.Co = cout;
    if (p.Ho < 2 || p.Wo < 2) return SS_ERR_ARG;
    p.kt = 1; p.kh = kh; p.kw = kw; p.s = stride; p.pt = 0; p.ph = pad_h; p.pw = pad_w;
    long long K = (long long)kh * kw * cin;
    long long M = (long long)n * p.Ho * p.Wo;
    long long in_elems = (long long)n * h * w * cin;
    long long w_elems = (long long)cout * K;
    if (K >= 65536 || M >= (1ll << 31) || in_elems * 4 >= (1ll << 31) || w_elems * 4 >= (1ll << 31)) return SS_ERR_UNSUPPORTED;
    p.px_b = cin * 4;
    p.row_b = w * cin * 4;
    p.in_gs = in_gs; p.w_gs = w_gs; p.out_gs = out_gs;
    p.in_bytes = (unsigned)(in_elems * 4);
    p.w_bytes = (unsigned)(w_elems * 4);
    p.relu = relu; p.out_cs = out_cs; p.pool2 = 1;
    return conv_dispatch(p, K, M, groups, kh * kw, ws, ws_floats, (hipStream_t)stream);
}

// Tile choice, split-K plan and launch of a filled ConvP (shared by ss_conv_nhwc and ss_conv_stem3)
static int conv_dispatch(ConvP& p, long long K, long long M, int groups, int taps, float* ws, long long ws_floats,
                         hipStream_t st) {
    const int cout = p.Co;
    const int kt = p.kt, kh = p.kh, kw = p.kw, cin = p.C;
    (void)kt; (void)kh; (void)kw;
    p.K = (int)K; p.M = (int)M;
    p.divC = ss_fastdiv_make((uint32_t)cin);
    p.divKw = ss_fastdiv_make((uint32_t)kw);
    p.divKh = ss_fastdiv_make((uint32_t)kh);
    p.divWo = ss_div32_make((uint32_t)p.Wo);
    p.divHo = ss_div32_make((uint32_t)p.Ho);
    p.divTo = ss_div32_make((uint32_t)p.To);
    const int nk = ss_cdiv(K, 32);
    p.splits = 1;
    p.tiles_per_split = nk;
#ifdef SS_TUNING
    p.ablate = g_ablate;
    p.dbg = g_dbg;
#endif

    // Tile choice.  Measured on MI355X (tools/ab_conv.py, interleaved in-process A/B on the layer shapes of the pipeline):
    // 64x64 with ONE 18 KB LDS buffer is the best or equal-best tile on every shape except large launches with
    // Cout % 128 == 0, where 64x128 wins 2-3 % (below).  128x64, 128x128, 256x64, BK = 64 and a double-buffered LDS
    // variant were all measured equal or slower (throughput is flat between 3 and 7 resident workgroups per CU, so
    // occupancy does not separate them; the smaller tile fills 256 CUs better and has the shortest prologue).  Three
    // of them stay selectable for tuning (ss_debug_set key 0 / SS_CONV_TILE): 5 = 128x128, 7 = 128x64, 16 = 64x128.
    int best = 6;
    const int force = g_force_tile;   // tuning aid only
    if (force) best = force;
    if (p.pool2) {                    // the pool lives in the split-K reduction: only launches that split (decided BEFORE anything is launched)
        const int splits = conv_splits(M, cout, groups, nk);
        const long long need = (long long)groups * splits * M * cout;
        if (best != 6 || !(splits > 1 && ws && need <= ws_floats && need < (1ll << 31))) return SS_ERR_UNSUPPORTED;
    }
    const bool tail = (K % 32) != 0 && (K % 32) < 16;
#ifdef SS_TUNING
    if (best == 5) {
        launch_auto<2, 2, 2, 2, 1, 32, 1>(p, groups, st, taps, false);
    } else if (best == 7) {
        launch_auto<2, 2, 2, 1, 1, 32, 1>(p, groups, st, taps, false);
    } else if (best == 16) {
        launch_auto<2, 2, 1, 2, 1, 32, 1>(p, groups, st, taps, false);
    } else
#endif
    {
        // default: 64x64 tiles, single LDS buffer; small problems are additionally split along K so that ~512
        // workgroups exist
        const int splits = conv_splits(M, cout, groups, nk);
        const long long need = (long long)groups * splits * M * cout;
        if (splits > 1 && ws && need <= ws_floats && need < (1ll << 31)) {
            p.tiles_per_split = ss_cdiv(nk, splits);
            p.splits = ss_cdiv(nk, p.tiles_per_split);
            p.partial = ws;
        }
        // Cout a multiple of 128 and at least one full round of 64x128 workgroups: each input tile is staged once for
        // 128 filters instead of twice for 64 (+2-3 % on the 128-channel layers; with fewer workgroups the smaller tile
        // fills the chip better: layer3, 1380 workgroups, loses 6 %)
        const long long b128 = (long long)ss_cdiv(M, 64) * (cout / 128) * groups;
        // otherwise 128x64 (each filter tile staged once for 128 rows, half the prologues) when at least one full round of
        // those exists: +2-4 % on conv1 / layer1 / the 124-channel regressor conv
        const long long b128m = (long long)ss_cdiv(M, 128) * ss_cdiv(cout, 64) * groups;
        if (!force && p.splits == 1 && cout % 128 == 0 && b128 >= 2048 && !tail)
            launch_auto<2, 2, 1, 2, 1, 32, 1>(p, groups, st, taps, false);
        else if (!force && p.splits == 1 && b128m >= 2048)
            launch_auto<2, 2, 2, 1, 1, 32, 1>(p, groups, st, taps, tail);
        else
            launch_auto<2, 2, 1, 1, 1, 32, 1>(p, groups, st, taps, tail);   // loads pinned ahead of the MFMAs: +2-3 %
        if (p.splits > 1) {
            long long per = p.pool2 ? (M / ((long long)p.Ho * p.Wo)) * (p.Ho / 2) * (p.Wo / 2) * cout : M * cout;
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3(ss_cdiv(per, 256), groups), dim3(256), 0, st, p);
        }
    }
    return ss_launch_status();
}

// ------------------------------------------------------------------------------------------------
// The network stem: 7x7 / stride 2 / pad 3 convolution of the 3-channel frame (spatial_network.py:127-129).  With the
// 4-channel NHWC input of ss_conv_nhwc every filter tap carries a zero channel: K = 49 x 4 = 196 for 147 real products
// (25 % of the MFMAs multiply zeros).  Here the input is [n][H][W + 8][3] (ss_nchw_to_nhwc3_padded: 3 zero pixels left,
// 5 right), so the 7 x 3 = 21 values of one filter ROW are contiguous in memory; a row is padded to 24 (the 22nd..24th
// read the next pixel against zero weights) and the K axis becomes (kh, 24): K = 168, 12.5 % padding, and the tap table
// has 7 entries (one per filter row: row validity is the only halo left, the x halo is real zeros in the buffer).
// Same kernel (conv_igemm_kernel), different pitches: x-neighbours are 12 bytes apart, a "tap" is 24 channels wide.
//   wgt [cout][7][24]: wgt[co][dh][3 dw + c] = w[co][c][dh][dw], entries 21..23 zero  (layers.pack_stem3)
extern "C" int ss_conv_stem3(const float* in_padded, const float* wgt, const float* bias, float* out, int n, int h, int w,
                             int cout, int relu, int out_cs, int groups, long long in_gs, long long w_gs, long long out_gs,
                             void* stream) {
    if (!in_padded || !wgt || !out || n <= 0 || h <= 0 || w <= 0 || cout <= 0 || groups <= 0 || out_cs < cout)
        return SS_ERR_ARG;
    const int wp = w + 8;
    ConvP p;
    p.in = in_padded; p.wgt = wgt; p.bias = bias; p.res = nullptr; p.out = out; p.partial = nullptr;
    p.T = 1; p.H = h; p.W = wp; p.C = 24;
    p.To = 1;
    p.Ho = (h + 6 - 7) / 2 + 1;
    p.Wo = (w + 6 - 7) / 2 + 1;
    p.Co = cout;
    if (p.Ho <= 0 || p.Wo <= 0) return SS_ERR_ARG;
    p.kt = 1; p.kh = 7; p.kw = 1; p.s = 2; p.pt = 0; p.ph = 3; p.pw = 0;      // x padding lives in the buffer
    const long long K = 7 * 24, M = (long long)n * p.Ho * p.Wo;
    const long long in_elems = (long long)n * h * wp * 3, w_elems = (long long)cout * K;
    if (M >= (1ll << 31) || in_elems * 4 >= (1ll << 31) || M * (long long)out_cs * 4 >= (1ll << 32)) return SS_ERR_UNSUPPORTED;
    p.px_b = 3 * 4;
    p.row_b = wp * 3 * 4;
    p.in_gs = in_gs; p.w_gs = w_gs; p.out_gs = out_gs;
    p.in_bytes = (unsigned)(in_elems * 4);
    p.w_bytes = (unsigned)(w_elems * 4);
    p.relu = relu; p.out_cs = out_cs; p.pool2 = 0;
    return conv_dispatch(p, K, M, groups, 7, nullptr, 0, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// max pooling, nhwc, floor mode, -inf padding
// out2 != nullptr: the upper half of the channels goes to out2 (both outputs c4/2 float4 wide): the shared conv1 of the
// SpatialNet / TemporalNet stems writes 64 + 64 channels, each trunk continues from its own pooled tensor
__global__ void maxpool_kernel(const float* __restrict__ in, float* __restrict__ out, float* __restrict__ out2, int n,
                               int h, int w, int c4, int ho, int wo, int k, int s, int pad) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long total = (long long)n * ho * wo * c4;
    if (idx >= total) return;
    int cq = (int)(idx % c4);
    long long r = idx / c4;
    int x = (int)(r % wo);
    r /= wo;
    int y = (int)(r % ho);
    int b = (int)(r / ho);
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    for (int dy = 0; dy < k; ++dy) {
        int yy = y * s - pad + dy;
        if ((unsigned)yy >= (unsigned)h) continue;
        for (int dx = 0; dx < k; ++dx) {
            int xx = x * s - pad + dx;
            if ((unsigned)xx >= (unsigned)w) continue;
            float4 v = reinterpret_cast<const float4*>(in)[(((long long)b * h + yy) * w + xx) * c4 + cq];
            m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        }
    }
    if (out2) {
        const int half = c4 >> 1;
        const long long pixel = idx / c4;
        if (cq < half) reinterpret_cast<float4*>(out)[pixel * half + cq] = m;
        else reinterpret_cast<float4*>(out2)[pixel * half + cq - half] = m;
    } else {
        reinterpret_cast<float4*>(out)[idx] = m;
    }
}

extern "C" int ss_maxpool_nhwc(const float* in, float* out, int n, int h, int w, int c, int k, int stride, int pad,
                               void* stream) {
    if (!in || !out || n <= 0 || h <= 0 || w <= 0 || c <= 0 || (c & 3) || k <= 0 || stride <= 0) return SS_ERR_ARG;
    int ho = (h + 2 * pad - k) / stride + 1, wo = (w + 2 * pad - k) / stride + 1;
    if (ho <= 0 || wo <= 0) return SS_ERR_ARG;
    long long total = (long long)n * ho * wo * (c / 4);
    hipLaunchKernelGGL(maxpool_kernel, dim3(ss_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, in, out,
                       (float*)nullptr, n, h, w, c / 4, ho, wo, k, stride, pad);
    return ss_launch_status();
}

extern "C" int ss_maxpool_nhwc_split(const float* in, float* out0, float* out1, int n, int h, int w, int c, int k,
                                     int stride, int pad, void* stream) {
    if (!in || !out0 || !out1 || n <= 0 || h <= 0 || w <= 0 || c <= 0 || (c & 7) || k <= 0 || stride <= 0)
        return SS_ERR_ARG;
    int ho = (h + 2 * pad - k) / stride + 1, wo = (w + 2 * pad - k) / stride + 1;
    if (ho <= 0 || wo <= 0) return SS_ERR_ARG;
    long long total = (long long)n * ho * wo * (c / 4);
    hipLaunchKernelGGL(maxpool_kernel, dim3(ss_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, in, out0, out1, n, h,
                       w, c / 4, ho, wo, k, stride, pad);
    return ss_launch_status();
}

// ------------------------------------------------------------------------------------------------
// FC: one wave per output neuron, weight row streamed once with 16-byte loads, up to MT batch rows
// accumulated per pass (x comes from L2).
template <int MT>
__global__ __launch_bounds__(256) void linear_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ b, float* __restrict__ y, int m,
                                                     int k, int nout, int relu) {
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int o = blockIdx.x * 4 + wave;
    if (o >= nout) return;
    const float* wr = w + (long long)o * k;
    int m0 = blockIdx.y * MT;
    float acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = 0.f;
    if ((k & 3) == 0) {
        int k4 = k >> 2;
        const float4* xr[MT];           // rows past m re-read row m - 1, dropped at the store: no branches around the loads
#pragma unroll
        for (int i = 0; i < MT; ++i) xr[i] = reinterpret_cast<const float4*>(x + (long long)(m0 + i < m ? m0 + i : m - 1) * k);
        for (int q = lane; q < k4; q += 64) {
            float4 wv = reinterpret_cast<const float4*>(wr)[q];
            float4 xv[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) xv[i] = xr[i][q];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                acc[i] = fmaf(wv.x, xv[i].x, acc[i]);
                acc[i] = fmaf(wv.y, xv[i].y, acc[i]);
                acc[i] = fmaf(wv.z, xv[i].z, acc[i]);
                acc[i] = fmaf(wv.w, xv[i].w, acc[i]);
            }
        }
    } else {
        for (int q = lane; q < k; q += 64) {
            float wv = wr[q];
#pragma unroll
            for (int i = 0; i < MT; ++i)
                if (m0 + i < m) acc[i] = fmaf(wv, x[(long long)(m0 + i) * k + q], acc[i]);
        }
    }
    float bias = b ? b[o] : 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        float v = ss_wave_sum(acc[i]);
        if (lane == 0 && m0 + i < m) {
            v += bias;
            if (relu) v = fmaxf(v, 0.f);
            y[(long long)(m0 + i) * nout + o] = v;
        }
    }
}

// G fully connected layers of identical shape in ONE launch (blockIdx.z = group): x [G][m][k] (group stride x_gs floats),
// w [G][nout][k], b [G][nout] | null, group g's result [m][nout] at y.p[g] -- the regressors of SpatialNet's two stage-2
// heads and TemporalNet's two views share every launch (layers.run_regressor_quad), and each head's last layer lands where
// its consumer reads it.
struct LinearOuts {
    float* p[8];
};
template <int MT>
__global__ __launch_bounds__(256) void linear_grouped_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ b, LinearOuts y, int m, int k,
                                                             int nout, int relu, long long x_gs) {
    const int g = blockIdx.z;
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int o = blockIdx.x * 4 + wave;
    if (o >= nout) return;
    x += (long long)g * x_gs;
    const float* wr = w + ((long long)g * nout + o) * k;
    float* yg = g == 0 ? y.p[0] : g == 1 ? y.p[1] : g == 2 ? y.p[2] : g == 3 ? y.p[3] : g == 4 ? y.p[4] : g == 5 ? y.p[5] : g == 6 ? y.p[6] : y.p[7];
    int m0 = blockIdx.y * MT;
    float acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = 0.f;
    int k4 = k >> 2;
    // rows past m re-read row m - 1 (results dropped at the store): no branch around the loads -- with `if (m0 + i < m)` the
    // compiler put every row's load and its four FMAs into an exec-mask region of its own, MT load latencies in series per step
    const float4* xr[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) xr[i] = reinterpret_cast<const float4*>(x + (long long)(m0 + i < m ? m0 + i : m - 1) * k);
    for (int q = lane; q < k4; q += 64) {
        float4 wv = reinterpret_cast<const float4*>(wr)[q];
        float4 xv[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) xv[i] = xr[i][q];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            acc[i] = fmaf(wv.x, xv[i].x, acc[i]);
            acc[i] = fmaf(wv.y, xv[i].y, acc[i]);
            acc[i] = fmaf(wv.z, xv[i].z, acc[i]);
            acc[i] = fmaf(wv.w, xv[i].w, acc[i]);
        }
    }
    float bias = b ? b[(long long)g * nout + o] : 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        float v = ss_wave_sum(acc[i]);
        if (lane == 0 && m0 + i < m) {
            v += bias;
            if (relu) v = fmaxf(v, 0.f);
            yg[(long long)(m0 + i) * nout + o] = v;
        }
    }
}

extern "C" int ss_linear_grouped(const float* x, long long x_group_stride, const float* w, const float* b,
                                 float* const* y_groups, int groups, int m, int k, int nout, int relu, void* stream) {
    if (!x || !w || !y_groups || groups <= 0 || groups > 8 || m <= 0 || k <= 0 || (k & 3) || nout <= 0) return SS_ERR_ARG;
    LinearOuts y;
    for (int g = 0; g < 8; ++g) {
        y.p[g] = g < groups ? y_groups[g] : nullptr;
        if (g < groups && !y.p[g]) return SS_ERR_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    // (the same accumulation order as linear_kernel<MT> for the same m: a row's result does not depend on the grouping)
    if (m <= 1) hipLaunchKernelGGL((linear_grouped_kernel<1>), dim3(ss_cdiv(nout, 4), 1, groups), dim3(256), 0, st, x, w, b, y, m, k, nout, relu, x_group_stride);
    else if (m <= 4) hipLaunchKernelGGL((linear_grouped_kernel<4>), dim3(ss_cdiv(nout, 4), 1, groups), dim3(256), 0, st, x, w, b, y, m, k, nout, relu, x_group_stride);
    else hipLaunchKernelGGL((linear_grouped_kernel<8>), dim3(ss_cdiv(nout, 4), ss_cdiv(m, 8), groups), dim3(256), 0, st, x, w, b, y, m, k, nout, relu, x_group_stride);
    return ss_launch_status();
}

extern "C" int ss_linear(const float* x, const float* w, const float* b, float* y, int m, int k, int nout, int relu,
                         void* stream) {
    if (!x || !w || !y || m <= 0 || k <= 0 || nout <= 0) return SS_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (m <= 1) {
        hipLaunchKernelGGL((linear_kernel<1>), dim3(ss_cdiv(nout, 4), 1), dim3(256), 0, st, x, w, b, y, m, k, nout, relu);
    } else if (m <= 4) {
        hipLaunchKernelGGL((linear_kernel<4>), dim3(ss_cdiv(nout, 4), 1), dim3(256), 0, st, x, w, b, y, m, k, nout, relu);
    } else {
        hipLaunchKernelGGL((linear_kernel<8>), dim3(ss_cdiv(nout, 4), ss_cdiv(m, 8)), dim3(256), 0, st, x, w, b, y, m, k,
                           nout, relu);
    }
    return ss_launch_status();
}

// ------------------------------------------------------------------------------------------------
// layout plumbing
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out, int n, int c, int h, int w,
                                    int cp) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long total = (long long)n * h * w * cp;
    if (idx >= total) return;
    int ch = (int)(idx % cp);
    long long pix = idx / cp;
    long long hw = (long long)h * w;
    int b = (int)(pix / hw);
    long long rem = pix - (long long)b * hw;
    out[idx] = ch < c ? in[((long long)b * c + ch) * hw + rem] : 0.f;
}

// c <= 4 -> 4 channels (the network inputs): one thread per pixel, plane reads coalesced across lanes, one 16-byte store
__global__ __launch_bounds__(256) void nchw_to_nhwc4_kernel(const float* __restrict__ in, float4* __restrict__ out, int c,
                                                            int hw, long long pixels) {
    long long pix = (long long)blockIdx.x * 256 + threadIdx.x;
    if (pix >= pixels) return;
    long long b = pix / hw;
    int rem = (int)(pix - b * hw);
    const float* s = in + b * c * hw + rem;
    float4 v = make_float4(s[0], 0.f, 0.f, 0.f);
    if (c > 1) v.y = s[hw];
    if (c > 2) v.z = s[2ll * hw];
    if (c > 3) v.w = s[3ll * hw];
    out[pix] = v;
}

// [n][3][h][w] -> [n][h][w + 8][3] with 3 zero pixels on the left and 5 on the right (the stem's x halo + row padding)
__global__ __launch_bounds__(256) void nchw_to_nhwc3_padded_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                   int h, int w, long long total) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;      // one thread per (image, row, padded x)
    if (idx >= total) return;
    const int wp = w + 8;
    const int xp = (int)(idx % wp);
    const long long r = idx / wp;                  // image * h + row
    const int y = (int)(r % h);
    const long long b = r / h;
    const int x = xp - 3;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f;
    if ((unsigned)x < (unsigned)w) {
        const float* s = in + (b * 3 * h + y) * (long long)w + x;
        const long long hw = (long long)h * w;
        v0 = s[0]; v1 = s[hw]; v2 = s[2 * hw];
    }
    float* o = out + idx * 3;
    o[0] = v0; o[1] = v1; o[2] = v2;
}

extern "C" int ss_nchw_to_nhwc3_padded(const float* in, float* out, int n, int h, int w, void* stream) {
    if (!in || !out || n <= 0 || h <= 0 || w <= 0) return SS_ERR_ARG;
    const long long total = (long long)n * h * (w + 8);
    hipLaunchKernelGGL(nchw_to_nhwc3_padded_kernel, dim3(ss_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, in, out, h,
                       w, total);
    return ss_launch_status();
}

__global__ void nhwc_to_nchw_kernel(const float* __restrict__ in, float* __restrict__ out, int n, int c, int h, int w,
                                    int cs) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long hw = (long long)h * w;
    long long total = (long long)n * c * hw;
    if (idx >= total) return;
    long long rem = idx % hw;
    long long bc = idx / hw;
    int ch = (int)(bc % c);
    int b = (int)(bc / c);
    out[idx] = in[((long long)b * hw + rem) * cs + ch];
}

extern "C" int ss_nchw_to_nhwc(const float* in, float* out, int n, int c, int h, int w, int c_pad, void* stream) {
    if (!in || !out || n <= 0 || c <= 0 || h <= 0 || w <= 0 || c_pad < c) return SS_ERR_ARG;
    long long total = (long long)n * h * w * c_pad;
    if (c_pad == 4 && (((uintptr_t)out) & 15) == 0) {
        long long pixels = (long long)n * h * w;
        hipLaunchKernelGGL(nchw_to_nhwc4_kernel, dim3(ss_cdiv(pixels, 256)), dim3(256), 0, (hipStream_t)stream, in,
                           reinterpret_cast<float4*>(out), c, h * w, pixels);
        return ss_launch_status();
    }
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(ss_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, in, out, n, c,
                       h, w, c_pad);
    return ss_launch_status();
}

extern "C" int ss_nhwc_to_nchw(const float* in, float* out, int n, int c, int h, int w, int c_stride, void* stream) {
    if (!in || !out || n <= 0 || c <= 0 || h <= 0 || w <= 0 || c_stride < c) return SS_ERR_ARG;
    long long total = (long long)n * c * h * w;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(ss_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, in, out, n, c,
                       h, w, c_stride);
    return ss_launch_status();
}

extern "C" int ss_version(void) { return 600; }

extern "C" const char* ss_error_string(int code) {
    switch (code) {
        case SS_OK: return "ok";
        case SS_ERR_ARG: return "bad argument";
        case SS_ERR_LAUNCH: return "kernel launch failed";
        case SS_ERR_UNSUPPORTED: return "unsupported size";
        case SS_ERR_DEVICE: return "kernel not available on this device";
        default: return "unknown error";
    }
}
