// 3x3 / stride-1 / pad-1 convolution as FUSED Winograd F(2x2,3x3) on the fp32 matrix cores (gfx950).
//
// The stride-1 3x3 layers are ~80 % of the path's 41 GFLOP per frame (ResNet-18 layer1..3 bodies, the regressor
// convs; spatial_network.py:132-136,147-209, temporal_network.py:65-93).  Winograd's minimal filtering computes a
// 2x2 output tile from a 4x4 input tile with 16 instead of 36 multiplications per (cin, cout) pair:
//     Y = A^T [ (G g G^T) .* (B^T d B) ] A        (Lavin & Gray 2015; all in fp32, filters transformed in fp64)
// so the contraction over cin becomes 16 independent GEMMs (one per position of the 4x4 transform domain) with 2.25x
// fewer MFMA flops than the implicit GEMM of conv.hip.  An unfused Winograd would write and re-read the transformed
// input and the pre-output (4x the activation bytes each) through HBM and lose to the direct kernel on every layer of
// this network (64..256 channels); here both transforms live inside the GEMM kernel:
//
//   workgroup = 256 threads = 4 waves; tile = 32 output tiles (TBH x TBW block of 2x2 tiles = 2TBH x 2TBW pixels of ONE
//   image) x 64 output channels x all 16 positions; K loop over cin in chunks of 16.  Wave i owns ROW i of the 4x4
//   transform domain (positions 4i..4i+3) for all 32 tiles and all 64 output channels: 4 x 2 accumulator tiles of
//   v_mfma_f32_32x32x2_f32 = 128 registers.
//   per chunk:
//     * raw (2TBH+2) x (2TBW+2) x 16ch input patch: coalesced 16-byte buffer loads issued two chunks ahead (halo / M tail
//       / channel tail through the descriptor's bounds check: invalid lanes get offset 0xFFFFFFFF and read zeros),
//       registers -> LDS transposed to channel planes [channel][row][col] (two buffers, one barrier per chunk);
//     * input transform B^T d B IN THE MFMA OPERAND LAYOUT: lane (tile = lane & 31, k half = lane >> 5) reads rows
//       (ra, rb) of its tile's 4x4 patch for its 8 channels (16 conflict-free 8-byte-pair LDS reads), one packed
//       add/sub for the row stage, four scalar add/subs per channel for the column stage -> the 32 A-operand registers
//       of the wave's 4 positions.  The transformed input never exists in memory (first version: 128 KB of LDS
//       reads per chunk for the A operands, LDS-bound);
//     * 64 MFMAs per wave: B operand = transformed filters PRE-PACKED in the register layout of the instruction, global
//       -> VGPR, fully coalesced 1 KB loads, prefetched three 8-MFMA blocks (1536 MFMA cycles) ahead; each filter
//       element is needed by exactly one wave of the workgroup, so nothing is lost by skipping LDS;
//     all of it as ONE instruction stream per wave: the chunk is 8 blocks of 8 MFMAs, and the transform of the channels
//     the next blocks need, the LDS staging of the next chunk and the filter loads sit between the MFMAs (see STREAM below);
//   epilogue: wave i forms T[i][b] = sum_j M[i][j] A[j][b] in registers, stages it in LDS; then each thread sums the
//   three T rows of its pixels in a fixed order (A^T), adds bias (folded BatchNorm) / residual, applies ReLU and stores
//   whole 256-byte pixel rows with 16-byte accesses (out-of-range offsets for pixels outside the image: no branches).
// Two workgroups per CU (64 KB LDS, <= 256 registers): one workgroup's prologue / epilogue runs under the other's stream.
#include "common.h"
#include <type_traits>

typedef float w_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned w_u32x4 __attribute__((ext_vector_type(4)));

struct WinoP {
    const float* in;
    const float* U;          // packed transformed filters, see wino_pack_kernel
    const float* bias;
    const float* res;
    float* out;
    int N, H, W, C, Co;      // C = input channels (multiple of 4), Co = output channels (multiple of 64)
    int nchunk;              // ceil(C / 16)
    int relu, out_cs;
    SsDiv32 divBx, divBy;    // m-block index -> (image, block row, block column)
    unsigned nbx, nby;       // tile blocks per image row / column
    unsigned ncb;            // 64-channel output blocks
    unsigned nmb, njobs;     // pair kernel: tile blocks in total; (pair of tile blocks, cout block) jobs of one group
    SsDiv32 divNcb;
    long long in_gs, u_gs, out_gs;      // element strides between groups
    unsigned in_bytes, out_bytes, u_bytes;
#ifdef SS_TUNING
    unsigned long long* dbg;            // per-workgroup phase stamps (tools/diag_wino.py)
    int ablate;                         // 1: no filter loads, 2: no raw loads / LDS / transform, 4: no epilogue (wrong results)
    int knob[4];                        // round-3 experiments (ss_debug_set keys 16..19): [0] first-round stagger of the second
                                        // workgroup of a CU in units of 1024 clocks, [1] epilogue priority + 1 (0 = default 3),
                                        // [2] prologue priority + 1 (0 = default 3), [3] K-loop priority + 1 (0 = default 0)
#endif
};

#ifdef SS_TUNING
#define W_STAMP(i) do { if (p.dbg) ts[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define W_ABLATE(bit) (p.ablate & (bit))
#define W_SETPRIO_KNOB(k, dflt) do { const int v_ = p.knob[k]; if (v_ == 1) __builtin_amdgcn_s_setprio(0); else if (v_ == 2) __builtin_amdgcn_s_setprio(1); \
        else if (v_ == 3) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(dflt); } while (0)
extern int g_wino_ablate;
extern int g_wino_knob[4];
#else
#define W_STAMP(i) do { } while (0)
#define W_ABLATE(bit) 0
#define W_SETPRIO_KNOB(k, dflt) __builtin_amdgcn_s_setprio(dflt)
#endif

__device__ __forceinline__ __amdgpu_buffer_rsrc_t w_rsrc(const float* base, unsigned bytes) {
    unsigned long long a = (unsigned long long)base;
    unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    void* ub = (void*)(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(ub, 0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

typedef float w_f32x16 __attribute__((ext_vector_type(16)));
typedef float w_f32x2 __attribute__((ext_vector_type(2)));

// Two schedules of the K loop (template parameter STREAM; same tiles, prologue and epilogue, bit-identical results):
//   STREAM (dispatched): every wave runs ONE instruction stream in which its own input transform, LDS staging and filter
//     loads sit in the issue slots between its MFMAs (sched_group_barrier).  Measured against the alternating schedule
//     (tools/cmp_wino_variants.py): layer1 / 2 / 3 and the regressor shape 4 / 10 / 7 / 23 % faster, MFMA pipe 64 -> 69 %
//     busy over the Winograd launches of a clip, +2 % frames/s.
//   alternating (first version; tuning build only, ss_debug_set(7, 1)): per chunk a transform phase at raised priority,
//     then 64 MFMAs with the filter loads pinned between them.  The matrix pipe idles whenever both resident waves of a
//     SIMD are outside their MFMA runs at the same time, and a workgroup whose neighbour is in its prologue / epilogue
//     (a quarter of a workgroup's life on layer1) keeps the pipe only ~60 % busy on its own.
// NB = 32-channel output blocks per workgroup: 2 (64 channels, 128 accumulator registers, two workgroups per CU) or
// 1 (32 channels, 64 accumulator registers, THREE workgroups per CU: the epilogue / prologue of one workgroup hides
// behind the MFMAs of two others -- the short-K layers (cin = 64: 4 chunks) spend a quarter of a workgroup's life there)
// VAR (stream schedule only): 0 = round-2 placement of the loads; 1 = the filter loads of block (1, 2) are issued at the END of
// block (0, 2), i.e. BEFORE the next chunk's raw-patch loads (buffer loads return in order: a wait for a filter load also waits
// for every older load, and the raw patch mostly comes from HBM)
template <int TBH, int TBW, int NB, bool RES, bool STREAM, bool SLICED = false, int VAR = 0>
__global__ __launch_bounds__(256, NB == 1 ? 3 : 2) void conv_wino_kernel(WinoP p) {
    static_assert(TBH * TBW == 32, "32 tiles per workgroup");
    static_assert(!STREAM || NB == 2, "the stream schedule is written for 64-channel blocks");
    static_assert(!SLICED || (NB == 2 && !STREAM), "sliced products: 64-channel blocks, phase-alternating schedule");
    constexpr int BN = 32 * NB;                             // output channels per workgroup
    constexpr int RH = 2 * TBH + 2, RW = 2 * TBW + 2;      // raw input patch (pixels)
    constexpr int RPIX = RH * RW;
    constexpr int RWP = TBW == 4 ? 12 : 24;                 // LDS row pitch of a channel plane (dwords): see `transform`
    constexpr int PLANE = RH * RWP + 4;                     // channel plane stride; = 4 (mod 8): 2 lanes per bank on the raw store
    constexpr int NE = (RPIX * 4 + 255) / 256;              // 16-byte raw items per thread
    constexpr int RAWF = 16 * PLANE;                        // dwords of one raw buffer (16 channel planes)
    constexpr int SMEMF = 8192 * NB;                        // T stage: 4 waves x 2 x 32 tiles x BN channels
    static_assert(2 * RAWF <= SMEMF, "two raw buffers fit the block");
    // two raw buffers ([channel][row][col], channel-major) during the K loop, the T stage in the epilogue
    __shared__ __attribute__((aligned(16))) float smem[SMEMF];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef SS_TUNING
    unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    // Setup and epilogue are VALU / LDS work that shares its SIMD with the OTHER resident workgroup's MFMAs; at equal
    // priority every such instruction waits for an MFMA boundary.  Raised priority lets it through (as in conv.hip).
#ifdef SS_TUNING
    // experiment: the second workgroup of a CU (LDS base != 0) of the FIRST round sleeps, so that the two resident workgroups
    // run in anti-phase (one's prologue / epilogue under the other's K loop) instead of in lock-step
    if (p.knob[0] > 0 && blockIdx.x < 512u && (__builtin_amdgcn_s_getreg((31 << 11) | 6) & 0x1FFu) != 0u) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)p.knob[0] * 1024ull) __builtin_amdgcn_s_sleep(32);
    }
#endif
    W_SETPRIO_KNOB(2, 3);
    W_STAMP(0);

    // XCD-aware block order (as conv.hip): consecutive workgroups go round-robin to the 8 XCDs; give every XCD one
    // contiguous run of m-blocks (all cout blocks of an m-block back to back) so halo rows and the cout-block re-reads of
    // an input patch hit that XCD's L2
    unsigned lin = blockIdx.x;
    {
        const unsigned nwg = gridDim.x;
        if (nwg >= 16) {
            const unsigned q = nwg / 8, r = nwg % 8, xcd = lin % 8, idx = lin / 8;
            lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        }
    }
    const unsigned mb = ss_div32(lin, p.divNcb);
    const unsigned cbk = lin - mb * p.ncb;                  // output block (BN channels)
    const unsigned t1 = ss_div32(mb, p.divBx);
    const int bx = (int)(mb - t1 * p.nbx);
    const unsigned img = ss_div32(t1, p.divBy);
    const int by = (int)(t1 - img * p.nby);
    const int oy0 = by * (2 * TBH), ox0 = bx * (2 * TBW);  // first output pixel of the block
    const int grp = blockIdx.z;

    const __amdgpu_buffer_rsrc_t rin = w_rsrc(p.in + (long long)grp * p.in_gs, p.in_bytes);
    const __amdgpu_buffer_rsrc_t ru = w_rsrc(p.U + (long long)grp * p.u_gs, p.u_bytes);

    // raw items of this thread: (pixel, channel quad) -> byte offset of channel 0 of the quad, or invalid
    unsigned rbase[NE], rinv[NE];
    int rlds[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int item = tid + 256 * e;
        const int pix = item >> 2, q = item & 3;
        if constexpr (VAR == 4) {        // experiment (timing only, wrong results): the prologue without its index arithmetic
            rbase[e] = (unsigned)(tid * 16 + e * 4096); rinv[e] = 0u; rlds[e] = (tid & 3) * 4 * PLANE + (tid >> 2) + e * 64;
            continue;
        }
        const int ry = pix / RW, rx = pix - ry * RW;
        const int iy = oy0 - 1 + ry, ix = ox0 - 1 + rx;
        const bool ok = item < RPIX * 4 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        rbase[e] = ((((unsigned)img * p.H + iy) * p.W + ix) * (unsigned)p.C + 4u * q) * 4u;
        rinv[e] = ok ? 0u : 0xFFFFFFFFu;
        // channel 4q of this pixel (the quad's channels are PLANE apart); items past the patch (RPIX * 4 is not a multiple
        // of 256) land in the 4-dword pad behind their planes: no branch around the LDS writes
        rlds[e] = (4 * q) * PLANE + (item < RPIX * 4 ? ry * RWP + rx : RH * RWP);
    }
    const int myq = tid & 3;                                // channel quad of this thread's raw items

    // This lane in the GEMMs (v_mfma_f32_32x32x2_f32: A[i = lane & 31][k = lane >> 5]): tile `lane & 31`, channels
    // 8 * (lane >> 5) .. + 7 of the chunk; the wave's transform row = its position row.
    const int kh = lane >> 5;
    const int m_tile = lane & 31;
    const int m_ty = m_tile / TBW, m_tx = m_tile - m_ty * TBW;
    // top-left dword of the tile's 4x4 patch in channel plane 8 kh.  Bank picture of one 16-lane group of the 8-byte
    // reads (16 tiles, same channel): (8,4) blocks: 2 tx + 24 ty covers all 32 banks once (row pitch 12);
    // (4,8) blocks: 2 tx (0..14) + 48 ty = +16 (row pitch 24).
    const int t_src = (8 * kh) * PLANE + (2 * m_ty) * RWP + 2 * m_tx;
    const int t_ra = wave == 0 ? 0 : (wave == 2 ? 2 : 1);  // B^T rows: 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3
    const int t_rb = wave == 3 ? 3 : (wave == 2 ? 1 : 2);

    // packed filters: [cout/32][chunk][pos][half][lane][4] floats -> 32 KB per (cout block, chunk), 2 KB per position
    const unsigned u_lane = (unsigned)lane * 16u;
    const unsigned u_wave = (cbk * NB) * (unsigned)p.nchunk * 32768u + (unsigned)wave * 8192u;     // + blk * nchunk * 32 KB
    const unsigned u_blk = (unsigned)p.nchunk * 32768u;

    // epilogue addressing, and the residual's loads (issued inside the last chunk by the stream schedule)
    float* __restrict__ out = p.out + (long long)grp * p.out_gs;
    const __amdgpu_buffer_rsrc_t rout = w_rsrc(out, p.out_bytes);
    const __amdgpu_buffer_rsrc_t rres = w_rsrc(RES ? p.res + (long long)grp * p.out_gs : out, p.out_bytes);
    // this thread's output items: (pixel, channel quad); BN / 4 quads per pixel -> 128 * BN / 4 / 256 = 4 NB items
    constexpr int QP = BN / 4, NI = 4 * NB, PSTEP = 256 / QP;
    const int cq = tid % QP;
    // POOL (VAR = 8; no residual): the 2 x 2 / 2 max-pool that follows the convolution in the regressors (spatial_network.py:
    // conv, ReLU, conv, ReLU, MaxPool2d(2, 2)) taken inside the epilogue -- an F(2x2, 3x3) output tile IS a pooling window.
    // max over the four pixels first, then bias and ReLU (both monotone: bit-identical to pooling the stored map); the
    // un-pooled map is never written.  A thread owns 2 x (tile, 4 couts) instead of 8 x (pixel, 4 couts).
    constexpr bool POOL = STREAM && !RES && VAR == 8;
    constexpr int NIP = 32 * QP / 256;                      // pooled items per thread
    unsigned goff[NI];
    w_u32x4 rv[NI];
    auto res_issue = [&]() {
        if constexpr (POOL) {
            const int hp = p.H >> 1, wp = p.W >> 1;         // MaxPool2d(2, 2): floor
#pragma unroll
            for (int e = 0; e < NIP; ++e) {
                const int tile = e * PSTEP + tid / QP;
                const int ty = tile / TBW, tx = tile - ty * TBW;
                const int oyp = (oy0 >> 1) + ty, oxp = (ox0 >> 1) + tx;
                const bool ok = oyp < hp && oxp < wp;
                goff[e] = ok ? ((((unsigned)img * hp + oyp) * wp + oxp) * (unsigned)p.out_cs + cbk * BN + 4u * cq) * 4u : 0xFFFFFFFFu;
            }
            return;
        }
#pragma unroll
        for (int e = 0; e < NI; ++e) {
            const int px = e * PSTEP + tid / QP;
            const int tile = px >> 2, a = (px >> 1) & 1, b = px & 1;
            const int ty = tile / TBW, tx = tile - ty * TBW;
            const int oy = oy0 + 2 * ty + a, ox = ox0 + 2 * tx + b;
            const bool ok = oy < p.H && ox < p.W;
            goff[e] = ok ? ((((unsigned)img * p.H + oy) * p.W + ox) * (unsigned)p.out_cs + cbk * BN + 4u * cq) * 4u : 0xFFFFFFFFu;
        }
        if (RES) {
#pragma unroll
            for (int e = 0; e < NI; ++e) rv[e] = __builtin_amdgcn_raw_buffer_load_b128(rres, goff[e], 0, 0);
        }
    };

    w_f32x16 acc[4][NB];
    auto acc_clear = [&]() {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    };

    if constexpr (SLICED) {
        // ---- fp32 x fp32 products on the bf16 matrix pipe (opt-in, ss_conv3x3_wino3_nhwc): every operand is the EXACT sum of
        // three bf16 slices (a = a1 + a2 + a3: 3 x 8 significand bits; the filters are sliced once at pack time, the
        // transformed input here), and all nine slice products -- each exact in fp32 -- go through
        // v_mfma_f32_32x32x16_bf16 into the fp32 accumulators: 9 instructions of 8 passes per 16 channels where the fp32
        // pipe needs 8 instructions of 16 passes.  Nothing of a product is dropped; only the order of the fp32
        // accumulation differs from the fp32-MFMA kernel (results agree to fp32 rounding, not bit for bit).
        typedef __bf16 w_bf8 __attribute__((ext_vector_type(8)));
        typedef __bf16 w_bf2 __attribute__((ext_vector_type(2)));
        acc_clear();
        w_f32x4 rr[NE];
        auto raw_issue = [&](int c) {
            const unsigned coff = (unsigned)c * 64u;
            const unsigned cinv = ((c * 16 + 4 * myq) < p.C) ? 0u : 0xFFFFFFFFu;
#pragma unroll
            for (int e = 0; e < NE; ++e)
                rr[e] = __builtin_bit_cast(w_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (rbase[e] + coff) | rinv[e] | cinv, 0, 0));
        };
        // packed filter slices: [cout/32][chunk][pos][slice][lane][8 bf16] -> 3 KB per position, 48 KB per (cout block, chunk)
        const unsigned u3_wave = (cbk * NB) * (unsigned)p.nchunk * 49152u + (unsigned)wave * 12288u;
        const unsigned u3_blk = (unsigned)p.nchunk * 49152u;
        auto u_issue3 = [&](w_u32x4 (&u)[3], int c, int j, int blk) {
            const int so = (int)__builtin_amdgcn_readfirstlane(u3_wave + (unsigned)c * 49152u + (unsigned)j * 3072u + (unsigned)blk * u3_blk);
#pragma unroll
            for (int sl = 0; sl < 3; ++sl) u[sl] = __builtin_amdgcn_raw_buffer_load_b128(ru, u_lane + 1024u * sl, so, 0);
        };
        float av[4][8];
        const w_f32x2 t_sg = wave == 1 ? (w_f32x2){1.f, 1.f} : (w_f32x2){-1.f, -1.f};
        auto transform = [&](const float* buf) {
            const float* src = buf + t_src;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float* pa = src + c * PLANE + t_ra * RWP;
                const float* pb = src + c * PLANE + t_rb * RWP;
                const w_f32x2 xa0 = *reinterpret_cast<const w_f32x2*>(pa), xa1 = *reinterpret_cast<const w_f32x2*>(pa + 2);
                const w_f32x2 xb0 = *reinterpret_cast<const w_f32x2*>(pb), xb1 = *reinterpret_cast<const w_f32x2*>(pb + 2);
                const w_f32x2 r0 = __builtin_elementwise_fma(xb0, t_sg, xa0);
                const w_f32x2 r1 = __builtin_elementwise_fma(xb1, t_sg, xa1);
                const w_f32x2 d = r0 - r1;
                av[0][c] = d[0];
                av[1][c] = r0[1] + r1[0];
                av[2][c] = r1[0] - r0[1];
                av[3][c] = d[1];
            }
        };
        // the 8 fp32 A operands of position j -> three bf16x8 slices (round to nearest even; the residuals are exact)
        w_u32x4 as[3];
        auto split = [&](int j) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const w_f32x2 x = {av[j][2 * q], av[j][2 * q + 1]};
                const w_bf2 p1 = __builtin_convertvector(x, w_bf2);
                const w_f32x2 r = x - __builtin_convertvector(p1, w_f32x2);
                const w_bf2 p2 = __builtin_convertvector(r, w_bf2);
                const w_f32x2 r2 = r - __builtin_convertvector(p2, w_f32x2);
                const w_bf2 p3 = __builtin_convertvector(r2, w_bf2);
                as[0][q] = __builtin_bit_cast(unsigned, p1);
                as[1][q] = __builtin_bit_cast(unsigned, p2);
                as[2][q] = __builtin_bit_cast(unsigned, p3);
            }
        };
        auto mma9 = [&](int j, int blk, const w_u32x4 (&u)[3]) {
#pragma unroll
            for (int sa = 2; sa >= 0; --sa)          // smallest products first
#pragma unroll
                for (int sb = 2; sb >= 0; --sb)
                    acc[j][blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(w_bf8, as[sa]), __builtin_bit_cast(w_bf8, u[sb]),
                                                                        acc[j][blk], 0, 0, 0);
        };
        w_u32x4 ua[3], ub[3];
        raw_issue(0);
        u_issue3(ua, 0, 0, 0);
        W_STAMP(1);
        for (int c = 0; c < p.nchunk; ++c) {
            float* buf = smem + (c & 1) * RAWF;
#pragma unroll
            for (int e = 0; e < NE; ++e)
#pragma unroll
                for (int k = 0; k < 4; ++k) buf[rlds[e] + k * PLANE] = rr[e][k];
            __syncthreads();
            const int cn = c + 1 < p.nchunk ? c + 1 : c;
            raw_issue(cn);
            transform(buf);
            __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                split(j);
                __builtin_amdgcn_sched_barrier(0);
                u_issue3(ub, c, j, 1);
                __builtin_amdgcn_sched_barrier(0);
                mma9(j, 0, ua);
                __builtin_amdgcn_sched_barrier(0);
                if (j < 3) u_issue3(ua, c, j + 1, 0); else u_issue3(ua, cn, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                mma9(j, 1, ub);
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_setprio(3);
        }
    } else
    if constexpr (STREAM) {
        // compile-time ablations of the stream (tuning build, timing only -- results are wrong): VAR = 16 + mask,
        // 1 no filter loads, 2 no raw-patch loads, 4 no LDS staging writes, 8 no LDS reads / transform, 16 no barrier
        constexpr int ABL = VAR >= 16 ? VAR - 16 : 0;
        w_f32x4 rr[NE];
        auto raw_issue = [&](int c) {
            if constexpr (ABL & 2) return;
            const unsigned coff = (unsigned)c * 64u;
            const unsigned cinv = ((c * 16 + 4 * myq) < p.C) ? 0u : 0xFFFFFFFFu;
#pragma unroll
            for (int e = 0; e < NE; ++e)
                rr[e] = __builtin_bit_cast(w_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (rbase[e] + coff) | rinv[e] | cinv, 0, 0));
        };
        auto raw_store = [&](float* buf) {
            if constexpr (ABL & 4) return;
#pragma unroll
            for (int e = 0; e < NE; ++e)
#pragma unroll
                for (int k = 0; k < 4; ++k) buf[rlds[e] + k * PLANE] = rr[e][k];
        };
        // filters of one BLOCK = (chunk c, position 4 wave + j, half h): u[blk] = 4 floats = MFMA steps 4 h .. 4 h + 3.
        // Four register sets: block k multiplies with set k & 3 while the loads of block k + 3 go to the set block k - 1 used.
        w_f32x4 u[4][NB] = {};
        auto u_issue = [&](int set, int c, int j, int h) {
            if constexpr (ABL & 1) return;
            const unsigned base = u_wave + (unsigned)c * 32768u + (unsigned)j * 2048u;
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) {
                const int so = (int)__builtin_amdgcn_readfirstlane(base + (unsigned)blk * u_blk);
                u[set][blk] = __builtin_bit_cast(w_f32x4, __builtin_amdgcn_raw_buffer_load_b128(ru, u_lane + 1024u * h, so, 0));
            }
        };
        float av[4][8];              // A operands: av[j][s] = V[position 4 wave + j][tile][channel 8 kh + s]
        if constexpr (ABL & 8) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < 8; ++q) av[j][q] = (float)(lane + j + q);
        }
        const w_f32x2 t_sg = wave == 1 ? (w_f32x2){1.f, 1.f} : (w_f32x2){-1.f, -1.f};     // row stage: d[ra] + sg * d[rb] (exact)
        // the input transform, one channel at a time: rd() requests the two patch rows, xf() (one block later) turns them
        // into the four A operands of that channel
        w_f32x2 tq[4] = {};
        auto rd = [&](const float* buf, int ch) {
            if constexpr (ABL & 8) return;
            const float* pa = buf + t_src + ch * PLANE + t_ra * RWP;
            const float* pb = buf + t_src + ch * PLANE + t_rb * RWP;
            tq[0] = *reinterpret_cast<const w_f32x2*>(pa);
            tq[1] = *reinterpret_cast<const w_f32x2*>(pa + 2);
            tq[2] = *reinterpret_cast<const w_f32x2*>(pb);
            tq[3] = *reinterpret_cast<const w_f32x2*>(pb + 2);
        };
        auto xf = [&](int ch) {
            if constexpr (ABL & 8) return;
            const w_f32x2 r0 = __builtin_elementwise_fma(tq[2], t_sg, tq[0]);      // (r0, r1)
            const w_f32x2 r1 = __builtin_elementwise_fma(tq[3], t_sg, tq[1]);      // (r2, r3)
            const w_f32x2 d = r0 - r1;                                             // (r0 - r2, r1 - r3) = positions 0 and 3
            av[0][ch] = d[0];
            av[1][ch] = r0[1] + r1[0];
            av[2][ch] = r1[0] - r0[1];
            av[3][ch] = d[1];
        };
        // 8 MFMAs: position j, channels 4 h .. 4 h + 3 of the chunk, both 32-channel blocks alternating
        auto mma = [&](int set, int j, int h) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int blk = 0; blk < NB; ++blk)
                    acc[j][blk] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j][4 * h + s], u[set][blk][s], acc[j][blk], 0, 0, 0);
                // VAR 2 (experiment): keep the two accumulators of a position ALTERNATING (left alone the scheduler issues four
                // MFMAs on one accumulator, then four on the other: back-to-back dependent MFMAs with fillers in between)
                if (VAR == 2) __builtin_amdgcn_sched_barrier(0);
            }
        };
        // interleave pattern of one block: the block's 2 LDS reads behind the first MFMAs, its 2 filter loads behind the next
        // two, `DSW` LDS writes per MFMA where the block stages a raw patch, VALU work everywhere
#define W_SGB_BLOCK(DSW, VM)                                                            \
        do {                                                                                \
            _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                              \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                          \
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                          \
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                          \
            }                                                                               \
            _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) {                              \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                          \
                if (DSW) __builtin_amdgcn_sched_group_barrier(0x200, DSW, 0);               \
                if (VM) __builtin_amdgcn_sched_group_barrier(0x020, VM, 0);                 \
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                          \
            }                                                                               \
        } while (0)
        auto lds_barrier = [&]() {       // __syncthreads() minus its global-memory fence (it would drain every prefetch in flight)
            if constexpr (ABL & 16) return;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        };

        // ---- stream prologue: chunk 0 in LDS, its channels 0..3 transformed, chunk 1 and the first three filter blocks in flight
        raw_issue(0);
        u_issue(0, 0, 0, 0);
        u_issue(1, 0, 1, 0);
        u_issue(2, 0, 2, 0);
        __builtin_amdgcn_sched_barrier(0);
        acc_clear();                                        // 128 moves under the latency of the first loads
        __builtin_amdgcn_sched_barrier(0);
        raw_store(smem);
        lds_barrier();
        W_STAMP(5);
        raw_issue(p.nchunk > 1 ? 1 : 0);
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) { rd(smem, ch); xf(ch); }
        rd(smem, 4);
        W_STAMP(6);
        W_STAMP(1);
        W_SETPRIO_KNOB(3, 0);
        // ---- the stream: 8 blocks of 8 MFMAs per chunk.  First half: channels 0..3 of the chunk multiply while channels 4..7
        // are transformed; second half: channels 4..7 multiply while channels 0..3 of the NEXT chunk are transformed (its raw
        // patch goes registers -> LDS inside block (0, 2), one barrier per chunk after that block).  Every accumulator sees its
        // (chunk, step) products in the same order as in conv_wino_kernel: the results are bit-identical.
        // The LAST chunk is peeled: nothing of a next chunk to stage, transform or request -- the residual's loads take
        // the freed registers and arrive under its MFMAs instead of in front of the epilogue.
        for (int c = 0; c + 1 < p.nchunk; ++c) {
            float* bc = smem + (c & 1) * RAWF;                  // raw patch of chunk c
            float* bn = smem + ((c + 1) & 1) * RAWF;            // chunk c + 1
            const int cn = c + 1;
            const int cnn = c + 2 < p.nchunk ? c + 2 : cn;      // (the last but one iteration re-requests chunk c + 1: branch free, unused)
            __builtin_amdgcn_sched_barrier(0);
            xf(4); rd(bc, 5); mma(0, 0, 0); u_issue(3, c, 3, 0); W_SGB_BLOCK(0, 1);
            __builtin_amdgcn_sched_barrier(0);
            xf(5); rd(bc, 6); mma(1, 1, 0); u_issue(0, c, 0, 1); W_SGB_BLOCK(0, 1);
            __builtin_amdgcn_sched_barrier(0);
            xf(6); rd(bc, 7); raw_store(bn); mma(2, 2, 0); u_issue(1, c, 1, 1); W_SGB_BLOCK(1, 1);
            __builtin_amdgcn_sched_barrier(0);
            if (VAR == 1) { u_issue(2, c, 2, 1); __builtin_amdgcn_sched_barrier(0); }
            lds_barrier();
            __builtin_amdgcn_sched_barrier(0);
            raw_issue(cnn);
            if (VAR == 1) { xf(7); rd(bn, 0); mma(3, 3, 0); W_SGB_BLOCK(0, 1); }
            else { xf(7); rd(bn, 0); mma(3, 3, 0); u_issue(2, c, 2, 1); W_SGB_BLOCK(0, 1); }
            __builtin_amdgcn_sched_barrier(0);
            xf(0); rd(bn, 1); mma(0, 0, 1); u_issue(3, c, 3, 1); W_SGB_BLOCK(0, 1);
            __builtin_amdgcn_sched_barrier(0);
            xf(1); rd(bn, 2); mma(1, 1, 1); u_issue(0, cn, 0, 0); W_SGB_BLOCK(0, 1);
            __builtin_amdgcn_sched_barrier(0);
            xf(2); rd(bn, 3); mma(2, 2, 1); u_issue(1, cn, 1, 0); W_SGB_BLOCK(0, 1);
            __builtin_amdgcn_sched_barrier(0);
            xf(3); rd(bn, 4); mma(3, 3, 1); u_issue(2, cn, 2, 0); W_SGB_BLOCK(0, 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        {
            const int c = p.nchunk - 1;
            float* bc = smem + (c & 1) * RAWF;
            __builtin_amdgcn_sched_barrier(0);
            xf(4); rd(bc, 5); mma(0, 0, 0); u_issue(3, c, 3, 0); W_SGB_BLOCK(0, 1);
            __builtin_amdgcn_sched_barrier(0);
            xf(5); rd(bc, 6); mma(1, 1, 0); u_issue(0, c, 0, 1); W_SGB_BLOCK(0, 1);
            __builtin_amdgcn_sched_barrier(0);
            xf(6); rd(bc, 7); mma(2, 2, 0); u_issue(1, c, 1, 1); W_SGB_BLOCK(0, 1);
            __builtin_amdgcn_sched_barrier(0);
            xf(7); mma(3, 3, 0); u_issue(2, c, 2, 1); W_SGB_BLOCK(0, 1);
            __builtin_amdgcn_sched_barrier(0);
            res_issue();
            mma(0, 0, 1); u_issue(3, c, 3, 1); W_SGB_BLOCK(0, 2);
            __builtin_amdgcn_sched_barrier(0);
            mma(1, 1, 1);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (VAR == 3) {
                // The first stage of the output transform INSIDE the stream.  A VALU instruction between a wave's own MFMAs costs ~2.3
                // matrix-pipe clocks; the same instruction in the epilogue, beside the OTHER workgroup's dense MFMA stream, gets an
                // issue slot once per MFMA (~51 clocks, tools/micro/mfma_neighbor: s_setprio does not change that).  Positions 0 and 1
                // of the wave's row are final after block (1, 1): T0 = (m0 + m1) + m2 and T1 = (m1 - m2) - m3 are formed in place
                // (same operation order as the epilogue used: bit-identical) while positions 2 and 3 still multiply.
                auto tsum = [&](int d, int a, float sg) {      // acc[d] += sg * acc[a]
#pragma unroll
                    for (int blk = 0; blk < NB; ++blk)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[d][blk][r] = sg > 0.f ? acc[d][blk][r] + acc[a][blk][r] : acc[d][blk][r] - acc[a][blk][r];
                };
                tsum(0, 1, 1.f);                                // m0 + m1
                mma(2, 2, 1);
#pragma unroll
                for (int i_ = 0; i_ < 8; ++i_) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                tsum(0, 2, 1.f);                                // (m0 + m1) + m2 = T0
                tsum(1, 2, -1.f);                               // m1 - m2
                mma(3, 3, 1);
#pragma unroll
                for (int i_ = 0; i_ < 8; ++i_) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                tsum(1, 3, -1.f);                               // (m1 - m2) - m3 = T1 (the one stage left outside the stream)
                __builtin_amdgcn_sched_barrier(0);
            } else {
                mma(2, 2, 1);
                __builtin_amdgcn_sched_barrier(0);
                mma(3, 3, 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        W_SETPRIO_KNOB(1, 3);
#undef W_SGB_BLOCK
    } else {
        acc_clear();
        w_f32x4 rr[NE];
        auto raw_issue = [&](int c) {
            if (W_ABLATE(2)) return;
            const unsigned coff = (unsigned)c * 64u;
            const unsigned cinv = ((c * 16 + 4 * myq) < p.C) ? 0u : 0xFFFFFFFFu;
#pragma unroll
            for (int e = 0; e < NE; ++e)
                rr[e] = __builtin_bit_cast(w_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (rbase[e] + coff) | rinv[e] | cinv, 0, 0));
        };
        // filters of (chunk c, position 4 wave + j): u[blk][half] = 4 floats = MFMA steps 4 half .. 4 half + 3
        auto u_issue = [&](w_f32x4 (&u)[NB][2], int c, int j) {
            if (W_ABLATE(1)) return;
            const unsigned base = u_wave + (unsigned)c * 32768u + (unsigned)j * 2048u;
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) {
                const int so = (int)__builtin_amdgcn_readfirstlane(base + (unsigned)blk * u_blk);
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    u[blk][h] = __builtin_bit_cast(w_f32x4, __builtin_amdgcn_raw_buffer_load_b128(ru, u_lane + 1024u * h, so, 0));
            }
        };
        float av[4][8];              // A operands of this chunk: av[j][s] = V[position 4 wave + j][tile][channel 8 kh + s]
        const w_f32x2 t_sg = wave == 1 ? (w_f32x2){1.f, 1.f} : (w_f32x2){-1.f, -1.f};     // row stage: d[ra] + sg * d[rb] (exact)
        auto transform = [&](const float* buf) {
            const float* src = buf + t_src;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float* pa = src + c * PLANE + t_ra * RWP;
                const float* pb = src + c * PLANE + t_rb * RWP;
                const w_f32x2 xa0 = *reinterpret_cast<const w_f32x2*>(pa), xa1 = *reinterpret_cast<const w_f32x2*>(pa + 2);
                const w_f32x2 xb0 = *reinterpret_cast<const w_f32x2*>(pb), xb1 = *reinterpret_cast<const w_f32x2*>(pb + 2);
                const w_f32x2 r0 = __builtin_elementwise_fma(xb0, t_sg, xa0);      // (r0, r1)
                const w_f32x2 r1 = __builtin_elementwise_fma(xb1, t_sg, xa1);      // (r2, r3)
                const w_f32x2 d = r0 - r1;                                         // (r0 - r2, r1 - r3) = positions 0 and 3
                av[0][c] = d[0];
                av[1][c] = r0[1] + r1[0];
                av[2][c] = r1[0] - r0[1];
                av[3][c] = d[1];
            }
        };
        auto mma = [&](int j, const w_f32x4 (&u)[NB][2]) {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
#pragma unroll
                for (int blk = 0; blk < NB; ++blk)
                    acc[j][blk] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j][s], u[blk][s >> 2][s & 3], acc[j][blk], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        };

        // filters one position (16 NB/2 MFMAs) ahead in two alternating register sets.  (One set per position, reloaded three
        // positions ahead, was measured slower: 256 registers with spills, 367 vs 313 us on layer1.)
        w_f32x4 ua[NB][2], ub[NB][2];
        raw_issue(0);
        u_issue(ua, 0, 0);
        W_STAMP(1);
        for (int c = 0; c < p.nchunk; ++c) {
            float* buf = smem + (c & 1) * RAWF;
            // registers -> LDS, transposed to channel planes.  Two buffers: the waves still transforming chunk c - 1 read the
            // other one, and everyone passed the previous barrier after its chunk c - 2 reads: ONE barrier per chunk.
            if (!W_ABLATE(2)) {
#pragma unroll
                for (int e = 0; e < NE; ++e)
#pragma unroll
                    for (int k = 0; k < 4; ++k) buf[rlds[e] + k * PLANE] = rr[e][k];
            }
            __syncthreads();
            if (c == 0) W_STAMP(5);
            // next chunk's raw patch: branch free (the last iteration re-requests its own chunk, results unused -- with a
            // conditional the compiler drains the prefetch of the path that issued none)
            const int cn = c + 1 < p.nchunk ? c + 1 : c;
            raw_issue(cn);
            if (!W_ABLATE(2)) transform(buf);
            if (c == 0) W_STAMP(6);
            __builtin_amdgcn_s_setprio(0);
            // the sched_barriers pin every group of loads in front of the MFMAs it overlaps with (left alone, the scheduler
            // sinks them into the MFMA run to shorten live ranges and the next position then waits for L2)
            __builtin_amdgcn_sched_barrier(0);
            u_issue(ub, c, 1);
            __builtin_amdgcn_sched_barrier(0);
            mma(0, ua);
            __builtin_amdgcn_sched_barrier(0);
            u_issue(ua, c, 2);
            __builtin_amdgcn_sched_barrier(0);
            mma(1, ub);
            __builtin_amdgcn_sched_barrier(0);
            u_issue(ub, c, 3);
            __builtin_amdgcn_sched_barrier(0);
            mma(2, ua);
            __builtin_amdgcn_sched_barrier(0);
            u_issue(ua, cn, 0);
            __builtin_amdgcn_sched_barrier(0);
            mma(3, ub);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(3);
        }
    }
    W_STAMP(2);
    if (W_ABLATE(4)) return;

    // ---------------------------------------------------------------- epilogue: Y = A^T M A, bias, residual, ReLU
    // Wave i holds M[i][0..3]: it forms T[i][b] = sum_j M[i][j] A[j][b] in registers and stages it in LDS,
    // S[i][b][tile][64 couts]; then every thread owns 8 x (pixel, 4 couts), adds the three T rows of its pixel
    // (Y[0][b] = T0 + T1 + T2, Y[1][b] = T1 - T2 - T3, fixed order), bias (folded BatchNorm), residual, ReLU, and
    // global memory sees whole 256-byte pixel rows as 16-byte accesses (out-of-range offsets outside the image).
    if (!STREAM) res_issue();
    // (RES is a template parameter and ReLU a clamp against 0 / -inf: with uniform branches per item the combine below
    // ran one pixel at a time -- three LDS reads, wait, add, branch, store)
    const float relu_lo = p.relu ? 0.f : -__builtin_inff();
    w_f32x4 bias4 = (w_f32x4){0.f, 0.f, 0.f, 0.f};
    if (p.bias) bias4 = *reinterpret_cast<const w_f32x4*>(p.bias + (long long)grp * p.Co + cbk * BN + 4 * cq);
    __syncthreads();                            // every wave is done with the raw buffers (the stage aliases them)
    W_STAMP(4);
    float neg_one = -1.f;
    asm("" : "+v"(neg_one));                    // opaque to the optimiser: it would turn fma(y, -1, x) back into an unpacked v_sub_f32
    {
        float* srow = smem + (wave * 2) * 32 * BN + (lane & 31);
#pragma unroll
        for (int blk = 0; blk < NB; ++blk)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const int tile = (r & 3) + 8 * (r >> 2) + 4 * kh;          // accumulator r + 1 = tile + 1
                if constexpr (STREAM && VAR == 3) {             // T0 / T1 were formed inside the stream
                    srow[tile * BN + blk * 32] = acc[0][blk][r];
                    srow[(tile + 1) * BN + blk * 32] = acc[0][blk][r + 1];
                    srow[(32 + tile) * BN + blk * 32] = acc[1][blk][r];
                    srow[(32 + tile + 1) * BN + blk * 32] = acc[1][blk][r + 1];
                } else {
                    // two accumulators per packed instruction (the registers of a pair are adjacent): beside the other
                    // workgroup's MFMA stream every VALU instruction of the epilogue waits for an issue slot
                    const w_f32x2 m0 = {acc[0][blk][r], acc[0][blk][r + 1]}, m1 = {acc[1][blk][r], acc[1][blk][r + 1]};
                    const w_f32x2 m2 = {acc[2][blk][r], acc[2][blk][r + 1]}, m3 = {acc[3][blk][r], acc[3][blk][r + 1]};
                    const w_f32x2 neg1 = {neg_one, neg_one};      // x - y as fma(y, -1, x): same rounding, and it packs (fsub does not)
                    const w_f32x2 t0 = (m0 + m1) + m2;
                    const w_f32x2 t1 = __builtin_elementwise_fma(m3, neg1, __builtin_elementwise_fma(m2, neg1, m1));
                    srow[tile * BN + blk * 32] = t0[0];
                    srow[(tile + 1) * BN + blk * 32] = t0[1];
                    srow[(32 + tile) * BN + blk * 32] = t1[0];
                    srow[(32 + tile + 1) * BN + blk * 32] = t1[1];
                }
            }
    }
    __syncthreads();
    W_STAMP(7);
    if constexpr (POOL) {
#pragma unroll
        for (int e = 0; e < NIP; ++e) {
            const int tile = e * PSTEP + tid / QP;
            w_f32x4 x[4], y[4], z[4];
#pragma unroll
            for (int ab = 0; ab < 4; ++ab) {                // pixel (a, b) of the tile: 12 LDS reads in flight
                const float* s0 = smem + (ab * 32 + tile) * BN + 4 * cq;
                x[ab] = *reinterpret_cast<const w_f32x4*>(s0);
                y[ab] = *reinterpret_cast<const w_f32x4*>(s0 + 2 * 32 * BN);
                z[ab] = *reinterpret_cast<const w_f32x4*>(s0 + 4 * 32 * BN);
            }
            w_f32x4 m;
#pragma unroll
            for (int ab = 0; ab < 4; ++ab) {
                const float sg = (ab >> 1) ? -1.f : 1.f;
                const w_f32x4 sg4 = {sg, sg, sg, sg};
                const w_f32x4 v = __builtin_elementwise_fma(sg4, z[ab], __builtin_elementwise_fma(sg4, y[ab], x[ab]));
                if (ab == 0) m = v;
                else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) m[k] = fmaxf(m[k], v[k]);
                }
            }
            m = m + bias4;
#pragma unroll
            for (int k = 0; k < 4; ++k) m[k] = fmaxf(m[k], relu_lo);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(w_u32x4, m), rout, goff[e], 0, 0);
        }
    } else
    // four pixels at a time: 12 LDS reads in flight, then the adds and stores
#pragma unroll
    for (int e0 = 0; e0 < NI; e0 += 4) {
        w_f32x4 x[4], y[4], z[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int px = (e0 + i) * PSTEP + tid / QP;
            const int tile = px >> 2, a = (px >> 1) & 1, b = px & 1;
            // a = 0: rows 0, 1, 2 added; a = 1: row 1 minus rows 2, 3
            const float* s0 = smem + ((a * 2 + b) * 32 + tile) * BN + 4 * cq;
            x[i] = *reinterpret_cast<const w_f32x4*>(s0);
            y[i] = *reinterpret_cast<const w_f32x4*>(s0 + 2 * 32 * BN);
            z[i] = *reinterpret_cast<const w_f32x4*>(s0 + 4 * 32 * BN);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = e0 + i;
            const int px = e * PSTEP + tid / QP;
            const float sg = ((px >> 1) & 1) ? -1.f : 1.f;
            const w_f32x4 sg4 = {sg, sg, sg, sg};
            // (+-1) * y is exact: the fused form rounds exactly like the multiply-then-add it replaces, in half the instructions
            w_f32x4 v = __builtin_elementwise_fma(sg4, z[i], __builtin_elementwise_fma(sg4, y[i], x[i]));
            v = v + bias4;
            if (RES) v = v + __builtin_bit_cast(w_f32x4, rv[e]);
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], relu_lo);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(w_u32x4, v), rout, goff[e], 0, 0);
        }
    }
#ifdef SS_TUNING
    if (p.dbg && tid == 0) {
        unsigned long long* d = p.dbg + ((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 10;
        for (int i = 0; i < 8; ++i) d[i] = ts[i];
        d[8] = __builtin_amdgcn_s_memtime();
        d[9] = __builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// EXPERIMENT (tuning build only, ss_debug_set(7, 2); tools/cmp_wino_variants.py, tools/diag_wino_pair.py;
// profiles/r02_wino_pair_experiment.txt): bit-identical to the kernel above, its K loop runs at the matrix pipe's pace
// (9.0k cycles per 128-MFMA chunk vs 9.2k for MFMAs alone), but with one wave per SIMD nothing hides the epilogue and
// the restart of the stream (28k cycles per job against 36k of MFMAs on layer1), and hipcc spills the job state around
// them: 391 / 321 / 268 us on layer1 / 2 / 3 where the kernel above takes 342 / 296 / 249 us.  Not dispatched.
#ifdef SS_TUNING
// PAIR kernel: the same arithmetic, scheduled for ONE workgroup per CU (512 registers per lane: 256 accumulator
// registers + 256 for everything else).  With two workgroups per CU (kernel above) the matrix pipe idles whenever both
// resident waves of a SIMD are outside their MFMA runs at the same time (measured: 64-66 % MFMA busy); here ONE wave
// per SIMD keeps the pipe fed by interleaving everything else into its own MFMA stream:
//   job = TWO consecutive tile blocks (2 x 32 tiles) x 64 output channels: every filter register feeds two MFMAs
//         (half the filter traffic per flop of the kernel above), 4 x 2 x 2 accumulator tiles = 256 registers;
//   the K loop is one stream of 8 blocks of 16 MFMAs per chunk (tile block 0: positions 0..3, tile block 1: positions
//   0..3); while tile block 0 multiplies, the wave transforms tile block 1 of the same chunk, and while tile block 1
//   multiplies, tile block 0 of the NEXT chunk -- a quarter of a transform (2 channels: 8 LDS reads, ~20 VALU) per
//   block, the reads issued one block before their use.  Raw patches go global -> registers two chunks ahead and
//   registers -> LDS one chunk ahead (one barrier per chunk, in the middle of the stream); a position's filters are
//   reloaded for the next chunk right after their last MFMA of this one (3 blocks = 3072 MFMA cycles before the next use);
//   PERSISTENT workgroups: the grid is one workgroup per CU and the stream simply continues into the next job -- the
//   last chunk of a job prefetches / transforms chunk 0 of the job after it -- so only the T stage + combine of the
//   epilogue (own LDS region, it never aliases the raw buffers) interrupt the MFMA stream.
template <int TBH, int TBW, bool RES>
__global__ __launch_bounds__(256, 1) void conv_wino_pair_kernel(WinoP p) {
    static_assert(TBH * TBW == 32, "32 tiles per tile block");
    constexpr int BN = 64;
    constexpr int RH = 2 * TBH + 2, RW = 2 * TBW + 2;
    constexpr int RPIX = RH * RW;
    constexpr int RWP = TBW == 4 ? 12 : 24;
    constexpr int PLANE = RH * RWP + 4;
    constexpr int NE = (RPIX * 4 + 255) / 256;
    constexpr int RAWF = 16 * PLANE;                        // one raw buffer: 16 channel planes of one tile block
    constexpr int STAGEF = 4 * 2 * 32 * BN;                 // T stage of one tile block (64 KB)
    // raw[parity][tile block] during the K loop + the T stage: 120..127 KB of the CU's 160 KB
    // (the stage first: all its addresses within the 64 KB reach of an LDS instruction's offset field from one base)
    __shared__ __attribute__((aligned(16))) float smem_all[STAGEF + 4 * RAWF];
    float* const stage = smem_all;
    float* const smem = smem_all + STAGEF;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // job sequence of this workgroup: every XCD sweeps one contiguous run of jobs (all cout blocks of a pair of tile
    // blocks back to back), its workgroups together
    unsigned job, job_end, job_step;
    {
        const unsigned nwg = gridDim.x, b = blockIdx.x;
        if (nwg >= 16) {
            const unsigned q = p.njobs / 8, r = p.njobs % 8, xcd = b % 8, slot = b / 8;
            const unsigned start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
            job_step = nwg / 8 + (xcd < nwg % 8 ? 1u : 0u);
            job = start + slot;
            job_end = start + q + (xcd < r ? 1u : 0u);
        } else {
            job = b; job_end = p.njobs; job_step = nwg;
        }
    }
    if (job >= job_end) return;
    const int grp = blockIdx.z;
    const __amdgpu_buffer_rsrc_t rin = w_rsrc(p.in + (long long)grp * p.in_gs, p.in_bytes);
    const __amdgpu_buffer_rsrc_t ru = w_rsrc(p.U + (long long)grp * p.u_gs, p.u_bytes);
    float* __restrict__ out = p.out + (long long)grp * p.out_gs;
    const __amdgpu_buffer_rsrc_t rout = w_rsrc(out, p.out_bytes);
    const __amdgpu_buffer_rsrc_t rres = w_rsrc(RES ? p.res + (long long)grp * p.out_gs : out, p.out_bytes);

    // invalid raw items: an offset that stays outside the buffer when the chunk offset (< 64 KB) is added -- the
    // descriptor's bounds check returns zeros (host: in_bytes <= RAW_INVALID)
    constexpr unsigned RAW_INVALID = 0xFFFF0000u;
    struct Job {
        unsigned cbk, u_wave;
        unsigned img[2];
        int oy0[2], ox0[2];
        unsigned rbase[2][NE];             // byte offset of (pixel, channel quad), or RAW_INVALID
    };
    int rlds[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int item = tid + 256 * e;
        const int pix = item >> 2, q = item & 3;
        const int ry = pix / RW, rx = pix - ry * RW;
        // items past the patch (RPIX * 4 is not a multiple of 256) land in the 4-dword pad behind their planes: no branch
        rlds[e] = (4 * q) * PLANE + (item < RPIX * 4 ? ry * RWP + rx : RH * RWP);
    }
    auto setup = [&](Job& j, unsigned lin) {
        const unsigned pm = ss_div32(lin, p.divNcb);
        j.cbk = lin - pm * p.ncb;
        j.u_wave = (j.cbk * 2) * (unsigned)p.nchunk * 32768u + (unsigned)wave * 8192u;
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
            const unsigned mb = 2 * pm + tb;
            const unsigned t1 = ss_div32(mb, p.divBx);
            const int bx = (int)(mb - t1 * p.nbx);
            const unsigned img = ss_div32(t1, p.divBy);
            const int by = (int)(t1 - img * p.nby);
            const bool valid = mb < p.nmb;                  // an odd number of tile blocks: the last job's second one is empty
            j.img[tb] = valid ? img : 0u;
            j.oy0[tb] = valid ? by * (2 * TBH) : (1 << 28);
            j.ox0[tb] = bx * (2 * TBW);
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const int item = tid + 256 * e;
                const int pix = item >> 2, q = item & 3;
                const int ry = pix / RW, rx = pix - ry * RW;
                const int iy = j.oy0[tb] - 1 + ry, ix = j.ox0[tb] - 1 + rx;
                const bool ok = item < RPIX * 4 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                j.rbase[tb][e] = ok ? ((((unsigned)j.img[tb] * p.H + iy) * p.W + ix) * (unsigned)p.C + 4u * q) * 4u : RAW_INVALID;
            }
        }
    };
    Job cur, nxt;
    setup(cur, job);
    const int myq = tid & 3;

    const int kh = lane >> 5;
    const int m_tile = lane & 31;
    const int m_ty = m_tile / TBW, m_tx = m_tile - m_ty * TBW;
    const int t_src = (8 * kh) * PLANE + (2 * m_ty) * RWP + 2 * m_tx;
    const int t_ra = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
    const int t_rb = wave == 3 ? 3 : (wave == 2 ? 1 : 2);
    const w_f32x2 t_sg = wave == 1 ? (w_f32x2){1.f, 1.f} : (w_f32x2){-1.f, -1.f};
    const unsigned u_lane = (unsigned)lane * 16u;
    const unsigned u_blk = (unsigned)p.nchunk * 32768u;

    w_f32x16 acc[4][2][2];       // [position of the wave's row][tile block][32-channel block]
    float av[2][4][8];           // A operands: [tile block][position][channel 8 kh + s]
    w_f32x4 U[4][2][2];          // filters of the chunk: [position][32-channel block][half] (4 MFMA steps each)
    w_f32x4 rr[2][NE];           // raw patch in flight
    w_f32x2 tq[2][4];            // LDS reads in flight: 2 channels x (row a cols 01, row a cols 23, row b cols 01, row b cols 23)

    auto raw_issue = [&](int cc, bool from_next) {
        const unsigned coff = (unsigned)cc * 64u;
        const unsigned cinv = ((cc * 16 + 4 * myq) < p.C) ? 0u : 0xFFFFFFFFu;
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const unsigned off = ((from_next ? nxt.rbase[tb][e] : cur.rbase[tb][e]) + coff) | cinv;
                rr[tb][e] = __builtin_bit_cast(w_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, off, 0, 0));
            }
    };
    auto raw_store = [&](float* dst) {                       // dst: the parity's two tile-block buffers
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
            for (int e = 0; e < NE; ++e)
#pragma unroll
                for (int k = 0; k < 4; ++k) dst[tb * RAWF + rlds[e] + k * PLANE] = rr[tb][e][k];
    };
    auto u_issue = [&](int j, unsigned u_wave, int cc) {
        if (W_ABLATE(1)) return;
        const unsigned base = u_wave + (unsigned)cc * 32768u + (unsigned)j * 2048u;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const int so = (int)__builtin_amdgcn_readfirstlane(base + (unsigned)cb * u_blk);
#pragma unroll
            for (int h = 0; h < 2; ++h)
                U[j][cb][h] = __builtin_bit_cast(w_f32x4, __builtin_amdgcn_raw_buffer_load_b128(ru, u_lane + 1024u * h, so, 0));
        }
    };
    // a quarter of a tile block's input transform: channels 2 pair, 2 pair + 1 of this lane's 8
    auto rd = [&](const float* buf, int pair) {
        if (W_ABLATE(8)) return;
        const float* src = buf + t_src + (2 * pair) * PLANE;
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            const float* pa = src + ch * PLANE + t_ra * RWP;
            const float* pb = src + ch * PLANE + t_rb * RWP;
            tq[ch][0] = *reinterpret_cast<const w_f32x2*>(pa);
            tq[ch][1] = *reinterpret_cast<const w_f32x2*>(pa + 2);
            tq[ch][2] = *reinterpret_cast<const w_f32x2*>(pb);
            tq[ch][3] = *reinterpret_cast<const w_f32x2*>(pb + 2);
        }
    };
    auto xf = [&](int tb, int pair, bool pin = false) {
        if (W_ABLATE(8)) return;
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            const int c = 2 * pair + ch;
            const w_f32x2 r0 = __builtin_elementwise_fma(tq[ch][2], t_sg, tq[ch][0]);
            const w_f32x2 r1 = __builtin_elementwise_fma(tq[ch][3], t_sg, tq[ch][1]);
            const w_f32x2 d = r0 - r1;
            av[tb][0][c] = d[0];
            av[tb][1][c] = r0[1] + r1[0];
            av[tb][2][c] = r1[0] - r0[1];
            av[tb][3][c] = d[1];
            // pin the results here: in the last chunk of a job they are used only after the epilogue, and the compiler
            // otherwise SINKS the transform there -- spilling the 64 LDS-read registers it needs to scratch on the way
            if (pin) asm volatile("" : "+v"(av[tb][0][c]), "+v"(av[tb][1][c]), "+v"(av[tb][2][c]), "+v"(av[tb][3][c]));
        }
    };
    const w_f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // 16 MFMAs: position j of tile block tb, both channel blocks alternating (independent accumulators back to back).
    // `first`: chunk 0 of a job starts its accumulators from zero (no 256-register clear per job).
    auto mma = [&](int tb, int j, bool first) {
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
                acc[j][tb][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tb][j][s], U[j][cb][s >> 2][s & 3],
                                                                     (first && s == 0) ? zero16 : acc[j][tb][cb], 0, 0, 0);
    };
    // interleave pattern of one block of 16 MFMAs: the block's LDS reads (4 ds_read2) behind the first MFMAs, then per
    // MFMA up to `DSW` LDS writes / `VM` buffer loads, and VALU work everywhere
#define W_SGB_BLOCK(DSW, VM)                                                            \
    do {                                                                                \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                              \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                          \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                          \
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                          \
        }                                                                               \
        _Pragma("unroll") for (int i_ = 0; i_ < 12; ++i_) {                             \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                          \
            if (DSW) __builtin_amdgcn_sched_group_barrier(0x200, DSW, 0);               \
            if (VM) __builtin_amdgcn_sched_group_barrier(0x020, VM, 0);                 \
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                          \
        }                                                                               \
    } while (0)

    // LDS-only barrier: __syncthreads() also fences global memory (s_waitcnt vmcnt(0)): every prefetch in flight --
    // and, after an epilogue, every store -- would be waited for at every barrier
    auto lds_barrier = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    };
#ifdef SS_TUNING
    unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
    const unsigned long long tstart = tprev;
#define P_STAMP(i) do { if (p.dbg) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tacc[i] += t_ - tprev; tprev = t_; } } while (0)
#else
#define P_STAMP(i) do { } while (0)
#endif

    // epilogue addressing.  thread -> (pixel, channel quad): pixel 16 e + (tid >> 4) -> tile 4 e + wave (wave-uniform:
    // scalar address arithmetic), row / column inside the 2x2 tile from tid bits 5 / 4, channels 4 (tid & 15) ..
    constexpr int NI = 8;
    const int cq = tid & 15;
    const int pa = (tid >> 5) & 1, pb = (tid >> 4) & 1;
    const unsigned thr_off = ((unsigned)(pa * p.W + pb) * (unsigned)p.out_cs + 4u * cq) * 4u;
    const float* srd = stage + ((pa * 2 + pb) * 32 + wave) * BN + 4 * cq;              // + 4 e tiles
    const float sg = pa ? -1.f : 1.f;                       // row 0 of the tile: T0 + T1 + T2, row 1: T1 - T2 - T3
    const float relu_lo = p.relu ? 0.f : -__builtin_inff();

    int par = 0;
    raw_issue(0, false);                                    // chunk 0 of the first job
    for (;;) {
        const bool has_next = job + job_step < job_end;
        setup(nxt, has_next ? job + job_step : job);      // the last job names itself: prefetches requested, never used
        // ---- job start: chunk 0 (requested before the previous epilogue) -> LDS, transformed for tile block 0; filters
        // of chunk 0 and the raw patch of chunk 1 requested.  ~2k cycles without MFMAs per job.
        raw_store(smem + (par * 2) * RAWF);
#pragma unroll
        for (int j = 0; j < 4; ++j) u_issue(j, cur.u_wave, 0);
        lds_barrier();
        raw_issue(1, false);
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) { rd(smem + (par * 2) * RAWF, pr); xf(0, pr); }
        rd(smem + (par * 2 + 1) * RAWF, 0);
        P_STAMP(0);

        unsigned goff[2][NI];
        w_u32x4 rv[2][NI];
        auto res_issue = [&](int tb) {
#pragma unroll
            for (int e = 0; e < NI; ++e) {
                const int tile = 4 * e + wave;
                const int ty = tile / TBW, tx = tile - ty * TBW;
                const int oy = cur.oy0[tb] + 2 * ty, ox = cur.ox0[tb] + 2 * tx;        // scalars
                const unsigned sbase = ((((unsigned)cur.img[tb] * p.H + oy) * p.W + ox) * (unsigned)p.out_cs + cur.cbk * BN) * 4u;
                const bool ok = oy + pa < p.H && ox + pb < p.W;
                goff[tb][e] = ok ? sbase + thr_off : 0xFFFFFFFFu;
            }
            if (RES) {
#pragma unroll
                for (int e = 0; e < NI; ++e) rv[tb][e] = __builtin_amdgcn_raw_buffer_load_b128(rres, goff[tb][e], 0, 0);
            }
        };
        // one chunk = 8 blocks of 16 MFMAs.  kind: 0 first chunk of the job (accumulators start from zero), 1 middle,
        // 2 last: nothing of the next chunk to stage or transform (the next job restarts the stream after the epilogue:
        // carrying the transformed operands, LDS reads and filters of its chunk 0 across the epilogue was measured -- 60+
        // registers too many, scratch traffic in the epilogue); the residual is requested instead.
        auto chunk = [&](int c, auto kind_c) __attribute__((always_inline)) {
            constexpr int kind = decltype(kind_c)::value;
            constexpr bool first = kind == 0, last = kind == 2;
            float* bc = smem + (par * 2) * RAWF;           // chunk c:     [tile block] raw buffers
            float* bn = smem + ((par ^ 1) * 2) * RAWF;     // chunk c + 1
            // tile block 0 multiplies, tile block 1 of this chunk is transformed
            xf(1, 0); rd(bc + RAWF, 1); mma(0, 0, first); W_SGB_BLOCK(0, 0);
            __builtin_amdgcn_sched_barrier(0);
            xf(1, 1); rd(bc + RAWF, 2); mma(0, 1, first); W_SGB_BLOCK(0, 0);
            __builtin_amdgcn_sched_barrier(0);
            // chunk c + 1: registers -> LDS under the MFMAs (its buffers were last read one chunk ago, before the previous barrier)
            xf(1, 2); rd(bc + RAWF, 3);
            if (!last && !W_ABLATE(2)) raw_store(bn);
            mma(0, 2, first); W_SGB_BLOCK(1, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (!last && !W_ABLATE(16)) lds_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // chunk c + 2 requested (the next job's chunk 0 from the last but one chunk: it waits in registers)
            if (!last && !W_ABLATE(2)) {
                const bool wrap = c + 2 >= p.nchunk;
                raw_issue(wrap ? 0 : c + 2, wrap);
            }
            xf(1, 3); if (!last) rd(bn, 0); mma(0, 3, first); W_SGB_BLOCK(0, 1);
            __builtin_amdgcn_sched_barrier(0);
            // tile block 1 multiplies, tile block 0 of the next chunk is transformed; a position's filters are reloaded
            // for the next chunk behind their last MFMA of this one
            if (!last) { xf(0, 0); rd(bn, 1); }
            mma(1, 0, first); if (!last) u_issue(0, cur.u_wave, c + 1); W_SGB_BLOCK(0, 1);
            __builtin_amdgcn_sched_barrier(0);
            if (last) res_issue(0);                         // into the registers of the (dead) tile block 0 operands
            if (!last) { xf(0, 1); rd(bn, 2); }
            mma(1, 1, first); if (!last) u_issue(1, cur.u_wave, c + 1); W_SGB_BLOCK(0, 1);
            __builtin_amdgcn_sched_barrier(0);
            if (!last) { xf(0, 2); rd(bn, 3); }
            mma(1, 2, first); if (!last) u_issue(2, cur.u_wave, c + 1); W_SGB_BLOCK(0, 1);
            __builtin_amdgcn_sched_barrier(0);
            if (!last) { xf(0, 3); rd(bn + RAWF, 0); }
            mma(1, 3, first); if (!last) u_issue(3, cur.u_wave, c + 1); W_SGB_BLOCK(0, 1);
            __builtin_amdgcn_sched_barrier(0);
            par ^= 1;
        };
        chunk(0, std::integral_constant<int, 0>());
        P_STAMP(1);
        for (int c = 1; c + 1 < p.nchunk; ++c) chunk(c, std::integral_constant<int, 1>());
        chunk(p.nchunk - 1, std::integral_constant<int, 2>());
        P_STAMP(2);

        // ------------------------------------------------------------ epilogue (as the kernel above), one tile block at a time
        w_f32x4 bias4 = (w_f32x4){0.f, 0.f, 0.f, 0.f};
        if (p.bias) bias4 = *reinterpret_cast<const w_f32x4*>(p.bias + (long long)grp * p.Co + cur.cbk * BN + 4 * cq);
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
            if (tb == 1) lds_barrier();                     // the combine of tile block 0 is done with the stage
            {
                float* srow = stage + (wave * 2) * 32 * BN + (lane & 31);
#pragma unroll
                for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int tile = (r & 3) + 8 * (r >> 2) + 4 * kh;
                        const float m0 = acc[0][tb][blk][r], m1 = acc[1][tb][blk][r], m2 = acc[2][tb][blk][r], m3 = acc[3][tb][blk][r];
                        srow[tile * BN + blk * 32] = (m0 + m1) + m2;
                        srow[(32 + tile) * BN + blk * 32] = (m1 - m2) - m3;
                        // four accumulator rows at a time: left alone the scheduler reads all 128 accumulators of the
                        // tile block into registers first (scratch)
                        ;
                    }
            }
            // residual of both tile blocks requested here: the accumulators of this tile block are dead, their registers
            // take the loads (requested inside the last chunk they were spilled to scratch one by one)
            if (tb == 0) res_issue(1);
            lds_barrier();
            P_STAMP(3);
#pragma unroll
            for (int e = 0; e < NI; ++e) {
                const float* s0 = srd + 4 * e * BN;
                const w_f32x4 x = *reinterpret_cast<const w_f32x4*>(s0);
                const w_f32x4 y = *reinterpret_cast<const w_f32x4*>(s0 + 2 * 32 * BN);
                const w_f32x4 z = *reinterpret_cast<const w_f32x4*>(s0 + 4 * 32 * BN);
                w_f32x4 v = (x + sg * y) + sg * z;
                v = v + bias4;
                if (RES) v = v + __builtin_bit_cast(w_f32x4, rv[tb][e]);
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], relu_lo);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(w_u32x4, v), rout, goff[tb][e], 0, 0);
                ;
            }
            P_STAMP(4);
        }
        if (!has_next) break;
        job += job_step;
        cur = nxt;
    }
#undef W_SGB_BLOCK
#undef P_STAMP
#ifdef SS_TUNING
    if (p.dbg && tid == 0) {
        unsigned long long* d = p.dbg + ((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 10;
        for (int i = 0; i < 6; ++i) d[i] = tacc[i];
        d[6] = tstart;
        d[7] = __builtin_amdgcn_s_memtime();
    }
#endif
}
#endif  // SS_TUNING

// ------------------------------------------------------------------------------------------------
// Filter transform + packing: U = G g G^T in fp64, rounded once to fp32, stored in the B-operand register layout of the
// kernel above (v_mfma_f32_32x32x2_f32: B[k = lane >> 5][j = lane & 31]):
//   U[cout/32][chunk][pos][half][lane][e] = (G g G^T)[pos] of (cout = 32 cb + (lane & 31), cin = 16 chunk + 8 (lane >> 5) + 4 half + e),
// zero for cin >= C.   wgt: [cout][1][3][3][cin] (BN folded).
__global__ void wino_pack_kernel(const float* __restrict__ wgt, float* __restrict__ U, int cout, int cin, int nchunk,
                                 long long w_gs, long long u_gs) {
    const long long per = (long long)(cout / 32) * nchunk * 16 * 2 * 64;   // float4 slots per group
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= per) return;
    const int grp = blockIdx.y;
    const int lane = (int)(idx & 63);
    const int half = (int)((idx >> 6) & 1);
    const int pos = (int)((idx >> 7) & 15);
    const long long cc = idx >> 11;
    const int chunk = (int)(cc % nchunk);
    const int cb = (int)(cc / nchunk);
    const int co = cb * 32 + (lane & 31);
    const int i = pos >> 2, j = pos & 3;
    const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
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

// Filter slices for the bf16-pipe variant: U = G g G^T in fp64, rounded once to fp32 (the SAME value the fp32 kernel
// multiplies with), then written as three bf16 slices u1 + u2 + u3 = U exactly, in the B-operand register layout of
// v_mfma_f32_32x32x16_bf16 (B[k = 8 (lane >> 5) + e][n = lane & 31]):
//   U3[cout/32][chunk][pos][slice][lane][e] = slice of (G g G^T)[pos] of (cout = 32 cb + (lane & 31), cin = 16 chunk + 8 (lane >> 5) + e)
__global__ void wino_pack3_kernel(const float* __restrict__ wgt, unsigned* __restrict__ U3, int cout, int cin, int nchunk,
                                  long long w_gs, long long u_gs) {
    const long long per = (long long)(cout / 32) * nchunk * 16 * 64;      // (cb, chunk, pos, lane) items; 3 x 16 bytes each
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= per) return;
    const int grp = blockIdx.y;
    const int lane = (int)(idx & 63);
    const int pos = (int)((idx >> 6) & 15);
    const long long cc = idx >> 10;
    const int chunk = (int)(cc % nchunk);
    const int cb = (int)(cc / nchunk);
    const int co = cb * 32 + (lane & 31);
    const int i = pos >> 2, j = pos & 3;
    const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
    unsigned short sl[3][8];
    for (int e = 0; e < 8; ++e) {
        const int ci = chunk * 16 + 8 * (lane >> 5) + e;
        double a = 0.0;
        if (ci < cin) {
            const float* g = wgt + (long long)grp * w_gs + (long long)co * 9 * cin + ci;
            for (int x = 0; x < 3; ++x)
                for (int y = 0; y < 3; ++y) a += G[i][x] * (double)g[(x * 3 + y) * cin] * G[j][y];
        }
        const float u = (float)a;
        const __bf16 u1 = (__bf16)u;
        const float r1 = u - (float)u1;
        const __bf16 u2 = (__bf16)r1;
        const __bf16 u3 = (__bf16)(r1 - (float)u2);
        sl[0][e] = __builtin_bit_cast(unsigned short, u1);
        sl[1][e] = __builtin_bit_cast(unsigned short, u2);
        sl[2][e] = __builtin_bit_cast(unsigned short, u3);
    }
    unsigned* dst = U3 + (long long)grp * u_gs + ((cc * 16 + pos) * 3) * 256 + lane * 4;       // dwords
    for (int s3 = 0; s3 < 3; ++s3)
        for (int q = 0; q < 4; ++q) dst[s3 * 256 + q] = (unsigned)sl[s3][2 * q] | ((unsigned)sl[s3][2 * q + 1] << 16);
}

extern "C" long long ss_wino_packed3_floats(int cout, int cin) {
    if (cout <= 0 || cin <= 0 || (cout & 31)) return 0;
    return (long long)(cout / 32) * ss_cdiv(cin, 16) * 16 * 3 * 64 * 4;
}

extern "C" int ss_wino_pack3(const float* wgt, float* packed, int cout, int cin, int groups, void* stream) {
    if (!wgt || !packed || cout <= 0 || cin <= 0 || (cout & 31) || (cin & 3) || groups <= 0) return SS_ERR_ARG;
    const int nchunk = ss_cdiv(cin, 16);
    const long long per = (long long)(cout / 32) * nchunk * 16 * 64;
    hipLaunchKernelGGL(wino_pack3_kernel, dim3(ss_cdiv(per, 256), groups), dim3(256), 0, (hipStream_t)stream, wgt,
                       reinterpret_cast<unsigned*>(packed), cout, cin, nchunk, (long long)cout * 9 * cin, per * 12);
    return ss_launch_status();
}

extern "C" long long ss_wino_packed_floats(int cout, int cin) {
    if (cout <= 0 || cin <= 0 || (cout & 31)) return 0;
    return (long long)(cout / 32) * ss_cdiv(cin, 16) * 16 * 2 * 64 * 4;
}

extern "C" int ss_wino_pack(const float* wgt, float* packed, int cout, int cin, int groups, void* stream) {
    if (!wgt || !packed || cout <= 0 || cin <= 0 || (cout & 31) || (cin & 3) || groups <= 0) return SS_ERR_ARG;
    const int nchunk = ss_cdiv(cin, 16);
    const long long per = (long long)(cout / 32) * nchunk * 16 * 2 * 64;
    hipLaunchKernelGGL(wino_pack_kernel, dim3(ss_cdiv(per, 256), groups), dim3(256), 0, (hipStream_t)stream, wgt, packed,
                       cout, cin, nchunk, (long long)cout * 9 * cin, per * 4);
    return ss_launch_status();
}

#ifdef SS_TUNING
int g_wino_ablate = 0;                   // ss_debug_set key 6
int g_wino_nb1_max_cin = 0;              // ss_debug_set key 5
int g_wino_variant = 0;                  // ss_debug_set key 7: 0 stream kernel (dispatched), 1 phase-alternating kernel, 2 pair kernel
int g_wino_knob[4] = {0, 0, 0, 0};       // ss_debug_set keys 16..19
int g_wino_lds_pad = 0;                  // ss_debug_set key 20: extra dynamic LDS bytes per workgroup (1 workgroup per CU from 32 KB up)
#else
constexpr int g_wino_nb1_max_cin = 0;
#endif

// tile-block shape: the one that wastes fewer tile slots on this map (8x4 or 4x8 blocks of 2x2 tiles)
static void wino_blocks(int h, int w, int& tbh, int& tbw, double& eff) {
    const int th = (h + 1) / 2, tw = (w + 1) / 2;
    const double e84 = (double)(h * w) / (4.0 * ss_cdiv(th, 8) * 8 * ss_cdiv(tw, 4) * 4);
    const double e48 = (double)(h * w) / (4.0 * ss_cdiv(th, 4) * 4 * ss_cdiv(tw, 8) * 8);
    if (e84 >= e48) { tbh = 8; tbw = 4; eff = e84; } else { tbh = 4; tbw = 8; eff = e48; }
}

// The engine's dispatch rule (also used by bench.py to count executed flops): Winograd where the geometry fits and the
// launch fills the chip; everything else stays on the implicit-GEMM kernel of conv.hip.
extern "C" int ss_conv_uses_winograd(int kt, int kh, int kw, int stride, int cin, int cout, int ho, int wo, int images) {
    if (kt != 1 || kh != 3 || kw != 3 || stride != 1) return 0;
    if (cin < 32 || (cin & 3) || cout < 64 || (cout & 63)) return 0;
    int tbh, tbw;
    double eff;
    wino_blocks(ho, wo, tbh, tbw, eff);
    const long long wgs = (long long)images * ss_cdiv((ho + 1) / 2, tbh) * ss_cdiv((wo + 1) / 2, tbw) * (cout / 64);
    // >= 96 workgroups: below that the launch is one workgroup deep and its K loop (~2.3 us per chunk) is slower than the
    // split-K implicit GEMM spread over the whole chip (tools/wino_threshold.py: 256->256 at 23x30 x2: 41 vs 30 us).
    // Maps that fill only 60-70 % of their tile slots (the regressors' 11x15: 64 %) still win once the launch is two workgroups
    // deep (tools/wino_small_maps.py, 62 images 128->128: 33 vs 44 us; at 32 images equal; 5x7 maps, 27 %, lose)
    return eff >= 0.70 ? wgs >= 96 : (eff >= 0.60 && wgs >= 192);
}

static int wino_launch(const float* in, const float* packed, const float* bias, const float* res, float* out,
                       int n, int h, int w, int cin, int cout, int relu, int out_cs, int groups,
                       long long in_gs, long long u_gs, long long out_gs, void* stream, bool sliced, bool pool = false) {
    if (!in || !packed || !out || n <= 0 || h <= 0 || w <= 0 || cin <= 0 || (cin & 3) || cout <= 0 || (cout & 63) ||
        groups <= 0 || out_cs < cout || (pool && (res || sliced || h < 2 || w < 2)))
        return SS_ERR_ARG;
    const long long in_elems = (long long)n * h * w * cin;
    const long long out_elems = pool ? (long long)n * (h / 2) * (w / 2) * out_cs : (long long)n * h * w * out_cs;
    const long long u_floats = sliced ? ss_wino_packed3_floats(cout, cin) : ss_wino_packed_floats(cout, cin);
    if (in_elems * 4 >= (1ll << 32) || out_elems * 4 >= (1ll << 32) || u_floats * 4 >= (1ll << 31)) return SS_ERR_UNSUPPORTED;
    WinoP p;
    p.in = in; p.U = packed; p.bias = bias; p.res = res; p.out = out;
    p.N = n; p.H = h; p.W = w; p.C = cin; p.Co = cout;
    p.nchunk = ss_cdiv(cin, 16);
    p.relu = relu; p.out_cs = out_cs;
    int tbh, tbw;
    double eff;
    wino_blocks(h, w, tbh, tbw, eff);
    p.nbx = (unsigned)ss_cdiv((w + 1) / 2, tbw);
    p.nby = (unsigned)ss_cdiv((h + 1) / 2, tbh);
    p.divBx = ss_div32_make(p.nbx);
    p.divBy = ss_div32_make(p.nby);
    const int nb = (!sliced && cin <= g_wino_nb1_max_cin) ? 1 : 2;         // 2 (64-channel blocks) in the product build
    p.ncb = (unsigned)(cout / (32 * nb));
    p.divNcb = ss_div32_make(p.ncb);
    p.in_gs = in_gs; p.u_gs = u_gs; p.out_gs = out_gs;
    p.in_bytes = (unsigned)(in_elems * 4);
    p.out_bytes = (unsigned)(out_elems * 4);
    p.u_bytes = (unsigned)(u_floats * 4);
#ifdef SS_TUNING
    p.dbg = ss_tuning_dbg;
    p.ablate = g_wino_ablate;
    for (int i = 0; i < 4; ++i) p.knob[i] = g_wino_knob[i];
#endif
    const long long wgs = (long long)n * p.nbx * p.nby * p.ncb;
    if (wgs >= (1ll << 31)) return SS_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
#ifdef SS_TUNING
    // pair kernel (one persistent workgroup per CU, two tile blocks per job): needs >= 2 chunks for its prefetch distance
    if (!sliced && !pool && g_wino_variant == 2 && nb == 2 && p.nchunk >= 2 && p.nchunk <= 512 && in_elems * 4 <= 0xFFFF0000ll) {
        p.nmb = (unsigned)((long long)n * p.nbx * p.nby);
        p.njobs = (unsigned)(((long long)p.nmb + 1) / 2 * p.ncb);
        long long cap = (long long)(256 / groups) & ~7ll;
        if (cap < 8) cap = 8;
        dim3 gp((unsigned)(p.njobs < cap ? p.njobs : cap), 1, groups);
        if (tbh == 8) {
            if (res) hipLaunchKernelGGL((conv_wino_pair_kernel<8, 4, true>), gp, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((conv_wino_pair_kernel<8, 4, false>), gp, dim3(256), 0, st, p);
        } else {
            if (res) hipLaunchKernelGGL((conv_wino_pair_kernel<4, 8, true>), gp, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((conv_wino_pair_kernel<4, 8, false>), gp, dim3(256), 0, st, p);
        }
        return ss_launch_status();
    }
#endif
    dim3 g((unsigned)wgs, 1, groups);
    if (pool) {             // conv + bias + ReLU + MaxPool2d(2, 2) in one kernel (VAR = 8)
        if (tbh == 8) hipLaunchKernelGGL((conv_wino_kernel<8, 4, 2, false, true, false, 8>), g, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((conv_wino_kernel<4, 8, 2, false, true, false, 8>), g, dim3(256), 0, st, p);
        return ss_launch_status();
    }
    if (sliced) {           // fp32 products from three bf16 slices per operand on the bf16 matrix pipe (opt-in entry point)
        if (tbh == 8) {
            if (res) hipLaunchKernelGGL((conv_wino_kernel<8, 4, 2, true, false, true>), g, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((conv_wino_kernel<8, 4, 2, false, false, true>), g, dim3(256), 0, st, p);
        } else {
            if (res) hipLaunchKernelGGL((conv_wino_kernel<4, 8, 2, true, false, true>), g, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((conv_wino_kernel<4, 8, 2, false, false, true>), g, dim3(256), 0, st, p);
        }
        return ss_launch_status();
    }
#ifdef SS_TUNING      // 32-channel blocks / three workgroups per CU: measured slower on every layer (tools/diag_wino.py); tools build only
    if (nb == 1) {
        if (tbh == 8) {
            if (res) hipLaunchKernelGGL((conv_wino_kernel<8, 4, 1, true, false>), g, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((conv_wino_kernel<8, 4, 1, false, false>), g, dim3(256), 0, st, p);
        } else {
            if (res) hipLaunchKernelGGL((conv_wino_kernel<4, 8, 1, true, false>), g, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((conv_wino_kernel<4, 8, 1, false, false>), g, dim3(256), 0, st, p);
        }
        return ss_launch_status();
    }
#endif
#ifdef SS_TUNING      // the phase-alternating kernel: comparison runs only (tools/cmp_wino_variants.py, tools/diag_wino.py)
    if (g_wino_variant == 1) {
        if (tbh == 8) {
            if (res) hipLaunchKernelGGL((conv_wino_kernel<8, 4, 2, true, false>), g, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((conv_wino_kernel<8, 4, 2, false, false>), g, dim3(256), 0, st, p);
        } else {
            if (res) hipLaunchKernelGGL((conv_wino_kernel<4, 8, 2, true, false>), g, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((conv_wino_kernel<4, 8, 2, false, false>), g, dim3(256), 0, st, p);
        }
        return ss_launch_status();
    }
#endif
#ifdef SS_TUNING
    const unsigned dyn = (unsigned)g_wino_lds_pad;
    if (g_wino_variant >= 100 && tbh == 8 && res) {       // stream ablations (timing only): variant = 100 + mask
        switch (g_wino_variant - 100) {
#define W_ABL_CASE(m) case m: hipLaunchKernelGGL((conv_wino_kernel<8, 4, 2, true, true, false, 16 + m>), g, dim3(256), dyn, st, p); return ss_launch_status();
            W_ABL_CASE(1) W_ABL_CASE(2) W_ABL_CASE(4) W_ABL_CASE(8) W_ABL_CASE(16) W_ABL_CASE(3) W_ABL_CASE(12) W_ABL_CASE(15) W_ABL_CASE(31)
#undef W_ABL_CASE
            default: break;
        }
    }
    if (g_wino_variant == 6) {      // prologue without index arithmetic (VAR = 4, timing only)
        if (tbh == 8) {
            if (res) hipLaunchKernelGGL((conv_wino_kernel<8, 4, 2, true, true, false, 4>), g, dim3(256), dyn, st, p);
            else hipLaunchKernelGGL((conv_wino_kernel<8, 4, 2, false, true, false, 4>), g, dim3(256), dyn, st, p);
        } else {
            if (res) hipLaunchKernelGGL((conv_wino_kernel<4, 8, 2, true, true, false, 4>), g, dim3(256), dyn, st, p);
            else hipLaunchKernelGGL((conv_wino_kernel<4, 8, 2, false, true, false, 4>), g, dim3(256), dyn, st, p);
        }
        return ss_launch_status();
    }
    if (g_wino_variant == 5) {      // output transform stage 1 inside the stream (VAR = 3)
        if (tbh == 8) {
            if (res) hipLaunchKernelGGL((conv_wino_kernel<8, 4, 2, true, true, false, 3>), g, dim3(256), dyn, st, p);
            else hipLaunchKernelGGL((conv_wino_kernel<8, 4, 2, false, true, false, 3>), g, dim3(256), dyn, st, p);
        } else {
            if (res) hipLaunchKernelGGL((conv_wino_kernel<4, 8, 2, true, true, false, 3>), g, dim3(256), dyn, st, p);
            else hipLaunchKernelGGL((conv_wino_kernel<4, 8, 2, false, true, false, 3>), g, dim3(256), dyn, st, p);
        }
        return ss_launch_status();
    }
    if (g_wino_variant == 4) {      // accumulator-alternation experiment (VAR = 2)
        if (tbh == 8) {
            if (res) hipLaunchKernelGGL((conv_wino_kernel<8, 4, 2, true, true, false, 2>), g, dim3(256), dyn, st, p);
            else hipLaunchKernelGGL((conv_wino_kernel<8, 4, 2, false, true, false, 2>), g, dim3(256), dyn, st, p);
        } else {
            if (res) hipLaunchKernelGGL((conv_wino_kernel<4, 8, 2, true, true, false, 2>), g, dim3(256), dyn, st, p);
            else hipLaunchKernelGGL((conv_wino_kernel<4, 8, 2, false, true, false, 2>), g, dim3(256), dyn, st, p);
        }
        return ss_launch_status();
    }
    if (g_wino_variant == 3) {      // load-order experiment (VAR = 1)
        if (tbh == 8) {
            if (res) hipLaunchKernelGGL((conv_wino_kernel<8, 4, 2, true, true, false, 1>), g, dim3(256), dyn, st, p);
            else hipLaunchKernelGGL((conv_wino_kernel<8, 4, 2, false, true, false, 1>), g, dim3(256), dyn, st, p);
        } else {
            if (res) hipLaunchKernelGGL((conv_wino_kernel<4, 8, 2, true, true, false, 1>), g, dim3(256), dyn, st, p);
            else hipLaunchKernelGGL((conv_wino_kernel<4, 8, 2, false, true, false, 1>), g, dim3(256), dyn, st, p);
        }
        return ss_launch_status();
    }
#else
    const unsigned dyn = 0u;
#endif
    if (tbh == 8) {
        if (res) hipLaunchKernelGGL((conv_wino_kernel<8, 4, 2, true, true>), g, dim3(256), dyn, st, p);
        else hipLaunchKernelGGL((conv_wino_kernel<8, 4, 2, false, true>), g, dim3(256), dyn, st, p);
    } else {
        if (res) hipLaunchKernelGGL((conv_wino_kernel<4, 8, 2, true, true>), g, dim3(256), dyn, st, p);
        else hipLaunchKernelGGL((conv_wino_kernel<4, 8, 2, false, true>), g, dim3(256), dyn, st, p);
    }
    return ss_launch_status();
}

extern "C" int ss_conv3x3_wino_pool2_nhwc(const float* in, const float* packed, const float* bias, float* out, int n, int h, int w,
                                          int cin, int cout, int relu, int out_cs, int groups, long long in_gs, long long u_gs,
                                          long long out_gs, void* stream) {
    return wino_launch(in, packed, bias, nullptr, out, n, h, w, cin, cout, relu, out_cs, groups, in_gs, u_gs, out_gs, stream, false,
                       true);
}

extern "C" int ss_conv3x3_wino_nhwc(const float* in, const float* packed, const float* bias, const float* res, float* out,
                                    int n, int h, int w, int cin, int cout, int relu, int out_cs, int groups,
                                    long long in_gs, long long u_gs, long long out_gs, void* stream) {
    return wino_launch(in, packed, bias, res, out, n, h, w, cin, cout, relu, out_cs, groups, in_gs, u_gs, out_gs, stream, false);
}

extern "C" int ss_conv3x3_wino3_nhwc(const float* in, const float* packed3, const float* bias, const float* res, float* out,
                                     int n, int h, int w, int cin, int cout, int relu, int out_cs, int groups,
                                     long long in_gs, long long u_gs, long long out_gs, void* stream) {
    return wino_launch(in, packed3, bias, res, out, n, h, w, cin, cout, relu, out_cs, groups, in_gs, u_gs, out_gs, stream, true);
}
