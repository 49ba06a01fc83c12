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
//   image) x 64 output channels x all 16 positions; K loop over cin in chunks of 16.
//   per chunk:
//     * raw (2TBH+2) x (2TBW+2) x 16ch input patch: coalesced 16-byte buffer loads issued one chunk ahead (halo / M tail
//       / channel tail through the descriptor's bounds check: invalid lanes get offset 0xFFFFFFFF and read zeros),
//       registers -> LDS `raw`;
//     * input transform B^T d B: 512 tasks (tile, channel quad, transform row) -> 2 per thread, 8 ds_read_b128 +
//       8 float4 adds + 4 ds_write_b128 each (the transform row is wave-uniform: no divergence); result V[pos][tile][k]
//       k-contiguous in LDS (row stride 20 dwords: conflict-free 16-byte reads);
//     * GEMMs on v_mfma_f32_16x16x4_f32: wave w owns output channels [16w, 16w+16) of the block for ALL 16 positions and
//       both 16-tile halves: 16 x 2 accumulator tiles of 4 registers = 128 registers; per chunk 128 MFMAs per wave
//       against 32 ds_read_b128 (A operand, shared by the 4 waves) and 16 global 16-byte loads (B operand: the
//       transformed filters are PRE-PACKED in the exact register layout of the B operand, so they go global -> VGPR,
//       fully coalesced, no LDS, no reuse lost: each element is needed by exactly one wave of the workgroup);
//   epilogue: the 16 positions of one (tile, cout) sit in ONE lane, so A^T M A is register arithmetic; the 2x2 results
//   are staged through LDS so that bias (folded BatchNorm), residual, ReLU and the NHWC store run on whole 256-byte pixel
//   rows with 16-byte accesses (out-of-range offsets for pixels outside the image: no branches).
// Two workgroups per CU (58 KB LDS, <= 256 registers): one transforms while the other feeds the MFMA pipe; everything
// outside the MFMA phase runs at raised wave priority so that it is not queued behind the other workgroup's MFMAs.
#include "common.h"

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
    SsDiv32 divNcb;
    long long in_gs, u_gs, out_gs;      // element strides between groups
    unsigned in_bytes, out_bytes, u_bytes;
#ifdef SS_TUNING
    unsigned long long* dbg;            // per-workgroup phase stamps (tools/diag_wino.py)
#endif
};

#ifdef SS_TUNING
#define W_STAMP(i) do { if (p.dbg) ts[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define W_STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ __amdgpu_buffer_rsrc_t w_rsrc(const float* base, unsigned bytes) {
    unsigned long long a = (unsigned long long)base;
    unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    void* ub = (void*)(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(ub, 0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

template <int TBH, int TBW>
__global__ __launch_bounds__(256, 2) void conv_wino_kernel(WinoP p) {
    static_assert(TBH * TBW == 32, "32 tiles per workgroup");
    constexpr int RH = 2 * TBH + 2, RW = 2 * TBW + 2;      // raw input patch (pixels)
    constexpr int RPIX = RH * RW;
    constexpr int RS = 24;                                  // raw pixel stride in dwords (16 channels + pad)
    constexpr int NE = (RPIX * 4 + 255) / 256;              // 16-byte raw items per thread
    constexpr int VS = 20;                                  // V row stride in dwords (16 k + pad)
    __shared__ __attribute__((aligned(16))) float raw[RPIX * RS];
    __shared__ __attribute__((aligned(16))) float V[16 * 32 * VS];      // the epilogue reuses it as the output stage (32 KB)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef SS_TUNING
    unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    // Everything outside the MFMA phase (setup, transforms, epilogue) is VALU / LDS work that shares its SIMD with the
    // OTHER resident workgroup's MFMAs; at equal priority every VALU instruction waits for an MFMA boundary (measured:
    // 3.9k cycles per input transform, 18k per epilogue).  Raised priority lets it through (same rule as conv.hip).
    __builtin_amdgcn_s_setprio(3);
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
    const unsigned cbk = lin - mb * p.ncb;                  // 64-channel output block
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
        const int ry = pix / RW, rx = pix - ry * RW;
        const int iy = oy0 - 1 + ry, ix = ox0 - 1 + rx;
        const bool ok = item < RPIX * 4 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        rbase[e] = ((((unsigned)img * p.H + iy) * p.W + ix) * (unsigned)p.C + 4u * q) * 4u;
        rinv[e] = ok ? 0u : 0xFFFFFFFFu;
        rlds[e] = pix * RS + 4 * q;
    }
    const int myq = tid & 3;                                // channel quad of this thread's raw items AND transform tasks

    // transform tasks: id = row * 128 + tile * 4 + quad; thread takes ids tid and tid + 256 -> rows (tid >> 7) and + 2
    const int t_tile = (tid >> 2) & 31;
    const int t_ty = t_tile / TBW, t_tx = t_tile - t_ty * TBW;
    const int t_row0 = __builtin_amdgcn_readfirstlane(tid >> 7);       // 0 or 1, wave-uniform
    const int t_src = ((2 * t_ty) * RW + 2 * t_tx) * RS + 4 * myq;      // top-left pixel of the 4x4 patch
    const int t_dst = t_tile * VS + 4 * myq;

    // MFMA operand addressing
    const int a_off = (lane & 15) * VS + 4 * (lane >> 4);               // + pos * 32 * VS + half * 16 * VS
    const unsigned u_lane = (unsigned)lane * 16u;
    const unsigned cb16 = cbk * 4u + (unsigned)wave;                    // 16-channel block of this wave
    // packed filters: [cout/16][chunk][pos][lane][4] floats -> per (cb16, chunk): 16 KB
    const unsigned u_wave = cb16 * (unsigned)p.nchunk * 16384u;

    w_f32x4 acc[16][2];
#pragma unroll
    for (int a = 0; a < 16; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (w_f32x4){0.f, 0.f, 0.f, 0.f};

    w_u32x4 rr[NE];
    w_f32x4 ua[8], ub[8];
    auto raw_issue = [&](int c) {
        const unsigned coff = (unsigned)c * 64u;
        const unsigned cinv = ((c * 16 + 4 * myq) < p.C) ? 0u : 0xFFFFFFFFu;
#pragma unroll
        for (int e = 0; e < NE; ++e)
            rr[e] = __builtin_amdgcn_raw_buffer_load_b128(rin, (rbase[e] + coff) | rinv[e] | cinv, 0, 0);
    };
    auto u_issue = [&](w_f32x4* dst, int c, int half) {
        const int so = (int)__builtin_amdgcn_readfirstlane(u_wave + (unsigned)c * 16384u + (unsigned)half * 8192u);
#pragma unroll
        for (int k = 0; k < 8; ++k)
            dst[k] = __builtin_bit_cast(w_f32x4, __builtin_amdgcn_raw_buffer_load_b128(ru, u_lane + 1024u * k, so, 0));
    };
    auto transform = [&](int row) {
        // B^T rows: 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3; `row` is wave-uniform (scalar branches, packed adds)
        const int ra = row == 0 ? 0 : (row == 2 ? 2 : 1);
        const int rb = row == 3 ? 3 : (row == 2 ? 1 : 2);
        w_f32x4 x[4], y[4], r[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            x[b] = *reinterpret_cast<const w_f32x4*>(&raw[t_src + (ra * RW + b) * RS]);
            y[b] = *reinterpret_cast<const w_f32x4*>(&raw[t_src + (rb * RW + b) * RS]);
        }
        if (row == 1) {
#pragma unroll
            for (int b = 0; b < 4; ++b) r[b] = x[b] + y[b];
        } else {
#pragma unroll
            for (int b = 0; b < 4; ++b) r[b] = x[b] - y[b];
        }
        float* dst = &V[(row * 4) * 32 * VS + t_dst];
        *reinterpret_cast<w_f32x4*>(dst) = r[0] - r[2];
        *reinterpret_cast<w_f32x4*>(dst + 32 * VS) = r[1] + r[2];
        *reinterpret_cast<w_f32x4*>(dst + 2 * 32 * VS) = r[2] - r[1];
        *reinterpret_cast<w_f32x4*>(dst + 3 * 32 * VS) = r[1] - r[3];
    };
    auto mma_half = [&](const w_f32x4* u, int half) {
        // A operands one position ahead of the MFMAs that use them (LDS latency behind 8 MFMAs = 256 cycles)
        w_f32x4 n0 = *reinterpret_cast<const w_f32x4*>(&V[half * 8 * 32 * VS + a_off]);
        w_f32x4 n1 = *reinterpret_cast<const w_f32x4*>(&V[half * 8 * 32 * VS + 16 * VS + a_off]);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int pos = half * 8 + k;
            const w_f32x4 a0 = n0, a1 = n1;
            if (k < 7) {
                n0 = *reinterpret_cast<const w_f32x4*>(&V[(pos + 1) * 32 * VS + a_off]);
                n1 = *reinterpret_cast<const w_f32x4*>(&V[(pos + 1) * 32 * VS + 16 * VS + a_off]);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                acc[pos][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s], u[k][s], acc[pos][0], 0, 0, 0);
                acc[pos][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], u[k][s], acc[pos][1], 0, 0, 0);
            }
        }
    };

    raw_issue(0);
    u_issue(ua, 0, 0);
    W_STAMP(1);
    for (int c = 0; c < p.nchunk; ++c) {
        // registers -> LDS raw patch (everyone is past the previous chunk's transform: it read `raw` before barrier B)
#pragma unroll
        for (int e = 0; e < NE; ++e)
            if (NE * 256 == RPIX * 4 || tid + 256 * e < RPIX * 4) *reinterpret_cast<w_u32x4*>(&raw[rlds[e]]) = rr[e];
        __syncthreads();                        // A: raw visible; every wave has finished the previous chunk's MFMAs (V free)
        if (c == 0) W_STAMP(5);
        if (c == 1) W_STAMP(7);
        transform(t_row0);
        transform(t_row0 + 2);
        __syncthreads();                        // B: V complete
        if (c == 0) W_STAMP(6);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(0);
        u_issue(ub, c, 1);                      // second half of this chunk's filters: used ~2000 cycles from now
        __builtin_amdgcn_sched_barrier(0);
        mma_half(ua, 0);
        __builtin_amdgcn_sched_barrier(0);
        {
            // next chunk: raw patch and first half of the filters.  Branch free (the last iteration re-requests its own
            // chunk, results unused): with a conditional the compiler has to wait for the loads of the path that issued
            // none, i.e. it drains the whole prefetch before the second half of the MFMAs.
            const int cn = c + 1 < p.nchunk ? c + 1 : c;
            raw_issue(cn);
            u_issue(ua, cn, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        mma_half(ub, 1);
        __builtin_amdgcn_s_setprio(3);
        if (c == 0) W_STAMP(4);
    }
    W_STAMP(2);

    // ---------------------------------------------------------------- epilogue: Y = A^T M A, bias, residual, ReLU
    // The 16 positions of one (tile, cout) sit in one lane: A^T M A is register arithmetic.  The 2x2 results go through
    // LDS (the V buffer, free now) so that global memory sees whole 256-byte pixel rows as 16-byte accesses: stage
    // S[pixel = tile * 4 + 2a + b][64 couts], the cout index rotated by 16 * ((pixel >> 4) & 3) so that the four 16-lane
    // groups of a wave (four different tiles) write four different bank groups.
    float* __restrict__ out = p.out + (long long)grp * p.out_gs;
    const float* __restrict__ res = p.res ? p.res + (long long)grp * p.out_gs : nullptr;
    const __amdgpu_buffer_rsrc_t rout = w_rsrc(out, p.out_bytes);
    const __amdgpu_buffer_rsrc_t rres = w_rsrc(res ? res : out, p.out_bytes);
    // this thread's 8 output items: (pixel = e * 16 + tid / 16, channel quad = tid % 16)
    const int cq = tid & 15;
    unsigned goff[8];
    w_u32x4 rv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int px = e * 16 + (tid >> 4);
        const int tile = px >> 2, a = (px >> 1) & 1, b = px & 1;
        const int ty = tile / TBW, tx = tile - ty * TBW;
        const int oy = oy0 + 2 * ty + a, ox = ox0 + 2 * tx + b;
        const bool ok = oy < p.H && ox < p.W;
        goff[e] = ok ? ((((unsigned)img * p.H + oy) * p.W + ox) * (unsigned)p.out_cs + cbk * 64u + 4u * cq) * 4u : 0xFFFFFFFFu;
    }
    if (res) {
#pragma unroll
        for (int e = 0; e < 8; ++e) rv[e] = __builtin_amdgcn_raw_buffer_load_b128(rres, goff[e], 0, 0);
    }
    w_f32x4 bias4 = (w_f32x4){0.f, 0.f, 0.f, 0.f};
    if (p.bias) bias4 = *reinterpret_cast<const w_f32x4*>(p.bias + (long long)grp * p.Co + cbk * 64 + 4 * cq);
    __syncthreads();                            // every wave is done reading V
    {
        const int cw = wave * 16 + (lane & 15);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int tile = half * 16 + 4 * (lane >> 4) + r;
                float T[4][2];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float m0 = acc[4 * i + 0][half][r], m1 = acc[4 * i + 1][half][r];
                    const float m2 = acc[4 * i + 2][half][r], m3 = acc[4 * i + 3][half][r];
                    T[i][0] = (m0 + m1) + m2;
                    T[i][1] = (m1 - m2) - m3;
                }
                // rotation: (pixel >> 4) & 3 = (tile >> 2) & 3 = lane >> 4 for every pixel of this tile
                float* srow = &V[(tile * 4) * 64 + ((cw + 16 * (lane >> 4)) & 63)];
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    srow[b * 64] = (T[0][b] + T[1][b]) + T[2][b];
                    srow[(2 + b) * 64] = (T[1][b] - T[2][b]) - T[3][b];
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int px = e * 16 + (tid >> 4);
        w_f32x4 v = *reinterpret_cast<const w_f32x4*>(&V[px * 64 + ((4 * cq + 16 * ((px >> 4) & 3)) & 63)]);
        v = v + bias4;
        if (res) v = v + __builtin_bit_cast(w_f32x4, rv[e]);
        if (p.relu) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(w_u32x4, v), rout, goff[e], 0, 0);
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
// Filter transform + packing: U = G g G^T in fp64, rounded once to fp32, stored in the B-operand register layout of the
// kernel above:  U[cout/16][chunk][pos][lane][s] = (G g G^T)[pos] of (cout = 16 cb + (lane & 15), cin = 16 chunk + 4 (lane >> 4) + s),
// zero for cin >= C.   wgt: [cout][1][3][3][cin] (BN folded).
__global__ void wino_pack_kernel(const float* __restrict__ wgt, float* __restrict__ U, int cout, int cin, int nchunk,
                                 long long w_gs, long long u_gs) {
    const long long per = (long long)(cout / 16) * nchunk * 16 * 64;       // float4 slots per group
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= per) return;
    const int grp = blockIdx.y;
    const int lane = (int)(idx & 63);
    const int pos = (int)((idx >> 6) & 15);
    const long long cc = idx >> 10;
    const int chunk = (int)(cc % nchunk);
    const int cb = (int)(cc / nchunk);
    const int co = cb * 16 + (lane & 15);
    const int i = pos >> 2, j = pos & 3;
    const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
    float v[4];
    for (int s = 0; s < 4; ++s) {
        const int ci = chunk * 16 + 4 * (lane >> 4) + s;
        double acc = 0.0;
        if (ci < cin) {
            const float* g = wgt + (long long)grp * w_gs + (long long)co * 9 * cin + ci;
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) acc += G[i][a] * (double)g[(a * 3 + b) * cin] * G[j][b];
        }
        v[s] = (float)acc;
    }
    reinterpret_cast<float4*>(U + (long long)grp * u_gs)[idx] = make_float4(v[0], v[1], v[2], v[3]);
}

extern "C" long long ss_wino_packed_floats(int cout, int cin) {
    if (cout <= 0 || cin <= 0 || (cout & 15)) return 0;
    return (long long)(cout / 16) * ss_cdiv(cin, 16) * 16 * 64 * 4;
}

extern "C" int ss_wino_pack(const float* wgt, float* packed, int cout, int cin, int groups, void* stream) {
    if (!wgt || !packed || cout <= 0 || cin <= 0 || (cout & 15) || (cin & 3) || groups <= 0) return SS_ERR_ARG;
    const int nchunk = ss_cdiv(cin, 16);
    const long long per = (long long)(cout / 16) * nchunk * 16 * 64;
    hipLaunchKernelGGL(wino_pack_kernel, dim3(ss_cdiv(per, 256), groups), dim3(256), 0, (hipStream_t)stream, wgt, packed,
                       cout, cin, nchunk, (long long)cout * 9 * cin, per * 4);
    return ss_launch_status();
}

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
    return eff >= 0.70 && wgs >= 512;
}

extern "C" int ss_conv3x3_wino_nhwc(const float* in, const float* packed, const float* bias, const float* res, float* out,
                                    int n, int h, int w, int cin, int cout, int relu, int out_cs, int groups,
                                    long long in_gs, long long u_gs, long long out_gs, void* stream) {
    if (!in || !packed || !out || n <= 0 || h <= 0 || w <= 0 || cin <= 0 || (cin & 3) || cout <= 0 || (cout & 63) ||
        groups <= 0 || out_cs < cout)
        return SS_ERR_ARG;
    const long long in_elems = (long long)n * h * w * cin, out_elems = (long long)n * h * w * out_cs;
    const long long u_floats = ss_wino_packed_floats(cout, cin);
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
    p.ncb = (unsigned)(cout / 64);
    p.divNcb = ss_div32_make(p.ncb);
    p.in_gs = in_gs; p.u_gs = u_gs; p.out_gs = out_gs;
    p.in_bytes = (unsigned)(in_elems * 4);
    p.out_bytes = (unsigned)(out_elems * 4);
    p.u_bytes = (unsigned)(u_floats * 4);
#ifdef SS_TUNING
    p.dbg = ss_tuning_dbg;
#endif
    const long long wgs = (long long)n * p.nbx * p.nby * p.ncb;
    if (wgs >= (1ll << 31)) return SS_ERR_UNSUPPORTED;
    dim3 g((unsigned)wgs, 1, groups);
    hipStream_t st = (hipStream_t)stream;
    if (tbh == 8) hipLaunchKernelGGL((conv_wino_kernel<8, 4>), g, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((conv_wino_kernel<4, 8>), g, dim3(256), 0, st, p);
    return ss_launch_status();
}
