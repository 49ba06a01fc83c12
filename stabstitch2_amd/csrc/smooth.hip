// SmoothNet glue (K11): per-vertex embeddings in front of the Conv3d stack and the mesh/path
// bookkeeping behind the decoder (smooth_network.py:29-34, 66-72, 139-157).  The three
// Conv3d(128,128,(5,3,3)) layers run on the implicit-GEMM engine in conv.hip (t = 7 frames).
#include "common.h"

// window wi covers frames wi*wstride + [0, t); tsflow = running sum of tsmotion inside the window,
// with the first entry optionally forced to 0 (test_online_tra.py:362-366)
__device__ __forceinline__ void window_flow(const float* __restrict__ ts, long long base, int tt, int v, int zero_first,
                                            float& fx, float& fy) {
    float ax = zero_first ? 0.f : ts[(base * SS_NV + v) * 2];
    float ay = zero_first ? 0.f : ts[(base * SS_NV + v) * 2 + 1];
    for (int s = 1; s <= tt; ++s) {
        ax = __fadd_rn(ax, ts[((base + s) * SS_NV + v) * 2]);
        ay = __fadd_rn(ay, ts[((base + s) * SS_NV + v) * 2 + 1]);
    }
    fx = ax;
    fy = ay;
}

__global__ void smooth_embed_kernel(const float* __restrict__ sm1, const float* __restrict__ sm2,
                                    const float* __restrict__ ts1, const float* __restrict__ ts2,
                                    const float* __restrict__ e1w, const float* __restrict__ e1b,
                                    const float* __restrict__ e3w, const float* __restrict__ e3b,
                                    float* __restrict__ hidden, int nw, int t, int wstride, int zero_first) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long total = (long long)nw * t * SS_NV * 128;
    if (idx >= total) return;
    int ch = (int)(idx & 127);
    long long r = idx >> 7;
    int v = (int)(r % SS_NV);
    r /= SS_NV;
    int tt = (int)(r % t);
    int wi = (int)(r / t);
    long long base = (long long)wi * wstride;
    int grp = ch >> 5, c = ch & 31;
    float x, y;
    const float *wq, *bq;
    if ((grp & 1) == 0) {
        const float* sm = grp == 0 ? sm1 : sm2;
        x = sm[((base + tt) * SS_NV + v) * 2];
        y = sm[((base + tt) * SS_NV + v) * 2 + 1];
        wq = e1w; bq = e1b;
    } else {
        window_flow(grp == 1 ? ts1 : ts2, base, tt, v, zero_first, x, y);
        wq = e3w; bq = e3b;
    }
    float o = __fadd_rn(__fadd_rn(__fmul_rn(x, wq[c * 2]), __fmul_rn(y, wq[c * 2 + 1])), bq[c]);
    hidden[idx] = fmaxf(o, 0.f);
}

extern "C" int ss_smooth_embed(const float* smesh1, const float* smesh2, const float* ts1, const float* ts2,
                               const float* e1w, const float* e1b, const float* e3w, const float* e3b, float* hidden,
                               int nw, int t, int wstride, int zero_first, void* stream) {
    if (!smesh1 || !smesh2 || !ts1 || !ts2 || !e1w || !e1b || !e3w || !e3b || !hidden || nw <= 0 || t <= 0 ||
        wstride <= 0)
        return SS_ERR_ARG;
    long long total = (long long)nw * t * SS_NV * 128;
    hipLaunchKernelGGL(smooth_embed_kernel, dim3(ss_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, smesh1, smesh2,
                       ts1, ts2, e1w, e1b, e3w, e3b, hidden, nw, t, wstride, zero_first);
    return ss_launch_status();
}

__global__ void smooth_finalize_kernel(const float* __restrict__ sm1, const float* __restrict__ sm2,
                                       const float* __restrict__ ts1, const float* __restrict__ ts2,
                                       const float* __restrict__ delta, float* __restrict__ om1,
                                       float* __restrict__ om2, float* __restrict__ op1, float* __restrict__ op2,
                                       float* __restrict__ smm1, float* __restrict__ smm2, float* __restrict__ sp1,
                                       float* __restrict__ sp2, int nw, int t, int wstride, int zero_first) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long total = (long long)nw * t * SS_NV;
    if (idx >= total) return;
    int v = (int)(idx % SS_NV);
    long long r = idx / SS_NV;
    int tt = (int)(r % t);
    int wi = (int)(r / t);
    long long base = (long long)wi * wstride;
    float4 d = *reinterpret_cast<const float4*>(delta + idx * 4);
    for (int view = 0; view < 2; ++view) {
        const float* sm = view ? sm2 : sm1;
        float mx = sm[((base + tt) * SS_NV + v) * 2], my = sm[((base + tt) * SS_NV + v) * 2 + 1];
        float fx, fy;
        window_flow(view ? ts2 : ts1, base, tt, v, zero_first, fx, fy);
        float dx = view ? d.z : d.x, dy = view ? d.w : d.y;
        float* om = view ? om2 : om1;
        float* op = view ? op2 : op1;
        float* smm = view ? smm2 : smm1;
        float* sp = view ? sp2 : sp1;
        if (om) { om[idx * 2] = mx; om[idx * 2 + 1] = my; }
        if (op) { op[idx * 2] = fx; op[idx * 2 + 1] = fy; }
        if (smm) { smm[idx * 2] = __fsub_rn(mx, dx); smm[idx * 2 + 1] = __fsub_rn(my, dy); }
        if (sp) { sp[idx * 2] = __fadd_rn(fx, dx); sp[idx * 2 + 1] = __fadd_rn(fy, dy); }
    }
}

extern "C" int ss_smooth_finalize(const float* smesh1, const float* smesh2, const float* ts1, const float* ts2,
                                  const float* delta, float* ori_mesh1, float* ori_mesh2, float* ori_path1,
                                  float* ori_path2, float* smooth_mesh1, float* smooth_mesh2, float* smooth_path1,
                                  float* smooth_path2, int nw, int t, int wstride, int zero_first, void* stream) {
    if (!smesh1 || !smesh2 || !ts1 || !ts2 || !delta || nw <= 0 || t <= 0 || wstride <= 0) return SS_ERR_ARG;
    long long total = (long long)nw * t * SS_NV;
    hipLaunchKernelGGL(smooth_finalize_kernel, dim3(ss_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, smesh1,
                       smesh2, ts1, ts2, delta, ori_mesh1, ori_mesh2, ori_path1, ori_path2, smooth_mesh1, smooth_mesh2,
                       smooth_path1, smooth_path2, nw, t, wstride, zero_first);
    return ss_launch_status();
}

// ------------------------------------------------------------------------------------------------
// The clip's tensors straight from the sliding windows (test_online_tra.py:377-392, test_metric_ssd.py:415-436):
// window 0 contributes its t frames, window k >= 1 its last frame -- frame f comes from window w = max(f - (t-1), 0) at
// position tt = f - w.  Meshes are written directly ([n][63][2], n = nw + t - 1); the metric harness's paths are chained
// across windows SEQUENTIALLY like the reference's frame loop:
//     ori_path[f]    = ori_path[f-1] + (op_w[t-1] - op_w[t-2])
//     smooth_path[f] = ori_path[f]   + (sp_w[t-1] - op_w[t-1])            (f >= t, w = f - t + 1)
// pass 1 (parallel over frames) leaves the two bracketed increments in place, pass 2 (one thread per coordinate) runs
// the chain over them.  Replaces smooth_finalize + a dozen torch cat / cumsum calls of mesh-sized tensors.
__global__ void smooth_stitch_kernel(const float* __restrict__ sm1, const float* __restrict__ sm2,
                                     const float* __restrict__ ts1, const float* __restrict__ ts2,
                                     const float* __restrict__ delta, float* __restrict__ om1, float* __restrict__ om2,
                                     float* __restrict__ smm1, float* __restrict__ smm2, float* __restrict__ op2,
                                     float* __restrict__ sp2, int nw, int t) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long n = (long long)nw + t - 1;
    if (idx >= n * SS_NV) return;
    const int v = (int)(idx % SS_NV);
    const long long f = idx / SS_NV;
    const long long w = f >= t - 1 ? f - (t - 1) : 0;
    const int tt = (int)(f - w);
    const float4 d = *reinterpret_cast<const float4*>(delta + ((w * t + tt) * SS_NV + v) * 4);
    const float m1x = sm1[idx * 2], m1y = sm1[idx * 2 + 1], m2x = sm2[idx * 2], m2y = sm2[idx * 2 + 1];
    om1[idx * 2] = m1x; om1[idx * 2 + 1] = m1y;
    om2[idx * 2] = m2x; om2[idx * 2 + 1] = m2y;
    smm1[idx * 2] = __fsub_rn(m1x, d.x); smm1[idx * 2 + 1] = __fsub_rn(m1y, d.y);
    smm2[idx * 2] = __fsub_rn(m2x, d.z); smm2[idx * 2 + 1] = __fsub_rn(m2y, d.w);
    if (!op2) return;
    float fx, fy;
    window_flow(ts2, w, tt, v, 1, fx, fy);                // op_w[tt]
    if (f < t) {                                          // window 0: the paths themselves
        op2[idx * 2] = fx; op2[idx * 2 + 1] = fy;
        sp2[idx * 2] = __fadd_rn(fx, d.z); sp2[idx * 2 + 1] = __fadd_rn(fy, d.w);
    } else {                                              // increments; chained by smooth_path_chain_kernel
        float gx, gy;
        window_flow(ts2, w, tt - 1, v, 1, gx, gy);        // op_w[t-2]
        op2[idx * 2] = __fsub_rn(fx, gx); op2[idx * 2 + 1] = __fsub_rn(fy, gy);
        sp2[idx * 2] = __fsub_rn(__fadd_rn(fx, d.z), fx); sp2[idx * 2 + 1] = __fsub_rn(__fadd_rn(fy, d.w), fy);
    }
}

__global__ void smooth_path_chain_kernel(float* __restrict__ op2, float* __restrict__ sp2, long long n, int t) {
    const int c = threadIdx.x;                            // one of the 126 coordinates
    if (c >= SS_NV * 2) return;
    float acc = op2[(long long)(t - 1) * SS_NV * 2 + c];
    for (long long f = t; f < n; ++f) {
        acc = __fadd_rn(acc, op2[f * SS_NV * 2 + c]);
        op2[f * SS_NV * 2 + c] = acc;
        sp2[f * SS_NV * 2 + c] = __fadd_rn(acc, sp2[f * SS_NV * 2 + c]);
    }
}

// smesh*/ts* [n,7,9,2] (n = nw + t - 1 frames, window stride 1, first tsmotion of every window zeroed), delta [nw,t,7,9,4]
// -> ori_mesh1/2, smooth_mesh1/2 [n,7,9,2]; ori_path2 / smooth_path2 [n,7,9,2] (both or neither)
extern "C" int ss_smooth_stitch(const float* smesh1, const float* smesh2, const float* ts1, const float* ts2,
                                const float* delta, float* ori_mesh1, float* ori_mesh2, float* smooth_mesh1,
                                float* smooth_mesh2, float* ori_path2, float* smooth_path2, int nw, int t, void* stream) {
    if (!smesh1 || !smesh2 || !ts1 || !ts2 || !delta || !ori_mesh1 || !ori_mesh2 || !smooth_mesh1 || !smooth_mesh2 ||
        nw <= 0 || t < 2 || (!ori_path2) != (!smooth_path2))
        return SS_ERR_ARG;
    const long long n = (long long)nw + t - 1;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(smooth_stitch_kernel, dim3(ss_cdiv(n * SS_NV, 256)), dim3(256), 0, st, smesh1, smesh2, ts1, ts2, delta,
                       ori_mesh1, ori_mesh2, smooth_mesh1, smooth_mesh2, ori_path2, smooth_path2, nw, t);
    if (ori_path2 && n > t)
        hipLaunchKernelGGL(smooth_path_chain_kernel, dim3(1), dim3(128), 0, st, ori_path2, smooth_path2, n, t);
    return ss_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Streaming mode: the sliding windows live in buffers of FIXED address (the steady state is a captured HIP graph), so a new
// frame SHIFTS them: rings [R][W][E] drop slot 0 and take row src + src_off[r] as slot W-1; in the same launch `blocks`
// blocks of `block` floats move inside `state` (dst block b at b * stride, its source `delta` floats further: last frame's
// spatial motions become "previous").  One launch where the torch form took cat + copy per ring (8 + 1 launches per pair).
struct WinPushArgs { long long src_off[8]; };
__global__ __launch_bounds__(256) void window_push_kernel(float* __restrict__ ring, const float* __restrict__ src,
                                                          WinPushArgs a, int R, int W, int E, float* __restrict__ state,
                                                          int blocks, int block, long long stride, long long delta, int per) {
    const int r = blockIdx.x, tid = threadIdx.x;
    if (r == R) {                                        // the state move (source and destination never overlap: delta >= block)
        for (int i = tid; i < blocks * block; i += 256) {
            const int b = i / block, e = i - b * block;
            state[(long long)b * stride + e] = state[(long long)b * stride + delta + e];
        }
        return;
    }
    float* g = ring + (long long)r * W * E;
    const int keep = (W - 1) * E;
    constexpr int MAXV = 8;                              // keep <= 2048 floats (host-checked)
    float v[MAXV];
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        const int i = tid + 256 * k;
        v[k] = i < keep ? g[E + i] : 0.f;
    }
    __syncthreads();                                     // every slot is read before any is overwritten
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        const int i = tid + 256 * k;
        if (i < keep) g[i] = v[k];
    }
    for (int e = tid; e < E; e += 256) g[keep + e] = src[a.src_off[r / per] + (long long)(r % per) * E + e];
}

// groups x per rings: ring g * per + j takes the row at src + src_off[g] + j * elems (per = S streams advancing together, one
// offset per ring KIND; per = 1: one offset per ring)
extern "C" int ss_window_push_groups(float* ring, const float* src, const long long* src_off, int groups, int per, int window,
                                     int elems, float* state, int blocks, int block, long long stride, long long delta,
                                     void* stream) {
    const int rings = groups * per;
    if (!ring || !src || !src_off || groups <= 0 || groups > 8 || per <= 0 || per > 4096 || window < 2 || elems <= 0 ||
        (long long)(window - 1) * elems > 2048 || blocks < 0 || (blocks > 0 && (!state || block <= 0 || delta < block)) ||
        (blocks > 1 && stride < delta + block))       // block b + 1's destination must not reach into block b's source
        return SS_ERR_ARG;
    WinPushArgs a;
    for (int r = 0; r < 8; ++r) a.src_off[r] = r < groups ? src_off[r] : 0;
    hipLaunchKernelGGL(window_push_kernel, dim3(rings + (blocks > 0 ? 1 : 0)), dim3(256), 0, (hipStream_t)stream, ring, src, a,
                       rings, window, elems, state, blocks, block, stride, delta, per);
    return ss_launch_status();
}

extern "C" int ss_window_push(float* ring, const float* src, const long long* src_off, int rings, int window, int elems,
                              float* state, int blocks, int block, long long stride, long long delta, void* stream) {
    return ss_window_push_groups(ring, src, src_off, rings, 1, window, elems, state, blocks, block, stride, delta, stream);
}
