// Dense thin-plate-spline warp and fusion (K12/K13).
//
// The reference materialises a [B,66,Hc*Wc] basis tensor (~500 MB at 720p) and makes ~15
// elementwise passes over it before a bmm and a gather.  Here one thread owns one
// canvas pixel: the 63 RBF terms of both views are evaluated together in registers (packed fp32 math,
// hardware log2) from wave-uniform control points / coefficients (scalar loads, no LDS), the sampling coordinate never leaves the register file, the
// bilinear taps are gathered straight from the planar source frames, and for the AVERAGE mode the
// two (three) views are fused before the single store.  HBM traffic = source frames once (L2 absorbs
// the 4-tap overlap) + the canvas once.
#include "common.h"
#include "device_math.h"

// grid_sample(bilinear, zeros, align_corners=True) taps: x = (xn+1)/2*(W-1), out-of-range taps -> 0
__device__ __forceinline__ float sample_fast(const float* __restrict__ pl, float xn, float yn, int W, int H) {
    float x = ((xn + 1.f) / 2.f) * (float)(W - 1);
    float y = ((yn + 1.f) / 2.f) * (float)(H - 1);
    float xf = fminf(fmaxf(floorf(x), -4.f), (float)W + 4.f);
    float yf = fminf(fmaxf(floorf(y), -4.f), (float)H + 4.f);
    int x0 = (int)xf, y0 = (int)yf, x1 = x0 + 1, y1 = y0 + 1;
    // ATen grid_sampler_2d: nw = (ix_se - ix)(iy_se - iy), ne = (ix - ix_sw)(iy_sw - iy), ...
    float w00 = (xf + 1.f - x) * (yf + 1.f - y), w01 = (x - xf) * (yf + 1.f - y);
    float w10 = (xf + 1.f - x) * (y - yf), w11 = (x - xf) * (y - yf);
    bool vx0 = (unsigned)x0 < (unsigned)W, vx1 = (unsigned)x1 < (unsigned)W;
    bool vy0 = (unsigned)y0 < (unsigned)H, vy1 = (unsigned)y1 < (unsigned)H;
    float r = 0.f;
    if (vx0 && vy0) r += pl[(long long)y0 * W + x0] * w00;
    if (vx1 && vy0) r += pl[(long long)y0 * W + x1] * w01;
    if (vx0 && vy1) r += pl[(long long)y1 * W + x0] * w10;
    if (vx1 && vy1) r += pl[(long long)y1 * W + x1] * w11;
    return r;
}

// zero padding on an all-ones plane: sum of the in-range tap weights (grid_sample of a ones image)
__device__ __forceinline__ float fast_mask(float xn, float yn, int w, int h) {
    float xx = ((xn + 1.f) / 2.f) * (float)(w - 1), yy = ((yn + 1.f) / 2.f) * (float)(h - 1);
    float xf = fminf(fmaxf(floorf(xx), -4.f), (float)w + 4.f);
    float yf = fminf(fmaxf(floorf(yy), -4.f), (float)h + 4.f);
    int x0 = (int)xf, y0 = (int)yf;
    bool vx0 = (unsigned)x0 < (unsigned)w, vx1 = (unsigned)(x0 + 1) < (unsigned)w;
    bool vy0 = (unsigned)y0 < (unsigned)h, vy1 = (unsigned)(y0 + 1) < (unsigned)h;
    float r = 0.f;
    if (vx0 && vy0) r += (xf + 1.f - xx) * (yf + 1.f - yy);
    if (vx1 && vy0) r += (xx - xf) * (yf + 1.f - yy);
    if (vx0 && vy1) r += (xf + 1.f - xx) * (yy - yf);
    if (vx1 && vy1) r += (xx - xf) * (yy - yf);
    return r;
}

// Both rows of a wave (tile = 64 columns x 8 rows, wave w owns rows w and w + 4) through tps_eval_rows: the row-only part
// of the radial terms from the wave's own LDS table.  -> normalised sampling coordinates of (x, ya) and (x, yb).
__device__ __forceinline__ void warp_rows_coords(const float* __restrict__ src, const float* __restrict__ T, ss_f2* tab,
                                                 int lane, int x, int ya, int yb, int hc, int wc, ss_f2& xn, ss_f2& yn) {
    const float gx = linspace_at(-1.f, 1.f, wc, min(x, wc - 1));
    const float gya = linspace_at(-1.f, 1.f, hc, ya), gyb = linspace_at(-1.f, 1.f, hc, min(yb, hc - 1));
    tps_rows_table(src, gya, gyb, lane, tab);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the same wave reads it back: ordering only
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    tps_eval_rows(src, T, tab, gx, gya, gyb, xn, yn);
}

// append_mask: emit one extra channel = warp of an all-ones plane (test_online_tra.py:144-147)
__global__ __launch_bounds__(256) void tps_warp_kernel(const float* __restrict__ U, const float* __restrict__ source,
                                                       const float* __restrict__ T, float* __restrict__ out, int c,
                                                       int h, int w, int hc, int wc, int mode, int append_mask) {
    __shared__ ss_f2 dytab[4][64];
    const int b = blockIdx.z;
    const int lx = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + lx;
    const int ya = blockIdx.y * 8 + wv, yb = ya + 4;
    if (ya >= hc) return;                       // whole waves only: lanes past the canvas edge still help build the table
    ss_f2 xn2, yn2;
    warp_rows_coords(source + (long long)b * SS_NV * 2, T + (long long)b * 2 * SS_NT, dytab[wv], lx, x, ya, yb, hc, wc, xn2, yn2);
    if (x >= wc) return;
    const long long hw = (long long)h * w, ohw = (long long)hc * wc;
    const int co = c + (append_mask ? 1 : 0);
    const float* in = U + (long long)b * c * hw;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int y = rr ? yb : ya;
        if (y >= hc) break;
        const float xn = rr ? xn2.y : xn2.x, yn = rr ? yn2.y : yn2.x;
        float* o = out + (long long)b * co * ohw + (long long)y * wc + x;
        if (mode == SS_WARP_NORMAL) {
            SsTaps t = taps_normal(xn, yn, w, h);
            long long ia = (long long)t.y0 * w + t.x0, ib = (long long)t.y1 * w + t.x0;
            long long ic = (long long)t.y0 * w + t.x1, id = (long long)t.y1 * w + t.x1;
            for (int ch = 0; ch < c; ++ch) {
                const float* pl = in + ch * hw;
                o[ch * ohw] = blend4(t, pl[ia], pl[ib], pl[ic], pl[id]);
            }
            if (append_mask) o[c * ohw] = blend4(t, 1.f, 1.f, 1.f, 1.f);
        } else {
            for (int ch = 0; ch < c; ++ch) o[ch * ohw] = sample_fast(in + ch * hw, xn, yn, w, h);
            if (append_mask) o[c * ohw] = fast_mask(xn, yn, w, h);
        }
    }
}

static int launch_warp(const float* U, const float* source, const float* T, float* out, int b, int c, int h, int w,
                       int hc, int wc, int mode, int append_mask, void* stream) {
    if (!U || !source || !T || !out || b <= 0 || c <= 0 || h <= 1 || w <= 1 || hc <= 1 || wc <= 1 ||
        (mode != SS_WARP_NORMAL && mode != SS_WARP_FAST))
        return SS_ERR_ARG;
    dim3 g(ss_cdiv(wc, 64), ss_cdiv(hc, 8), b);
    hipLaunchKernelGGL(tps_warp_kernel, g, dim3(256), 0, (hipStream_t)stream, U, source, T, out, c, h, w, hc, wc, mode,
                       append_mask);
    return ss_launch_status();
}

extern "C" int ss_tps_warp_nchw(const float* U, const float* source, const float* T, float* out, int b, int c, int h,
                                int w, int hc, int wc, int mode, void* stream) {
    return launch_warp(U, source, T, out, b, c, h, w, hc, wc, mode, 0, stream);
}

extern "C" int ss_tps_warp_mask_nchw(const float* U, const float* source, const float* T, float* out, int b, int c,
                                     int h, int w, int hc, int wc, int mode, void* stream) {
    return launch_warp(U, source, T, out, b, c, h, w, hc, wc, mode, 1, stream);
}

// ------------------------------------------------------------------------------------------------
// fused render, AVERAGE fusion (test_online_tra.py:138-142; three-view chaining threeview:486-490)
struct RenderViews {
    const float* img[3];
};

__global__ __launch_bounds__(256) void tps_warp_views_kernel(RenderViews rv, const float* __restrict__ source,
                                                             const float* __restrict__ T, float* __restrict__ out,
                                                             int h, int w, int hc, int wc, int mode) {
    __shared__ ss_f2 dytab[4][64];
    const int b = blockIdx.z;
    const int lx = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + lx;
    const int ya = blockIdx.y * 8 + wv, yb = ya + 4;
    if (ya >= hc) return;
    ss_f2 xn2, yn2;
    warp_rows_coords(source + (long long)b * SS_NV * 2, T + (long long)b * 2 * SS_NT, dytab[wv], lx, x, ya, yb, hc, wc, xn2, yn2);
    if (x >= wc) return;
    const long long hw = (long long)h * w, ohw = (long long)hc * wc;
    const float* in = b == 0 ? rv.img[0] : (b == 1 ? rv.img[1] : rv.img[2]);
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int y = rr ? yb : ya;
        if (y >= hc) break;
        const float xn = rr ? xn2.y : xn2.x, yn = rr ? yn2.y : yn2.x;
        float* o = out + (long long)b * 4 * ohw + (long long)y * wc + x;
        if (mode == SS_WARP_NORMAL) {
            SsTaps t = taps_normal(xn, yn, w, h);
            long long ia = (long long)t.y0 * w + t.x0, ib = (long long)t.y1 * w + t.x0;
            long long ic = (long long)t.y0 * w + t.x1, id = (long long)t.y1 * w + t.x1;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const float* pl = in + ch * hw;
                o[ch * ohw] = blend4(t, pl[ia], pl[ib], pl[ic], pl[id]);
            }
            o[3 * ohw] = blend4(t, 1.f, 1.f, 1.f, 1.f);
        } else {
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) o[ch * ohw] = sample_fast(in + ch * hw, xn, yn, w, h);
            o[3 * ohw] = fast_mask(xn, yn, w, h);
        }
    }
}

extern "C" int ss_tps_warp_views(const float* const* imgs, const float* source, const float* T, float* out, int views,
                                 int h, int w, int hc, int wc, int mode, void* stream) {
    if (!imgs || !source || !T || !out || views < 1 || views > 3 || h <= 1 || w <= 1 || hc <= 1 || wc <= 1 ||
        (mode != SS_WARP_NORMAL && mode != SS_WARP_FAST))
        return SS_ERR_ARG;
    RenderViews rv;
    for (int i = 0; i < 3; ++i) rv.img[i] = i < views ? imgs[i] : nullptr;
    for (int i = 0; i < views; ++i)
        if (!rv.img[i]) return SS_ERR_ARG;
    dim3 g(ss_cdiv(wc, 64), ss_cdiv(hc, 8), views);
    hipLaunchKernelGGL(tps_warp_views_kernel, g, dim3(256), 0, (hipStream_t)stream, rv, source, T, out, h, w, hc, wc,
                       mode);
    return ss_launch_status();
}

// small canvas-sized elementwise helpers of the harnesses
__global__ void affine_kernel(const float* __restrict__ in, float* __restrict__ out, float add, float mul, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = __fmul_rn(__fadd_rn(in[i], add), mul);
}
extern "C" int ss_add_mul(const float* in, float* out, float add, float mul, long long n, void* stream) {
    if (!in || !out || n <= 0) return SS_ERR_ARG;
    hipLaunchKernelGGL(affine_kernel, dim3(ss_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, in, out, add, mul, n);
    return ss_launch_status();
}
__global__ void mask_union_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                  long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = __fsub_rn(__fadd_rn(a[i], b[i]), __fmul_rn(a[i], b[i]));
}
extern "C" int ss_mask_union(const float* a, const float* b, float* out, long long n, void* stream) {
    if (!a || !b || !out || n <= 0) return SS_ERR_ARG;
    hipLaunchKernelGGL(mask_union_kernel, dim3(ss_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, a, b, out, n);
    return ss_launch_status();
}

__device__ __forceinline__ float avg_fuse(float a, float b) {
    float s = __fadd_rn(__fadd_rn(a, b), 1e-6f);
    return __fadd_rn(__fmul_rn(a, a / s), __fmul_rn(b, b / s));
}

// ---- footprints: which 64 x 8 canvas tiles can a view contribute to? -------------------------------------------------
// Outside a view's footprint the reference's clamped-index sampler returns no image content, only the rounding residue
// of (P + Q) - P - Q (|r| <~ 1e-2 grey levels, different on every machine -- DESIGN.md 4); yet those pixels cost the full
// 63-term spline evaluation, and on the benchmark canvas a third of all (pixel, view) pairs lie there.  With a footprint
// block the fused render skips them and treats the view's contribution as exactly 0.
//   lattice[v][i][j] = (xn, yn) = sampling coordinate of view v at canvas pixel (x = 32 j, y = 8 i), i <= ceil(hc / 8),
//   j <= 2 ceil(wc / 64) (half-tile spacing along the long side of a tile: a tile is judged on its four corners AND the
//   midpoints of its two long edges; the last row / column lies on or beyond the canvas edge: the spline is defined everywhere);
//   hull[v] = (xmin, xmax, ymin, ymax): bounding box of the view's 63 control points in normalised canvas coordinates --
//   the view's mesh on the canvas; its border vertices map exactly onto the image border (TPS interpolation).
// A tile is OUTSIDE view v when BOTH hold:
//   (a) it lies outside the mesh hull grown by 8 canvas pixels, and
//   (b) the exactly evaluated sampling coordinates of its four corners and two long-edge midpoints all lie beyond the same
//       image side by more than 8 source pixels (to come back, the spline would have to bend by 8 px within 32 canvas
//       pixels OUTSIDE its own mesh hull, where it has no control points and extrapolates smoothly).
// This is a test, not a proof: callers who need the reference's arithmetic at every pixel pass footprint = NULL.
// (A rigorous interpolation-error bound from sum |T_k| was tried instead of (b)'s fixed margin: the RBF weights of a
// near-affine warp cancel, the bound does not -- it came out at 100-140 source pixels and kept a third of the skippable
// tiles.)  tests/test_gpu_parity.py::test_render_footprint_skipping checks on pipeline meshes that every skipped pixel is
// outside (validity mask of the full evaluation ~ 0) and that nothing else changes.
__global__ void render_lattice_kernel(const float* __restrict__ source, const float* __restrict__ T,
                                      float* __restrict__ fp, long long frame_stride, int views, int hc, int wc,
                                      int ny, int nx, float guard, int* __restrict__ watch_i, float* __restrict__ watch_f) {
    const int fv = blockIdx.y;                       // frame * views + view
    const int frame = fv / views, view = fv - frame * views;
    float* lattice = fp + frame * frame_stride + (long long)view * ny * nx * 2;
    float* margin = fp + frame * frame_stride + (long long)views * ny * nx * 2 + 4 * view;       // hull (xmin, xmax, ymin, ymax)
    const float* src = source + (long long)fv * SS_NV * 2;
    const float* Tx = T + (long long)fv * 2 * SS_NT;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x == 0 && view == 0 && threadIdx.x >= 64 && threadIdx.x < 68)       // the frame's tile-class counters (render_order_kernel)
        reinterpret_cast<unsigned*>(fp + frame * frame_stride + (long long)views * ny * nx * 2 + 4 * views)[threadIdx.x - 64] = 0u;
    if (blockIdx.x == 0 && threadIdx.x < 2) {       // hull of the control points: thread 0 -> x range, thread 1 -> y range
        float lo = INFINITY, hi = -INFINITY;
        for (int k = 0; k < SS_NV; ++k) {
            const float c = src[2 * k + threadIdx.x];
            lo = fminf(lo, c);
            hi = fmaxf(hi, c);
        }
        margin[2 * threadIdx.x] = lo;
        margin[2 * threadIdx.x + 1] = hi;
    }
    if (watch_i && blockIdx.x == 0 && view == 0 && threadIdx.x < 64) {
        // the streaming canvas' overflow watcher over ALL views of this frame (canvas_watch_kernel's arithmetic; frame = stream):
        // the control points are read here anyway, and a launch of its own is a node on the push's critical path
        const float* s = source + (long long)frame * views * SS_NV * 2;
        float xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
        bool bad = false;
        for (int k = threadIdx.x; k < views * SS_NV; k += 64) {
            const float x = s[2 * k], y = s[2 * k + 1];
            bad = bad || x != x || y != y;
            xmin = fminf(xmin, x); xmax = fmaxf(xmax, x);
            ymin = fminf(ymin, y); ymax = fmaxf(ymax, y);
        }
        canvas_watch_update(xmin, xmax, ymin, ymax, bad, guard, watch_i + frame * 4, watch_f + frame * 4);
    }
    if (idx >= ny * nx) return;
    const int i = idx / nx, j = idx - i * nx;
    const float gx = -1.f + 2.f * (float)(32 * j) / (float)(wc - 1), gy = -1.f + 2.f * (float)(8 * i) / (float)(hc - 1);
    float xn, yn;
    tps_eval_fast(src, Tx, Tx + SS_NT, gx, gy, xn, yn);
    lattice[idx * 2] = xn;
    lattice[idx * 2 + 1] = yn;
}

// bit v set = view v may contribute to tile (by, bx)
__device__ __forceinline__ unsigned tile_views(const float* __restrict__ fp, int views, int by, int bx, int ny, int nx,
                                               int h, int w, int hc, int wc) {
    const float* hull = fp + (long long)views * ny * nx * 2;
    // the tile in normalised canvas coordinates, grown by 8 canvas pixels
    const float sxn = 2.f / (float)(wc - 1), syn = 2.f / (float)(hc - 1);
    const float tx0 = -1.f + sxn * (float)(64 * bx - 8), tx1 = -1.f + sxn * (float)(64 * bx + 63 + 8);
    const float ty0 = -1.f + syn * (float)(8 * by - 8), ty1 = -1.f + syn * (float)(8 * by + 7 + 8);
    const float mx = 1.f + 16.f / (float)w, my = 1.f + 16.f / (float)h;       // 8 source pixels beyond the image
    unsigned mask = 0u;
    for (int v = 0; v < views; ++v) {
        const float* L = fp + ((long long)v * ny * nx + (long long)by * nx + 2 * bx) * 2;      // lattice column 2 bx = pixel 64 bx
        float xlo = INFINITY, xhi = -INFINITY, ylo = INFINITY, yhi = -INFINITY;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {            // corners and long-edge midpoints
                const float xs = L[(r * nx + c) * 2], ys = L[(r * nx + c) * 2 + 1];
                xlo = fminf(xlo, xs); xhi = fmaxf(xhi, xs);
                ylo = fminf(ylo, ys); yhi = fmaxf(yhi, ys);
            }
        const bool off_hull = tx1 < hull[4 * v] || tx0 > hull[4 * v + 1] || ty1 < hull[4 * v + 2] || ty0 > hull[4 * v + 3];
        const bool off_image = xlo > mx || xhi < -mx || ylo > my || yhi < -my;
        if (!(off_hull && off_image)) mask |= 1u << v;
    }
    return mask;
}


// Tile order of one frame: the fused render takes its tiles most expensive class first (tiles reached by 3, 2, 1, 0 views cost
// ~2 : 1 : 0.5 : 0 spline evaluations per pixel).  With ~11 workgroups per CU and mixed costs a row-major order left the CUs 25 %
// apart (measured: 55 us per frame against 45 for the cost-weighted sum of the tile classes); longest-first hands the expensive
// tiles out evenly and fills in with the cheap ones.
//   entry = bx | by << 12 | mask << 24
// Round 6: one LIST per class (4 x nt entries behind 4 counters that render_lattice_kernel zeroes) instead of one sorted table: a
// tile's slot no longer depends on the totals of the classes in front of it, so the tiles are classified by nt / 256 workgroups,
// one tile per thread, one global atomic per wave and class -- 3-4 us instead of the single workgroup's 9 (two views, 3100 tiles) /
// 16 us (three views, 4158) on a streaming push's critical path.  The render finds its tile from the counters (render_tile_entry).
// Which tile a workgroup of a class gets depends on the atomics' arrival order; every tile is still rendered exactly once.
__global__ __launch_bounds__(256) void render_order_kernel(float* __restrict__ fp, long long frame_stride, int views, int h,
                                                           int w, int hc, int wc, int ny, int nx) {
    float* f = fp + (long long)blockIdx.y * frame_stride;
    unsigned* cnt = reinterpret_cast<unsigned*>(f + (long long)views * ny * nx * 2 + 4 * views);
    unsigned* lists = cnt + 4;
    const int nbx = (nx - 1) / 2, nby = ny - 1, nt = nbx * nby;
    const int t = blockIdx.x * 256 + (int)threadIdx.x;
    const int lane = threadIdx.x & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
    const int by = t / nbx, bx = t - by * nbx;
    const unsigned mm = t < nt ? tile_views(f, views, by, bx, ny, nx, h, w, hc, wc) : 0xFFu;
    const int c = mm == 0xFFu ? -1 : 3 - __popc(mm);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned long long b = __ballot(c == k);
        if (b == 0ull) continue;                    // (wave-uniform)
        unsigned wb = 0u;
        if (lane == 0) wb = atomicAdd(&cnt[k], (unsigned)__popcll(b));
        wb = (unsigned)__shfl((int)wb, 0, 64);
        if (c == k) lists[(long long)k * nt + wb + (unsigned)__popcll(b & below)] = (unsigned)bx | ((unsigned)by << 12) | (mm << 24);
    }
}

// the i-th tile of a frame in cost order: class lists back to back
__device__ __forceinline__ unsigned render_tile_entry(const float* __restrict__ fp, int views, int ny, int nx, unsigned i) {
    const unsigned* cnt = reinterpret_cast<const unsigned*>(fp + (long long)views * ny * nx * 2 + 4 * views);
    const unsigned nt = (unsigned)((nx - 1) / 2 * (ny - 1));
    const unsigned c0 = cnt[0], c1 = cnt[1], c2 = cnt[2];
    unsigned k = 0u, off = i;
    if (i >= c0) { k = 1u; off = i - c0; }
    if (i >= c0 + c1) { k = 2u; off = i - c0 - c1; }
    if (i >= c0 + c1 + c2) { k = 3u; off = i - c0 - c1 - c2; }
    return cnt[4 + (unsigned long long)k * nt + off];
}

extern "C" long long ss_render_footprint_floats(int views, int hc, int wc) {
    if (views <= 0 || hc <= 1 || wc <= 1) return 0;
    return (long long)views * ((long long)(ss_cdiv(hc, 8) + 1) * (2 * ss_cdiv(wc, 64) + 1) * 2 + 4) +
           4 + 4ll * ss_cdiv(hc, 8) * ss_cdiv(wc, 64);       // + 4 class counters + 4 class lists of tiles
}

// footprints of `frames` x `views` splines in one launch: fp [frames][ lattice [views][ny][nx][2] | margin [views][2] ]
static int render_footprints_launch(const float* source, const float* T, float* fp, int frames, int views, int h, int w,
                                    int hc, int wc, float guard, int* watch_i, float* watch_f, void* stream) {
    if (!source || !T || !fp || frames <= 0 || views <= 0 || views > 3 || h <= 1 || w <= 1 || hc <= 1 || wc <= 1)
        return SS_ERR_ARG;
    const int ny = ss_cdiv(hc, 8) + 1, nx = 2 * ss_cdiv(wc, 64) + 1;
    // the footprint of frame f (lattice of every view, then the margins) lives at fp + f * ss_render_footprint_floats(...)
    const long long stride = ss_render_footprint_floats(views, hc, wc);
    if (ny > 4096 || nx > 4096) return SS_ERR_UNSUPPORTED;
    // grid.y = (frame, view), capped at 65535 by HIP: whole frames per launch
    const int fmax = 65535 / views;
    for (int f0 = 0; f0 < frames; f0 += fmax) {
        const int nf = frames - f0 < fmax ? frames - f0 : fmax;
        hipLaunchKernelGGL(render_lattice_kernel, dim3(ss_cdiv(ny * nx, 128), nf * views), dim3(128), 0, (hipStream_t)stream,
                           source + (long long)f0 * views * SS_NV * 2, T + (long long)f0 * views * 2 * SS_NT,
                           fp + (long long)f0 * stride, stride, views, hc, wc, ny, nx, guard, watch_i ? watch_i + 4 * f0 : nullptr,
                           watch_f ? watch_f + 4 * f0 : nullptr);
    }
    const int nt = (ny - 1) * ((nx - 1) / 2);
    for (int f0 = 0; f0 < frames; f0 += 65535) {
        const int nf = frames - f0 < 65535 ? frames - f0 : 65535;
        hipLaunchKernelGGL(render_order_kernel, dim3(ss_cdiv(nt, 256), nf), dim3(256), 0, (hipStream_t)stream, fp + (long long)f0 * stride,
                           stride, views, h, w, hc, wc, ny, nx);
    }
    return ss_launch_status();
}

extern "C" int ss_render_footprints(const float* source, const float* T, float* fp, int frames, int views, int h, int w,
                                    int hc, int wc, void* stream) {
    return render_footprints_launch(source, T, fp, frames, views, h, w, hc, wc, 0.f, nullptr, nullptr, stream);
}

// the same + the streaming canvas' overflow watcher (ss_canvas_watch's update of watch_i / watch_f [frames][4], frame = stream) inside
// the lattice launch: `source` is what the watcher reads, and the push saves a graph node
extern "C" int ss_render_footprints_watch(const float* source, const float* T, float* fp, int frames, int views, int h, int w,
                                          int hc, int wc, float guard, int* watch_i, float* watch_f, void* stream) {
    if (!watch_i || !watch_f || !(guard >= 0.f)) return SS_ERR_ARG;
    return render_footprints_launch(source, T, fp, frames, views, h, w, hc, wc, guard, watch_i, watch_f, stream);
}

__device__ __forceinline__ void sample3(const float* __restrict__ in, float xn, float yn, int w, int h, long long hw, int mode,
                                        float (&v)[3]) {
    if (mode == SS_WARP_NORMAL) {
        SsTaps t = taps_normal(xn, yn, w, h);
        long long ia = (long long)t.y0 * w + t.x0, ib = (long long)t.y1 * w + t.x0;
        long long ic = (long long)t.y0 * w + t.x1, id = (long long)t.y1 * w + t.x1;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const float* pl = in + ch * hw;
            v[ch] = blend4(t, pl[ia], pl[ib], pl[ic], pl[id]);
        }
    } else {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) v[ch] = sample_fast(in + ch * hw, xn, yn, w, h);
    }
}

// the same samples from a decoded uint8 frame [h][w][3] (what cv2.imread hands the reference, test_online_tra.py:252-258):
// uint8 -> fp32 is exact, so the values equal those of sample3 on the converted planes bit for bit
__device__ __forceinline__ void sample3_u8(const unsigned char* __restrict__ in, float xn, float yn, int w, int h, int mode,
                                           float (&v)[3]) {
    if (mode == SS_WARP_NORMAL) {
        SsTaps t = taps_normal(xn, yn, w, h);
        const long long ia = ((long long)t.y0 * w + t.x0) * 3, ib = ((long long)t.y1 * w + t.x0) * 3;
        const long long ic = ((long long)t.y0 * w + t.x1) * 3, id = ((long long)t.y1 * w + t.x1) * 3;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch)
            v[ch] = blend4(t, (float)in[ia + ch], (float)in[ib + ch], (float)in[ic + ch], (float)in[id + ch]);
    } else {
        float x = ((xn + 1.f) / 2.f) * (float)(w - 1);
        float y = ((yn + 1.f) / 2.f) * (float)(h - 1);
        float xf = fminf(fmaxf(floorf(x), -4.f), (float)w + 4.f);
        float yf = fminf(fmaxf(floorf(y), -4.f), (float)h + 4.f);
        int x0 = (int)xf, y0 = (int)yf, x1 = x0 + 1, y1 = y0 + 1;
        float w00 = (xf + 1.f - x) * (yf + 1.f - y), w01 = (x - xf) * (yf + 1.f - y);
        float w10 = (xf + 1.f - x) * (y - yf), w11 = (x - xf) * (y - yf);
        bool vx0 = (unsigned)x0 < (unsigned)w, vx1 = (unsigned)x1 < (unsigned)w;
        bool vy0 = (unsigned)y0 < (unsigned)h, vy1 = (unsigned)y1 < (unsigned)h;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {            // sample_fast, term by term
            float r = 0.f;
            if (vx0 && vy0) r += (float)in[((long long)y0 * w + x0) * 3 + ch] * w00;
            if (vx1 && vy0) r += (float)in[((long long)y0 * w + x1) * 3 + ch] * w01;
            if (vx0 && vy1) r += (float)in[((long long)y1 * w + x0) * 3 + ch] * w10;
            if (vx1 && vy1) r += (float)in[((long long)y1 * w + x1) * 3 + ch] * w11;
            v[ch] = r;
        }
    }
}
__device__ __forceinline__ unsigned char render_to_u8(float v) {          // `.astype(np.uint8)` (frameio.hip to_u8)
    if (!(fabsf(v) < 2147483648.f)) return 0;
    return (unsigned char)((unsigned)((int)v) & 255u);
}

// fp == nullptr: every view is evaluated at every pixel (the reference's arithmetic everywhere, residues included);
// otherwise tiles are classified by `tile_views` and a view that cannot reach a tile contributes exactly 0 there.
// U8: frames are decoded uint8 [h][w][3] and the canvas is written as the video frame uint8 [hc][wc][3] (`.astype(np.uint8)`
// of the fused values, test_online_tra.py:413) -- the fp32 frame planes and the fp32 canvas never exist in memory.
// One launch renders a whole clip: blockIdx.y = frame (the hardware hands out a frame's tiles, most expensive first,
// then the next frame's: the light tail of one frame runs beside the heavy head of the next -- 32 launch boundaries and
// 32 partially filled last rounds per clip less than one launch per frame).  Per-frame strides: `img_fs` / `out_fs` in
// BYTES (frames of a view / canvases are equally spaced), `fp_fs` in floats; source [frame][VIEWS][63][2], T [frame][VIEWS][2][66].
template <int VIEWS, bool U8, bool FOLD = false>
__global__ __launch_bounds__(256) void render_average_kernel(RenderViews rv, const float* __restrict__ source,
                                                             const float* __restrict__ T, const float* __restrict__ fp,
                                                             float* __restrict__ out, int h, int w, int hc, int wc,
                                                             int mode, long long img_fs, long long out_fs,
                                                             long long fp_fs) {
    {
        const long long frame = blockIdx.y;
        source += frame * (VIEWS * SS_NV * 2);
        T += frame * (VIEWS * 2 * SS_NT);
        if (fp) fp += frame * fp_fs;
        out = reinterpret_cast<float*>(reinterpret_cast<char*>(out) + frame * out_fs);
#pragma unroll
        for (int k = 0; k < VIEWS; ++k)
            rv.img[k] = reinterpret_cast<const float*>(reinterpret_cast<const char*>(rv.img[k]) + frame * img_fs);
    }
    unsigned char* const out8 = reinterpret_cast<unsigned char*>(out);
    // workgroup = 64 x 8 canvas pixels; wave w owns rows w and w + 4 of the tile and evaluates, for every view that
    // reaches the tile, that view's spline at its 64 columns of both rows (packed over the rows)
    const int lx = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long hw = (long long)h * w, ohw = (long long)hc * wc;
    const int ny = (hc + 7) / 8 + 1, nx = 2 * ((wc + 63) / 64) + 1, nbx = (nx - 1) / 2;
    int tbx, tby;
    unsigned mask;
    if (fp) {                                       // tile and its view set from the frame's order table (longest first)
        const unsigned e = (unsigned)__builtin_amdgcn_readfirstlane((int)render_tile_entry(fp, VIEWS, ny, nx, blockIdx.x));
        tbx = (int)(e & 0xFFFu); tby = (int)((e >> 12) & 0xFFFu); mask = e >> 24;
    } else {
        tby = blockIdx.x / nbx; tbx = blockIdx.x - tby * nbx; mask = (1u << VIEWS) - 1u;
    }
    const int x = tbx * 64 + lx;
#ifdef SS_TUNING
    if ((mode >> 8) == 9) mask = ((tbx + tby) & 1) ? 3u : 1u;          // checkerboard of single / both
    else if ((mode >> 8) == 10) mask = (tbx < nbx / 2) ? 1u : 3u;        // left half single, right half both
    else if (mode >> 8) mask = (unsigned)(mode >> 8) - 1u;          // forced tile class (timing experiments)
    mode &= 0xFF;
#endif
    const int ya = tby * 8 + wv, yb = ya + 4;
    if (tbx * 64 >= wc || ya >= hc) return;       // (whole waves only: the lanes past the canvas edge still help build the table)
    const bool xin = x < wc;
    const float gx = linspace_at(-1.f, 1.f, wc, min(x, wc - 1));
    if (mask == 0u) {                               // no view reaches this tile: avg_fuse(0, 0) = 0
        if (xin) {
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                if (U8) {
                    out8[((long long)ya * wc + x) * 3 + ch] = 0;
                    if (yb < hc) out8[((long long)yb * wc + x) * 3 + ch] = 0;
                } else {
                    out[ch * ohw + (long long)ya * wc + x] = 0.f;
                    if (yb < hc) out[ch * ohw + (long long)yb * wc + x] = 0.f;
                }
            }
        }
        return;
    }
    // every view that reaches the tile: its spline at this lane's column of the wave's two rows (packed over the rows);
    // the row-only part of the radial terms comes from a per-wave LDS table (tps_rows_table)
    __shared__ ss_f2 dytab[4][VIEWS][64];
    const float gya = linspace_at(-1.f, 1.f, hc, ya), gyb = linspace_at(-1.f, 1.f, hc, min(yb, hc - 1));
#pragma unroll
    for (int k = 0; k < VIEWS; ++k)
        if (mask & (1u << k)) tps_rows_table<FOLD>(source + k * SS_NV * 2, gya, gyb, lx, dytab[wv][k]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // same wave reads it back: ordering only, no barrier
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float va[VIEWS][3], vb[VIEWS][3];
#pragma unroll
    for (int k = 0; k < VIEWS; ++k) {
        if (mask & (1u << k)) {
            ss_f2 px, py;
            tps_eval_rows<FOLD>(source + k * SS_NV * 2, T + k * 2 * SS_NT, dytab[wv][k], gx, gya, gyb, px, py);
            if (U8) {
                const unsigned char* img8 = reinterpret_cast<const unsigned char*>(rv.img[k]);
                sample3_u8(img8, px.x, py.x, w, h, mode, va[k]);
                sample3_u8(img8, px.y, py.y, w, h, mode, vb[k]);
            } else {
                sample3(rv.img[k], px.x, py.x, w, h, hw, mode, va[k]);
                sample3(rv.img[k], px.y, py.y, w, h, hw, mode, vb[k]);
            }
        } else {
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) { va[k][ch] = 0.f; vb[k][ch] = 0.f; }
        }
    }
    if (!xin) return;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        // the chained fusion in the reference's order ((1 (+) 2) (+) 3), zeros in the places of views that do not reach
        float fa = avg_fuse(va[0][ch], va[1][ch]), fb = avg_fuse(vb[0][ch], vb[1][ch]);
        if (VIEWS == 3) { fa = avg_fuse(fa, va[2][ch]); fb = avg_fuse(fb, vb[2][ch]); }
        if (U8) {
            out8[((long long)ya * wc + x) * 3 + ch] = render_to_u8(fa);
            if (yb < hc) out8[((long long)yb * wc + x) * 3 + ch] = render_to_u8(fb);
        } else {
            out[ch * ohw + (long long)ya * wc + x] = fa;
            if (yb < hc) out[ch * ohw + (long long)yb * wc + x] = fb;
        }
    }
}

static int render_average_launch(const void* const* imgs, const float* source, const float* T, const float* footprint,
                                 long long footprint_floats, void* out, int frames, long long img_fs, long long out_fs,
                                 int views, int h, int w, int hc, int wc, int mode, void* stream, bool u8) {
    if (!imgs || !source || !T || !out || frames <= 0 || (views != 2 && views != 3) || h <= 1 || w <= 1 || hc <= 1 ||
        wc <= 1 || ((mode & 0xEF) != SS_WARP_NORMAL && (mode & 0xEF) != SS_WARP_FAST))
        return SS_ERR_ARG;
#ifndef SS_TUNING
    if (mode >> 8) return SS_ERR_ARG;
#endif
    const bool fold = (mode & SS_WARP_EPS_FOLD) != 0;
    mode &= ~SS_WARP_EPS_FOLD;
    // a footprint row is only meaningful for the (views, canvas) it was built for: the kernel indexes its lattice and tile
    // order with this geometry, so a row of another size is an argument error, not an out-of-bounds read
    const long long fp_fs = ss_render_footprint_floats(views, hc, wc);
    if (footprint && footprint_floats != fp_fs) return SS_ERR_ARG;
    RenderViews rv;
    for (int i = 0; i < 3; ++i) rv.img[i] = i < views ? static_cast<const float*>(imgs[i]) : nullptr;
    for (int i = 0; i < views; ++i)
        if (!rv.img[i]) return SS_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    // grid.y = frame; HIP caps grid.y at 65535: longer clips go in several launches
    for (int f0 = 0; f0 < frames; f0 += 65535) {
        const int nf = frames - f0 < 65535 ? frames - f0 : 65535;
        dim3 g(ss_cdiv(wc, 64) * ss_cdiv(hc, 8), nf, 1);
        RenderViews r = rv;
        for (int i = 0; i < views; ++i)
            r.img[i] = reinterpret_cast<const float*>(reinterpret_cast<const char*>(rv.img[i]) + (long long)f0 * img_fs);
        const float* s_ = source + (long long)f0 * views * SS_NV * 2;
        const float* t_ = T + (long long)f0 * views * 2 * SS_NT;
        const float* p_ = footprint ? footprint + (long long)f0 * fp_fs : nullptr;
        float* o = reinterpret_cast<float*>(static_cast<char*>(out) + (long long)f0 * out_fs);
        if (fold) {        // opt-in: the reference's + 1e-6 folded into the row table (not its arithmetic; see device_math.h)
            if (views == 2) {
                if (u8) hipLaunchKernelGGL((render_average_kernel<2, true, true>), g, dim3(256), 0, st, r, s_, t_, p_, o, h, w, hc, wc, mode, img_fs, out_fs, fp_fs);
                else hipLaunchKernelGGL((render_average_kernel<2, false, true>), g, dim3(256), 0, st, r, s_, t_, p_, o, h, w, hc, wc, mode, img_fs, out_fs, fp_fs);
            } else {
                if (u8) hipLaunchKernelGGL((render_average_kernel<3, true, true>), g, dim3(256), 0, st, r, s_, t_, p_, o, h, w, hc, wc, mode, img_fs, out_fs, fp_fs);
                else hipLaunchKernelGGL((render_average_kernel<3, false, true>), g, dim3(256), 0, st, r, s_, t_, p_, o, h, w, hc, wc, mode, img_fs, out_fs, fp_fs);
            }
        } else if (views == 2) {
            if (u8) hipLaunchKernelGGL((render_average_kernel<2, true>), g, dim3(256), 0, st, r, s_, t_, p_, o, h, w, hc, wc, mode, img_fs, out_fs, fp_fs);
            else hipLaunchKernelGGL((render_average_kernel<2, false>), g, dim3(256), 0, st, r, s_, t_, p_, o, h, w, hc, wc, mode, img_fs, out_fs, fp_fs);
        } else {
            if (u8) hipLaunchKernelGGL((render_average_kernel<3, true>), g, dim3(256), 0, st, r, s_, t_, p_, o, h, w, hc, wc, mode, img_fs, out_fs, fp_fs);
            else hipLaunchKernelGGL((render_average_kernel<3, false>), g, dim3(256), 0, st, r, s_, t_, p_, o, h, w, hc, wc, mode, img_fs, out_fs, fp_fs);
        }
    }
    return ss_launch_status();
}

extern "C" int ss_render_average(const float* const* imgs, const float* source, const float* T, const float* footprint,
                                 long long footprint_floats, float* out, int views, int h, int w, int hc, int wc, int mode,
                                 void* stream) {
    return render_average_launch(reinterpret_cast<const void* const*>(imgs), source, T, footprint, footprint_floats, out, 1,
                                 0, 0, views, h, w, hc, wc, mode, stream, false);
}

extern "C" int ss_render_average_u8(const unsigned char* const* frames, const float* source, const float* T,
                                    const float* footprint, long long footprint_floats, unsigned char* out, int views, int h,
                                    int w, int hc, int wc, int mode, void* stream) {
    return render_average_launch(reinterpret_cast<const void* const*>(frames), source, T, footprint, footprint_floats, out, 1,
                                 0, 0, views, h, w, hc, wc, mode, stream, true);
}

// whole clip, one launch: view k's frame f at views[k] + f * 3 h w floats (planar fp32 [n,3,h,w]), canvas f at
// out + f * 3 hc wc floats, source [n,V,63,2], T [n,V,2,66], footprint [n][ss_render_footprint_floats] or NULL
extern "C" int ss_render_average_clip(const float* const* views_base, const float* source, const float* T,
                                      const float* footprint, long long footprint_floats, float* out, int frames, int views,
                                      int h, int w, int hc, int wc, int mode, void* stream) {
    return render_average_launch(reinterpret_cast<const void* const*>(views_base), source, T, footprint, footprint_floats,
                                 out, frames, 12ll * h * w, 12ll * hc * wc, views, h, w, hc, wc, mode, stream, false);
}

// the same from decoded uint8 frames [n,h,w,3] per view to uint8 video frames [n,hc,wc,3]
extern "C" int ss_render_average_clip_u8(const unsigned char* const* views_base, const float* source, const float* T,
                                         const float* footprint, long long footprint_floats, unsigned char* out, int frames,
                                         int views, int h, int w, int hc, int wc, int mode, void* stream) {
    return render_average_launch(reinterpret_cast<const void* const*>(views_base), source, T, footprint, footprint_floats,
                                 out, frames, 3ll * h * w, 3ll * hc * wc, views, h, w, hc, wc, mode, stream, true);
}

// ------------------------------------------------------------------------------------------------
// LINEAR fusion (linear_blender, test_online_tra.py:34-58).  Whole-canvas statistics (mask centroids,
// projection range of the overlap) are accumulated with device atomics into a 16-word scalar block at
// the head of the workspace; no host round trip.
//   ws: [0..15] scalars | X [hc*wc] | tmp [hc*wc]
//   scalars (as 64-bit words): 0 cnt1, 1 sumr1, 2 sumc1, 3 cnt2, 4 sumr2, 5 sumc2; floats at [12] pmin-key, [13] pmax-key
__device__ __forceinline__ unsigned f2key(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__global__ void lb_init_kernel(unsigned long long* s) {
    if (threadIdx.x < 6) s[threadIdx.x] = 0ull;
    unsigned* u = reinterpret_cast<unsigned*>(s);
    if (threadIdx.x == 0) { u[12] = 0xffffffffu; u[13] = 0u; }
}

// 2-D indexing (block = 64 columns x 4 rows, grid-stride over row groups): no per-pixel integer division
__global__ __launch_bounds__(256) void lb_centroid_kernel(const float* __restrict__ m1, const float* __restrict__ m2,
                                                          unsigned long long* s, int hc, int wc) {
    __shared__ unsigned long long red[6];
    if (threadIdx.x < 6) red[threadIdx.x] = 0ull;
    __syncthreads();
    unsigned c1 = 0, r1 = 0, k1 = 0, c2 = 0, r2 = 0, k2 = 0;      // per-thread partials fit 32 bits (<= hc/ (4 gridDim.y) rows)
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    if (c < wc) {
        for (int r = blockIdx.y * 4 + (threadIdx.x >> 6); r < hc; r += gridDim.y * 4) {
            long long i = (long long)r * wc + c;
            if (m1[i] != 0.f) { k1 += 1; r1 += r; c1 += c; }
            if (m2[i] != 0.f) { k2 += 1; r2 += r; c2 += c; }
        }
    }
    unsigned v[6] = {k1, r1, c1, k2, r2, c2};       // a wave's sums still fit 32 bits (<= 64 lanes x ~24 rows x 1882)
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        unsigned t = v[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
        if ((threadIdx.x & 63) == 0) atomicAdd(&red[k], (unsigned long long)t);
    }
    __syncthreads();
    if (threadIdx.x < 6) atomicAdd(&s[threadIdx.x], red[threadIdx.x]);
}

struct LbCenters {
    float c1r, c1c, v0, v1;
};
__device__ __forceinline__ LbCenters lb_centers(const unsigned long long* s) {
    LbCenters L;
    L.c1r = (float)((double)s[1] / (double)s[0]);
    L.c1c = (float)((double)s[2] / (double)s[0]);
    float c2r = (float)((double)s[4] / (double)s[3]);
    float c2c = (float)((double)s[5] / (double)s[3]);
    L.v0 = __fsub_rn(c2r, L.c1r);
    L.v1 = __fsub_rn(c2c, L.c1c);
    return L;
}
__device__ __forceinline__ float lb_proj(const LbCenters& L, int r, int c) {
    return __fadd_rn(__fmul_rn(__fsub_rn((float)r, L.c1r), L.v0), __fmul_rn(__fsub_rn((float)c, L.c1c), L.v1));
}
__device__ __forceinline__ bool lb_overlap(float a, float b) { return rintf(__fmul_rn(a, b)) != 0.f; }

__global__ __launch_bounds__(256) void lb_range_kernel(const float* __restrict__ m1, const float* __restrict__ m2,
                                                       unsigned long long* s, int hc, int wc) {
    // the centres cost four fp64 divisions: once per workgroup, not per thread (that was 55 us per frame)
    __shared__ LbCenters Ls;
    __shared__ float smn[4], smx[4];
    if (threadIdx.x == 0) Ls = lb_centers(s);
    __syncthreads();
    const LbCenters L = Ls;
    float mn = INFINITY, mx = -INFINITY;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    if (c < wc) {
        for (int r = blockIdx.y * 4 + (threadIdx.x >> 6); r < hc; r += gridDim.y * 4) {
            long long i = (long long)r * wc + c;
            if (lb_overlap(m1[i], m2[i])) {
                float p = lb_proj(L, r, c);
                mn = fminf(mn, p);
                mx = fmaxf(mx, p);
            }
        }
    }
    mn = ss_wave_min(mn);
    mx = ss_wave_max(mx);
    if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        mn = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
        mx = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
        unsigned* u = reinterpret_cast<unsigned*>(s);
        if (mn != INFINITY) atomicMin(&u[12], f2key(mn));
        if (mx != -INFINITY) atomicMax(&u[13], f2key(mx));
    }
}

// X = ref_only + (1 - ovl_mask) * m1
__global__ void lb_premask_kernel(const float* __restrict__ m1, const float* __restrict__ m2,
                                  const unsigned long long* s, float* __restrict__ X, int hc, int wc) {
    __shared__ LbCenters Ls;
    if (threadIdx.x == 0) Ls = lb_centers(s);
    __syncthreads();
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), r = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (c >= wc || r >= hc) return;
    const long long i = (long long)r * wc + c;
    const unsigned* u = reinterpret_cast<const unsigned*>(s);
    float a = m1[i], b = m2[i];
    float ovl = rintf(__fmul_rn(a, b));
    float ref_only = __fsub_rn(a, ovl);
    float om = 0.f;
    if (ovl != 0.f) {
        const LbCenters L = Ls;
        float pmin = key2f(u[12]), pmax = key2f(u[13]);
        om = __fsub_rn(lb_proj(L, r, c), pmin) / __fadd_rn(__fsub_rn(pmax, pmin), 1e-3f);
    }
    X[i] = __fadd_rn(ref_only, __fmul_rn(__fsub_rn(1.f, om), a));
}

struct Gauss21 {
    float k[21];
};

__device__ __forceinline__ int reflect_idx(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}

// separable 21-tap Gaussian, reflect border (torchvision GaussianBlur((21,21), 20))
__global__ void lb_blur_kernel(const float* __restrict__ in, float* __restrict__ out, int hc, int wc, int vertical,
                               Gauss21 g) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), r = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (c >= wc || r >= hc) return;
    const long long i = (long long)r * wc + c;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 21; ++k) {
        int rr = vertical ? reflect_idx(r + k - 10, hc) : r;
        int cc = vertical ? c : reflect_idx(c + k - 10, wc);
        acc = fmaf(g.k[k], in[(long long)rr * wc + cc], acc);
    }
    out[i] = acc;
}

__global__ void lb_final_kernel(const float* __restrict__ ref, const float* __restrict__ tgt,
                                const float* __restrict__ m1, const float* __restrict__ m2,
                                const float* __restrict__ blur, float* __restrict__ out, float* __restrict__ mask1_out,
                                int hc, int wc) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), r = blockIdx.y * 4 + (threadIdx.x >> 6);
    const long long n = (long long)hc * wc;
    if (c >= wc || r >= hc) return;
    const long long i = (long long)r * wc + c;
    float a = m1[i], b = m2[i];
    float ovl = rintf(__fmul_rn(a, b));
    float ref_only = __fsub_rn(a, ovl);
    float mk = fminf(fmaxf(__fadd_rn(__fmul_rn(blur[i], a), ref_only), 0.f), 1.f);
    if (mask1_out) mask1_out[i] = mk;
    if (out) {
        float mk2 = __fmul_rn(__fsub_rn(1.f, mk), b);
#pragma unroll
        for (int ch = 0; ch < 3; ++ch)
            out[ch * n + i] = __fadd_rn(__fmul_rn(ref[ch * n + i], mk), __fmul_rn(tgt[ch * n + i], mk2));
    }
}

extern "C" long long ss_linear_blend_workspace_floats(int hc, int wc) { return 32 + 2ll * hc * wc; }

extern "C" int ss_linear_blend(const float* ref, const float* tgt, const float* ref_m, const float* tgt_m, float* out,
                               float* mask1_out, int hc, int wc, float* ws, void* stream) {
    if (!ref_m || !tgt_m || !ws || (!out && !mask1_out) || (out && (!ref || !tgt)) || hc < 11 || wc < 11)
        return SS_ERR_ARG;
    // 1-D kernel exp(-0.5 (t/sigma)^2) on linspace(-10,10,21), normalised, fp32 like torchvision
    Gauss21 g;
    {
        float s = 0.f;
        for (int i = 0; i < 21; ++i) {
            float t = (float)(i - 10) / 20.0f;
            g.k[i] = expf(-0.5f * (t * t));
            s += g.k[i];
        }
        for (int i = 0; i < 21; ++i) g.k[i] /= s;
    }
    hipStream_t st = (hipStream_t)stream;
    unsigned long long* sc = reinterpret_cast<unsigned long long*>(ws);
    float* X = ws + 32;
    float* tmp = X + (long long)hc * wc;
    hipLaunchKernelGGL(lb_init_kernel, dim3(1), dim3(64), 0, st, sc);
    // few, long-running workgroups for the two reductions: every workgroup ends in atomics on the same scalars
    dim3 rg(ss_cdiv(wc, 64), ss_cdiv(hc, 16) < 8 ? ss_cdiv(hc, 16) : 8);
    hipLaunchKernelGGL(lb_centroid_kernel, rg, dim3(256), 0, st, ref_m, tgt_m, sc, hc, wc);
    hipLaunchKernelGGL(lb_range_kernel, rg, dim3(256), 0, st, ref_m, tgt_m, sc, hc, wc);
    const dim3 pg(ss_cdiv(wc, 64), ss_cdiv(hc, 4));
    hipLaunchKernelGGL(lb_premask_kernel, pg, dim3(256), 0, st, ref_m, tgt_m, (const unsigned long long*)sc, X, hc, wc);
    hipLaunchKernelGGL(lb_blur_kernel, pg, dim3(256), 0, st, (const float*)X, tmp, hc, wc, 0, g);
    hipLaunchKernelGGL(lb_blur_kernel, pg, dim3(256), 0, st, (const float*)tmp, X, hc, wc, 1, g);
    hipLaunchKernelGGL(lb_final_kernel, pg, dim3(256), 0, st, ref, tgt, ref_m, tgt_m, (const float*)X, out,
                       mask1_out, hc, wc);
    return ss_launch_status();
}

// ------------------------------------------------------------------------------------------------
// LINEAR fusion of a whole clip (round 4): the per-frame chain above -- one warp launch + seven blender launches per
// frame, 256 launches per 32-frame clip -- as THREE launches per clip (four with three views: a second blend pass), bit-identical to it:
//   A  lb_clip_warp_kernel    every view of every frame warped once (colour planes + ones-mask plane, the layout of
//                             ss_tps_warp_views), all views of a tile in ONE workgroup so that the blender's whole-canvas
//                             statistics fall out of the masks while they are still in registers: per wave and pass the six
//                             centroid sums (ballot + popcount on the scalar unit: count, sum of rows, sum of columns of the
//                             non-zero mask pixels) and, per row, the first and last overlap column of the tile;
//   R  lb_clip_reduce_kernel  per (frame, pass): the wave partials -> mask centroids -> projection range of the overlap.
//                             lb_proj is weakly monotone along a row (fp32 rounding preserves order), so its extremes over
//                             the overlap are attained at the first / last overlap column of some (row, tile) segment:
//                             the range needs those candidates only, not another pass over the masks;
//   C  lb_clip_blend_kernel   per 64 x 64 canvas tile: X = ref_only + (1 - ovl_mask) m1 on the tile + 10-pixel reflect halo
//                             -> LDS, horizontal 21-tap pass -> LDS, vertical pass, mask1, blend, store (fp32 planes or the
//                             uint8 video frame).  X and the horizontally blurred plane never exist in memory.
// Three views chain ((1 (+) 2) (+) 3) like test_online_tra_threeview.py:489-502: pass 2 takes the union mask
// m1 + m2 - m1 m2 (recomputed on the fly from the two planes) and the fused planes of pass 1.
//   ws: W [n][V][4][hc][wc] | F [n][3][hc][wc] (V = 3 only) | scalars [n][P][16] u64 | partials [n][tiles][4 waves][P][8] u32
//   scalars as in ss_linear_blend: words 0..5 = cnt1, sumr1, sumc1, cnt2, sumr2, sumc2; u32[12] / u32[13] = range keys
#define LBC_PW 8          // partial words per (wave, pass): k1, r1, c1, k2, r2, c2, candidates of row a, row b

extern "C" long long ss_linear_clip_workspace_floats(int frames, int views, int hc, int wc) {
    if (frames <= 0 || (views != 2 && views != 3) || hc <= 1 || wc <= 1) return 0;
    const long long ohw = (long long)hc * wc, p = views - 1;
    const long long tiles = (long long)ss_cdiv(wc, 64) * ss_cdiv(hc, 8);
    // (+ 1: the 64-bit scalars block starts on an even float behind W / F whatever the parity of frames * hc * wc)
    return (long long)frames * (views * 4 * ohw + (views == 3 ? 3 * ohw : 0) + p * 32 + tiles * 4 * p * LBC_PW) + 1;
}

// sum of the lane indices whose bit is set (scalar unit: six popcounts)
__device__ __forceinline__ unsigned lane_index_sum(unsigned long long m) {
    return (unsigned)__popcll(m & 0xAAAAAAAAAAAAAAAAull) + 2u * (unsigned)__popcll(m & 0xCCCCCCCCCCCCCCCCull) +
           4u * (unsigned)__popcll(m & 0xF0F0F0F0F0F0F0F0ull) + 8u * (unsigned)__popcll(m & 0xFF00FF00FF00FF00ull) +
           16u * (unsigned)__popcll(m & 0xFFFF0000FFFF0000ull) + 32u * (unsigned)__popcll(m & 0xFFFFFFFF00000000ull);
}

template <int VIEWS, bool U8>
__global__ __launch_bounds__(256) void lb_clip_warp_kernel(RenderViews rv, const float* __restrict__ source,
                                                           const float* __restrict__ T, float* __restrict__ W,
                                                           unsigned* __restrict__ partials, int h, int w, int hc, int wc,
                                                           int mode, long long img_fs) {
    constexpr int P = VIEWS - 1;
    const long long frame = blockIdx.y;
    const long long hw = (long long)h * w, ohw = (long long)hc * wc;
    source += frame * (VIEWS * SS_NV * 2);
    T += frame * (VIEWS * 2 * SS_NT);
    W += frame * (VIEWS * 4) * ohw;
#pragma unroll
    for (int k = 0; k < VIEWS; ++k)
        rv.img[k] = reinterpret_cast<const float*>(reinterpret_cast<const char*>(rv.img[k]) + frame * img_fs);
    const int lx = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nbx = (wc + 63) / 64;
    const int tby = blockIdx.x / nbx, tbx = blockIdx.x - tby * nbx;
    unsigned* part = partials + ((frame * gridDim.x + blockIdx.x) * 4 + wv) * (P * LBC_PW);
    const int x = tbx * 64 + lx;
    const int ya = tby * 8 + wv, yb = ya + 4;
    if (ya >= hc) {                                  // a wave below the canvas: its partials are read all the same
        if (lx < P * LBC_PW) part[lx] = (lx & 7) >= 6 ? 0xFFFFFFFFu : 0u;
        return;
    }
    const bool xin = x < wc;
    const float gx = linspace_at(-1.f, 1.f, wc, min(x, wc - 1));
    __shared__ ss_f2 dytab[4][VIEWS][64];
    const float gya = linspace_at(-1.f, 1.f, hc, ya), gyb = linspace_at(-1.f, 1.f, hc, min(yb, hc - 1));
#pragma unroll
    for (int k = 0; k < VIEWS; ++k) tps_rows_table(source + k * SS_NV * 2, gya, gyb, lx, dytab[wv][k]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // same wave reads it back: ordering only, no barrier
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float ma[VIEWS], mb[VIEWS];                      // the ones-mask plane of every view at the lane's pixel of row a / row b
    const bool rowb = yb < hc;
#pragma unroll
    for (int k = 0; k < VIEWS; ++k) {
        ss_f2 px, py;
        tps_eval_rows(source + k * SS_NV * 2, T + k * 2 * SS_NT, dytab[wv][k], gx, gya, gyb, px, py);
        float va[3], vb[3];
        if (U8) {
            const unsigned char* img8 = reinterpret_cast<const unsigned char*>(rv.img[k]);
            sample3_u8(img8, px.x, py.x, w, h, mode, va);
            sample3_u8(img8, px.y, py.y, w, h, mode, vb);
        } else {
            sample3(rv.img[k], px.x, py.x, w, h, hw, mode, va);
            sample3(rv.img[k], px.y, py.y, w, h, hw, mode, vb);
        }
        if (mode == SS_WARP_NORMAL) {
            const SsTaps ta = taps_normal(px.x, py.x, w, h), tb = taps_normal(px.y, py.y, w, h);
            ma[k] = blend4(ta, 1.f, 1.f, 1.f, 1.f);
            mb[k] = blend4(tb, 1.f, 1.f, 1.f, 1.f);
        } else {
            ma[k] = fast_mask(px.x, py.x, w, h);
            mb[k] = fast_mask(px.y, py.y, w, h);
        }
        if (xin) {
            float* o = W + (long long)k * 4 * ohw + (long long)ya * wc + x;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) o[ch * ohw] = va[ch];
            o[3 * ohw] = ma[k];
            if (rowb) {
                o += 4ll * wc;
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) o[ch * ohw] = vb[ch];
                o[3 * ohw] = mb[k];
            }
        }
    }
    // the blender's statistics of this wave's two rows, per pass: ref mask = m0 (pass 1) / m0 + m1 - m0 m1 (pass 2)
    float ra = ma[0], rb = mb[0];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const float ta = ma[p + 1], tb = mb[p + 1];
        const unsigned long long r_a = __ballot(xin && ra != 0.f), r_b = __ballot(xin && rowb && rb != 0.f);
        const unsigned long long t_a = __ballot(xin && ta != 0.f), t_b = __ballot(xin && rowb && tb != 0.f);
        const unsigned long long o_a = __ballot(xin && lb_overlap(ra, ta)), o_b = __ballot(xin && rowb && lb_overlap(rb, tb));
        const unsigned k1a = (unsigned)__popcll(r_a), k1b = (unsigned)__popcll(r_b);
        const unsigned k2a = (unsigned)__popcll(t_a), k2b = (unsigned)__popcll(t_b);
        if (lx == 0) {
            unsigned* q = part + p * LBC_PW;
            q[0] = k1a + k1b;
            q[1] = k1a * (unsigned)ya + k1b * (unsigned)yb;
            q[2] = (k1a + k1b) * (unsigned)(tbx * 64) + lane_index_sum(r_a) + lane_index_sum(r_b);
            q[3] = k2a + k2b;
            q[4] = k2a * (unsigned)ya + k2b * (unsigned)yb;
            q[5] = (k2a + k2b) * (unsigned)(tbx * 64) + lane_index_sum(t_a) + lane_index_sum(t_b);
            // first | last << 16 overlap column of the row inside this tile (0xFFFFFFFF: none)
            q[6] = o_a ? (unsigned)(tbx * 64 + __ffsll((long long)o_a) - 1) | ((unsigned)(tbx * 64 + 63 - __clzll((long long)o_a)) << 16)
                       : 0xFFFFFFFFu;
            q[7] = o_b ? (unsigned)(tbx * 64 + __ffsll((long long)o_b) - 1) | ((unsigned)(tbx * 64 + 63 - __clzll((long long)o_b)) << 16)
                       : 0xFFFFFFFFu;
        }
        if (p + 1 < P) {                              // mask12 = m1 + m2 - m1 m2 (threeview:498; ss_mask_union)
            ra = __fsub_rn(__fadd_rn(ra, ta), __fmul_rn(ra, ta));
            rb = __fsub_rn(__fadd_rn(rb, tb), __fmul_rn(rb, tb));
        }
    }
}

// one workgroup per (frame, pass): wave partials -> scalars block
__global__ __launch_bounds__(256) void lb_clip_reduce_kernel(const unsigned* __restrict__ partials, unsigned long long* __restrict__ scalars,
                                                             int tiles, int nbx, int passes) {
    const int frame = blockIdx.x / passes, p = blockIdx.x - frame * passes;
    const unsigned* q0 = partials + ((long long)frame * tiles * 4) * (passes * LBC_PW) + p * LBC_PW;
    const long long stride = (long long)passes * LBC_PW;
    const int nw = tiles * 4;                        // wave partials of this frame
    unsigned long long* s = scalars + (long long)blockIdx.x * 16;
    __shared__ unsigned long long red[6];
    __shared__ LbCenters Ls;
    __shared__ float smn[4], smx[4];
    if (threadIdx.x < 6) red[threadIdx.x] = 0ull;
    __syncthreads();
    unsigned long long acc[6] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
    for (int i = threadIdx.x; i < nw; i += 256) {
        const unsigned* q = q0 + i * stride;
#pragma unroll
        for (int k = 0; k < 6; ++k) acc[k] += q[k];
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        unsigned long long t = acc[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
        if ((threadIdx.x & 63) == 0) atomicAdd(&red[k], t);
    }
    __syncthreads();
    if (threadIdx.x < 6) s[threadIdx.x] = red[threadIdx.x];
    if (threadIdx.x == 0) Ls = lb_centers(red);
    __syncthreads();
    const LbCenters L = Ls;
    float mn = INFINITY, mx = -INFINITY;
    // wave partial i belongs to tile i / 4, wave i % 4: rows (tile / nbx) * 8 + wave and + 4 -- the row is recomputed from
    // the partial's position, the columns come packed
    for (int i = threadIdx.x; i < nw; i += 256) {
        const unsigned* q = q0 + i * stride;
        const int tile = i >> 2, wv = i & 3;
        const int ya = (tile / nbx) * 8 + wv;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const unsigned e = q[6 + rr];
            if (e != 0xFFFFFFFFu) {
                const int r = ya + 4 * rr;
                const float p0 = lb_proj(L, r, (int)(e & 0xFFFFu)), p1 = lb_proj(L, r, (int)(e >> 16));
                mn = fminf(mn, fminf(p0, p1));
                mx = fmaxf(mx, fmaxf(p0, p1));
            }
        }
    }
    mn = ss_wave_min(mn);
    mx = ss_wave_max(mx);
    if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        mn = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
        mx = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
        unsigned* u = reinterpret_cast<unsigned*>(s);
        u[12] = mn != INFINITY ? f2key(mn) : 0xffffffffu;         // (the values lb_init_kernel + atomicMin / atomicMax leave)
        u[13] = mx != -INFINITY ? f2key(mx) : 0u;
    }
}

struct LbClipArgs {
    const float *ref, *tgt, *m1a, *m1b, *m2;          // m1b != nullptr: ref mask = m1a + m1b - m1a m1b
    long long ref_fs, tgt_fs, m_fs;                   // frame strides in floats (all masks share W's)
    float* out;                                       // fp32 [n][3][hc][wc] or uint8 [n][hc][wc][3]
    float* mask1_out;                                 // nullable, this pass's plane of [n][P][hc][wc]
    long long mask1_fs;
    const unsigned long long* scalars;                // this pass's block of frame 0
    long long sc_fs;                                  // u64 words between frames
};

#define LBC_T 64
#define LBC_E (LBC_T + 20)
template <bool UNION, bool U8OUT>
__global__ __launch_bounds__(256) void lb_clip_blend_kernel(LbClipArgs a, int hc, int wc, Gauss21 g) {
    __shared__ float XL[LBC_E * LBC_E];              // X on the tile + halo, reflected coordinates resolved
    __shared__ float TL[LBC_E * LBC_T];              // horizontally blurred rows of the tile + vertical halo
    __shared__ LbCenters Ls;
    __shared__ float prange[2];
    const long long frame = blockIdx.z;
    const long long ohw = (long long)hc * wc;
    const float* m1a = a.m1a + frame * a.m_fs;
    const float* m1b = UNION ? a.m1b + frame * a.m_fs : nullptr;
    const float* m2 = a.m2 + frame * a.m_fs;
    const unsigned long long* s = a.scalars + frame * a.sc_fs;
    if (threadIdx.x == 0) {
        Ls = lb_centers(s);
        const unsigned* u = reinterpret_cast<const unsigned*>(s);
        prange[0] = key2f(u[12]);
        prange[1] = key2f(u[13]);
    }
    __syncthreads();
    const LbCenters L = Ls;
    const float pmin = prange[0], pmax = prange[1];
    const int r0 = blockIdx.y * LBC_T, c0 = blockIdx.x * LBC_T;
    for (int idx = threadIdx.x; idx < LBC_E * LBC_E; idx += 256) {
        const int row = idx / LBC_E, col = idx - row * LBC_E;
        // rows / columns of a partial last tile beyond the canvas feed no stored pixel: any in-range index will do
        const int r = min(max(reflect_idx(r0 - 10 + row, hc), 0), hc - 1), c = min(max(reflect_idx(c0 - 10 + col, wc), 0), wc - 1);
        const long long i = (long long)r * wc + c;
        float ma = m1a[i];
        if (UNION) { const float mb = m1b[i]; ma = __fsub_rn(__fadd_rn(ma, mb), __fmul_rn(ma, mb)); }
        const float b = m2[i];
        const float ovl = rintf(__fmul_rn(ma, b));
        const float ref_only = __fsub_rn(ma, ovl);
        float om = 0.f;
        if (ovl != 0.f) om = __fsub_rn(lb_proj(L, r, c), pmin) / __fadd_rn(__fsub_rn(pmax, pmin), 1e-3f);
        XL[idx] = __fadd_rn(ref_only, __fmul_rn(__fsub_rn(1.f, om), ma));
    }
    __syncthreads();
    const int lx = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int row = wv; row < LBC_E; row += 4) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 21; ++k) acc = fmaf(g.k[k], XL[row * LBC_E + lx + k], acc);
        TL[row * LBC_T + lx] = acc;
    }
    __syncthreads();
    const int c = c0 + lx;
    if (c >= wc) return;
    const float* ref = a.ref + frame * a.ref_fs;
    const float* tgt = a.tgt + frame * a.tgt_fs;
    float* mk_out = a.mask1_out ? a.mask1_out + frame * a.mask1_fs : nullptr;
    for (int row = wv; row < LBC_T; row += 4) {
        const int r = r0 + row;
        if (r >= hc) break;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 21; ++k) acc = fmaf(g.k[k], TL[(row + k) * LBC_T + lx], acc);
        const long long i = (long long)r * wc + c;
        float ma = m1a[i];
        if (UNION) { const float mb = m1b[i]; ma = __fsub_rn(__fadd_rn(ma, mb), __fmul_rn(ma, mb)); }
        const float b = m2[i];
        const float ovl = rintf(__fmul_rn(ma, b));
        const float ref_only = __fsub_rn(ma, ovl);
        const float mk = fminf(fmaxf(__fadd_rn(__fmul_rn(acc, ma), ref_only), 0.f), 1.f);
        if (mk_out) mk_out[i] = mk;
        const float mk2 = __fmul_rn(__fsub_rn(1.f, mk), b);
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const float v = __fadd_rn(__fmul_rn(ref[ch * ohw + i], mk), __fmul_rn(tgt[ch * ohw + i], mk2));
            if (U8OUT) reinterpret_cast<unsigned char*>(a.out)[(frame * ohw + i) * 3 + ch] = render_to_u8(v);
            else a.out[(frame * 3 + ch) * ohw + i] = v;
        }
    }
}

// The same blend as a ROLLING pass down the canvas (round 4, second version; the 64 x 64-tile kernel above stays as the
// reference form and for the tests).  One WAVE owns a strip of 64 columns x `rs` rows of one frame and walks it row by row:
//   row p:  X(p, c0-10 .. c0+73) from the masks (reflected columns; lanes 0..19 take a second column) -> 84 floats of LDS
//           -> the lane's horizontal 21-tap sum hb(p) -> pushed into a 21-deep ring held in REGISTERS;
//   once the ring is full, output row p-10 = the vertical 21-tap sum over the ring, mask1, blend, store.
// No workgroup barrier, no tile halo in the row direction (only the 20 warm-up rows per strip), X / hb never in memory; the
// loads of the next row are issued before the current row's arithmetic.  Same fmaf chains as the tile kernel and as
// lb_blur_kernel: bit-identical output.  The tile kernel moved 2.2 GB per 32-frame clip at 2.7 TB/s (three phases between
// barriers, 50 KB of LDS -> 3 workgroups per CU: its loads came in bursts); this one streams.
template <bool UNION, bool U8OUT>
__global__ __launch_bounds__(64) void lb_clip_blend_rows_kernel(LbClipArgs a, int hc, int wc, int rs, Gauss21 g) {
    __shared__ float XL[96];
    __shared__ LbCenters Ls;
    __shared__ float prange[2];
    const long long frame = blockIdx.z;
    const long long ohw = (long long)hc * wc;
    const float* __restrict__ m1a = a.m1a + frame * a.m_fs;
    const float* __restrict__ m1b = UNION ? a.m1b + frame * a.m_fs : nullptr;
    const float* __restrict__ m2 = a.m2 + frame * a.m_fs;
    const float* __restrict__ ref = a.ref + frame * a.ref_fs;
    const float* __restrict__ tgt = a.tgt + frame * a.tgt_fs;
    float* mk_out = a.mask1_out ? a.mask1_out + frame * a.mask1_fs : nullptr;
    const unsigned long long* s = a.scalars + frame * a.sc_fs;
    const int lane = threadIdx.x;
    if (lane == 0) {
        Ls = lb_centers(s);
        const unsigned* u = reinterpret_cast<const unsigned*>(s);
        prange[0] = key2f(u[12]);
        prange[1] = key2f(u[13]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const LbCenters L = Ls;
    const float pmin = prange[0], pden = __fadd_rn(__fsub_rn(prange[1], prange[0]), 1e-3f);
    const int c0 = blockIdx.x * 64, s0 = blockIdx.y * rs, s1 = min(s0 + rs, hc);
    const int c = c0 + lane;
    const bool cin = c < wc, second = lane < 20;
    const int ca = min(max(reflect_idx(c0 - 10 + lane, wc), 0), wc - 1);
    const int cb = min(max(reflect_idx(c0 + 54 + lane, wc), 0), wc - 1);
    const int cc = min(c, wc - 1);
    auto xval = [&](float ma, float mb, float b, int r, int col) -> float {
        if (UNION) ma = __fsub_rn(__fadd_rn(ma, mb), __fmul_rn(ma, mb));
        const float ovl = rintf(__fmul_rn(ma, b));
        const float ref_only = __fsub_rn(ma, ovl);
        float om = 0.f;
        if (ovl != 0.f) om = __fsub_rn(lb_proj(L, r, col), pmin) / pden;
        return __fadd_rn(ref_only, __fmul_rn(__fsub_rn(1.f, om), ma));
    };
    float ring[21];
#pragma unroll
    for (int k = 0; k < 21; ++k) ring[k] = 0.f;
    // masks of the row whose X comes next (columns ca / cb), requested one row ahead
    float nxa1, nxa1b = 0.f, nxa2, nxb1 = 0.f, nxb1b = 0.f, nxb2 = 0.f;
    auto row_of = [&](int p) { return min(max(reflect_idx(p, hc), 0), hc - 1); };
    {
        const long long i = (long long)row_of(s0 - 10) * wc;
        nxa1 = m1a[i + ca]; nxa2 = m2[i + ca];
        if (UNION) nxa1b = m1b[i + ca];
        if (second) { nxb1 = m1a[i + cb]; nxb2 = m2[i + cb]; if (UNION) nxb1b = m1b[i + cb]; }
    }
    for (int p = s0 - 10; p < s1 + 10; ++p) {
        const int r = row_of(p);
        const float xa1 = nxa1, xa1b = nxa1b, xa2 = nxa2, xb1 = nxb1, xb1b = nxb1b, xb2 = nxb2;
        if (p + 1 < s1 + 10) {                               // next row's masks
            const long long i = (long long)row_of(p + 1) * wc;
            nxa1 = m1a[i + ca]; nxa2 = m2[i + ca];
            if (UNION) nxa1b = m1b[i + ca];
            if (second) { nxb1 = m1a[i + cb]; nxb2 = m2[i + cb]; if (UNION) nxb1b = m1b[i + cb]; }
        }
        // this output row's planes (row p - 10), requested before the blur arithmetic
        const int ro = p - 10;
        const bool emit = ro >= s0;
        float oma = 0.f, omb = 0.f, ob = 0.f, orf[3] = {0.f, 0.f, 0.f}, otg[3] = {0.f, 0.f, 0.f};
        const long long io = (long long)max(ro, 0) * wc + cc;
        if (emit) {
            oma = m1a[io]; ob = m2[io];
            if (UNION) omb = m1b[io];
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) { orf[ch] = ref[ch * ohw + io]; otg[ch] = tgt[ch * ohw + io]; }
        }
        XL[lane] = xval(xa1, xa1b, xa2, r, ca);
        if (second) XL[64 + lane] = xval(xb1, xb1b, xb2, r, cb);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float hb = 0.f;
#pragma unroll
        for (int k = 0; k < 21; ++k) hb = fmaf(g.k[k], XL[lane + k], hb);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the row buffer is rewritten next iteration
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int k = 0; k < 20; ++k) ring[k] = ring[k + 1];
        ring[20] = hb;
        if (emit && cin) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 21; ++k) acc = fmaf(g.k[k], ring[k], acc);
            float ma = oma;
            if (UNION) ma = __fsub_rn(__fadd_rn(ma, omb), __fmul_rn(ma, omb));
            const float ovl = rintf(__fmul_rn(ma, ob));
            const float ref_only = __fsub_rn(ma, ovl);
            const float mk = fminf(fmaxf(__fadd_rn(__fmul_rn(acc, ma), ref_only), 0.f), 1.f);
            if (mk_out) mk_out[io] = mk;
            const float mk2 = __fmul_rn(__fsub_rn(1.f, mk), ob);
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const float v = __fadd_rn(__fmul_rn(orf[ch], mk), __fmul_rn(otg[ch], mk2));
                if (U8OUT) reinterpret_cast<unsigned char*>(a.out)[(frame * ohw + io) * 3 + ch] = render_to_u8(v);
                else a.out[(frame * 3 + ch) * ohw + io] = v;
            }
        }
    }
}

static Gauss21 lb_gauss() {
    // 1-D kernel exp(-0.5 (t/sigma)^2) on linspace(-10,10,21), normalised, fp32 like torchvision (as ss_linear_blend)
    Gauss21 g;
    float s = 0.f;
    for (int i = 0; i < 21; ++i) {
        float t = (float)(i - 10) / 20.0f;
        g.k[i] = expf(-0.5f * (t * t));
        s += g.k[i];
    }
    for (int i = 0; i < 21; ++i) g.k[i] /= s;
    return g;
}

// rows per strip of the rolling blend kernel: 0 = default (96), > 0 = that many, < 0 = the 64 x 64-tile kernel instead
// (ss_linear_clip_set_rows: an A/B and test knob; both forms give the same bits)
static int g_lb_rows = 0;
extern "C" int ss_linear_clip_set_rows(int rows) {
    g_lb_rows = rows;
    return SS_OK;
}

static int render_linear_clip_launch(const void* const* views_base, const float* source, const float* T, void* out,
                                     float* mask1_out, int frames, int views, int h, int w, int hc, int wc, int mode,
                                     float* ws, void* stream, bool u8) {
    if (!views_base || !source || !T || !out || !ws || frames <= 0 || frames > 65535 || (views != 2 && views != 3) || h <= 1 ||
        w <= 1 || hc < 11 || wc < 11 || wc > 65535 || hc > 65535 || (mode != SS_WARP_NORMAL && mode != SS_WARP_FAST))
        return SS_ERR_ARG;
    RenderViews rv;
    for (int i = 0; i < 3; ++i) rv.img[i] = i < views ? static_cast<const float*>(views_base[i]) : nullptr;
    for (int i = 0; i < views; ++i)
        if (!rv.img[i]) return SS_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const long long ohw = (long long)hc * wc;
    const int P = views - 1, nbx = ss_cdiv(wc, 64), tiles = nbx * ss_cdiv(hc, 8);
    float* W = ws;
    float* F = W + (long long)frames * views * 4 * ohw;
    long long sc_off = (long long)frames * views * 4 * ohw + (views == 3 ? (long long)frames * 3 * ohw : 0);
    sc_off += sc_off & 1;                                  // 8-byte alignment of the u64 block by construction (ws is 8-byte aligned)
    if (reinterpret_cast<unsigned long long>(ws) & 7ull) return SS_ERR_ARG;
    unsigned long long* scalars = reinterpret_cast<unsigned long long*>(ws + sc_off);
    unsigned* partials = reinterpret_cast<unsigned*>(scalars + (long long)frames * P * 16);
    const long long img_fs = u8 ? 3ll * h * w : 12ll * h * w;
    const dim3 ga(tiles, frames);
    if (views == 2) {
        if (u8) hipLaunchKernelGGL((lb_clip_warp_kernel<2, true>), ga, dim3(256), 0, st, rv, source, T, W, partials, h, w, hc, wc, mode, img_fs);
        else hipLaunchKernelGGL((lb_clip_warp_kernel<2, false>), ga, dim3(256), 0, st, rv, source, T, W, partials, h, w, hc, wc, mode, img_fs);
    } else {
        if (u8) hipLaunchKernelGGL((lb_clip_warp_kernel<3, true>), ga, dim3(256), 0, st, rv, source, T, W, partials, h, w, hc, wc, mode, img_fs);
        else hipLaunchKernelGGL((lb_clip_warp_kernel<3, false>), ga, dim3(256), 0, st, rv, source, T, W, partials, h, w, hc, wc, mode, img_fs);
    }
    hipLaunchKernelGGL(lb_clip_reduce_kernel, dim3(frames * P), dim3(256), 0, st, (const unsigned*)partials, scalars, tiles, nbx, P);
    const Gauss21 g = lb_gauss();
    const dim3 gc(ss_cdiv(wc, LBC_T), ss_cdiv(hc, LBC_T), frames);
    // rolling form: strips of 64 columns x rs rows, one wave each; rs so that a 720p canvas gives ~30 waves per CU
    const int rs = g_lb_rows > 0 ? g_lb_rows : 96;
    const dim3 gr(ss_cdiv(wc, 64), ss_cdiv(hc, rs), frames);
    const bool rolling = g_lb_rows >= 0;
    LbClipArgs a;
    a.m_fs = views * 4 * ohw;
    a.ref = W; a.ref_fs = a.m_fs;
    a.tgt = W + 4 * ohw; a.tgt_fs = a.m_fs;
    a.m1a = W + 3 * ohw; a.m1b = nullptr; a.m2 = W + 7 * ohw;
    a.mask1_out = mask1_out; a.mask1_fs = P * ohw;
    a.scalars = scalars; a.sc_fs = P * 16;
    if (views == 2) {
        a.out = static_cast<float*>(out);
        if (rolling) {
            if (u8) hipLaunchKernelGGL((lb_clip_blend_rows_kernel<false, true>), gr, dim3(64), 0, st, a, hc, wc, rs, g);
            else hipLaunchKernelGGL((lb_clip_blend_rows_kernel<false, false>), gr, dim3(64), 0, st, a, hc, wc, rs, g);
        } else {
            if (u8) hipLaunchKernelGGL((lb_clip_blend_kernel<false, true>), gc, dim3(256), 0, st, a, hc, wc, g);
            else hipLaunchKernelGGL((lb_clip_blend_kernel<false, false>), gc, dim3(256), 0, st, a, hc, wc, g);
        }
    } else {
        a.out = F;
        if (rolling) hipLaunchKernelGGL((lb_clip_blend_rows_kernel<false, false>), gr, dim3(64), 0, st, a, hc, wc, rs, g);
        else hipLaunchKernelGGL((lb_clip_blend_kernel<false, false>), gc, dim3(256), 0, st, a, hc, wc, g);
        a.ref = F; a.ref_fs = 3 * ohw;
        a.tgt = W + 8 * ohw;
        a.m1b = W + 7 * ohw; a.m2 = W + 11 * ohw;
        a.mask1_out = mask1_out ? mask1_out + ohw : nullptr;
        a.scalars = scalars + 16;
        a.out = static_cast<float*>(out);
        if (rolling) {
            if (u8) hipLaunchKernelGGL((lb_clip_blend_rows_kernel<true, true>), gr, dim3(64), 0, st, a, hc, wc, rs, g);
            else hipLaunchKernelGGL((lb_clip_blend_rows_kernel<true, false>), gr, dim3(64), 0, st, a, hc, wc, rs, g);
        } else {
            if (u8) hipLaunchKernelGGL((lb_clip_blend_kernel<true, true>), gc, dim3(256), 0, st, a, hc, wc, g);
            else hipLaunchKernelGGL((lb_clip_blend_kernel<true, false>), gc, dim3(256), 0, st, a, hc, wc, g);
        }
    }
    return ss_launch_status();
}

// whole clip, LINEAR fusion: view k's frame f at views_base[k] + f * 3 h w floats (planar fp32 [n,3,h,w]) -> out [n,3,hc,wc];
// mask1_out (nullable): the blender's mask1 of every pass [n][V-1][hc][wc]; ws: ss_linear_clip_workspace_floats(...) floats
extern "C" int ss_render_linear_clip(const float* const* views_base, const float* source, const float* T, float* out,
                                     float* mask1_out, int frames, int views, int h, int w, int hc, int wc, int mode,
                                     float* ws, void* stream) {
    return render_linear_clip_launch(reinterpret_cast<const void* const*>(views_base), source, T, out, mask1_out, frames, views,
                                     h, w, hc, wc, mode, ws, stream, false);
}

// the same from decoded uint8 frames [n,h,w,3] per view to uint8 video frames [n,hc,wc,3] (`.astype(np.uint8)` of the blend)
extern "C" int ss_render_linear_clip_u8(const unsigned char* const* views_base, const float* source, const float* T,
                                        unsigned char* out, float* mask1_out, int frames, int views, int h, int w, int hc,
                                        int wc, int mode, float* ws, void* stream) {
    return render_linear_clip_launch(reinterpret_cast<const void* const*>(views_base), source, T, out, mask1_out, frames, views,
                                     h, w, hc, wc, mode, ws, stream, true);
}
