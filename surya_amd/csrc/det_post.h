// Heat-map -> text boxes ON THE DEVICE: replaces the D2H of [B, 2, H, W] fp32 maps (134 MB per 16 pages
// at 1024^2) and the per-page CPU post-processing of surya/detection/heatmap.py:14-107 (get_dynamic_thresholds, cv2
// connectedComponentsWithStats, per-component dilate + minAreaRect + boxPoints, confidence) with a handful of launches over
// all pages of a batch at once; only [n, 4, 2] corner arrays and confidences travel back.
//
// Integer / HBM-bound work, no GEMM shape anywhere: coalesced row-major sweeps over the maps (16-byte loads), histograms in
// LDS, label equivalences by atomicMin on a union-find forest, one wave per surviving component for the geometry.
//
//   1 thresholds   mean of the top 10 % by an exact 3-level radix select on the float bit patterns (11 + 11 + 10 bits; the maps
//                  are sigmoid outputs, i.e. positive floats whose bit patterns order like the values) + a float64 sum of the
//                  values above the selected rank; thresholds then follow heatmap.py:14-24 in float32
//   2 labels       mask = heat > low_text; union-find over 4-neighbours with the SMALLEST raster index as root (roots sorted
//                  by index = cv2's / scipy's raster label order); flatten
//   3 statistics   area, bounding box, max heat per root by atomics
//   4 select       roots with area >= 10 and max >= text_threshold, compacted IN RASTER ORDER (block counts -> scan ->
//                  scatter); per component a slice of the row-extreme arrays (prefix sum of dilated heights)
//   5 row extremes leftmost / rightmost pixel of every component row (atomicMin / Max into the slice)
//   6 geometry     one wave per component: dilated row extremes, convex hull, rotating calipers in float64, corner order
//                  (det_post_core.h -- the same code the CPU tests run), confidence = max / page max
#pragma once
#include "common.h"
#include "det_post_core.h"

namespace sa {
namespace post {

struct PageState {             // one per page, device resident
    unsigned int rank;         // remaining ascending rank of the element searched by the radix select
    unsigned int prefix;       // bit pattern selected so far
    unsigned int cnt_gt;       // elements strictly above the selected value
    int n_sel;                 // surviving components
    int n_rows;                // dilated rows of all surviving components
    int overflow;              // 1: more than max_boxes components or row capacity exceeded
    float text_thr, low_thr, max_conf;
    float pad_;
    double sum_gt;
};

struct PostArgs {
    const float* heat;         // plane 0 of page b at heat + b * page_stride
    long page_stride;
    int B, H, W, max_boxes, row_cap;
    float text_threshold, low_text;
    PageState* st;             // [B]
    unsigned int* hist;        // [B][2048]
    double* psum;              // [B][1024] per-block partial sums / counts of the values above the selected rank
    unsigned int* pcnt;
    int sweep_blocks;
    int* label;                // [B][N]
    int* area;                 // [B][N]   (after select: compact index of a root, -1 if dropped)
    int* minx; int* maxx; int* miny; int* maxy;   // [B][N], valid at roots
    unsigned int* maxv;        // [B][N] float bits, valid at roots
    int2* blk;                 // [B][n_blk] (selected roots, dilated rows) per scan block, then exclusive prefix
    int n_blk;
    CompStats* comp;           // [B][max_boxes]
    int* comp_rowoff;          // [B][max_boxes] offset of the component's slice in rmin / rmax
    int* rmin; int* rmax;      // [B][row_cap]
    float* boxes;              // [B][max_boxes][8]
    float* conf;               // [B][max_boxes]
    int* count;                // [B]
};

constexpr int SCAN_PIX = 4096;      // pixels per scan block (256 threads x 16)

// ------------------------------------------------------------------------------------------------------ 1. thresholds
template <int LEVEL>             // 0: bits [31:21], 1: bits [20:10] under the selected prefix, 2: bits [9:0]
__global__ __launch_bounds__(256) void post_hist_kernel(PostArgs p) {
    __shared__ unsigned int h[2048];
    const int b = blockIdx.y, N = p.H * p.W;
    for (int i = threadIdx.x; i < 2048; i += 256) h[i] = 0;
    __syncthreads();
    const unsigned int prefix = p.st[b].prefix;
    const float* src = p.heat + (long)b * p.page_stride;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < N; i += (long)gridDim.x * 1024) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned int u = __float_as_uint(v[e]);
            if (LEVEL == 0) atomicAdd(&h[u >> 21], 1u);
            else if (LEVEL == 1) { if ((u >> 21) == (prefix >> 21)) atomicAdd(&h[(u >> 10) & 0x7ffu], 1u); }
            else { if ((u >> 10) == (prefix >> 10)) atomicAdd(&h[u & 0x3ffu], 1u); }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 256)
        if (h[i]) atomicAdd(&p.hist[b * 2048 + i], h[i]);
}

// one workgroup per page: find the bin holding the searched rank, update (prefix, rank), clear the histogram
template <int LEVEL>
__global__ __launch_bounds__(256) void post_pick_kernel(PostArgs p) {
    __shared__ unsigned int part[256];
    const int b = blockIdx.x, t = threadIdx.x;
    unsigned int* h = p.hist + b * 2048;
    if (LEVEL == 0 && t == 0) {
        const int N = p.H * p.W;
        p.st[b].rank = (unsigned int)((double)N * 0.9);          // k = int(len * 0.9): ascending rank of the smallest top value
        p.st[b].prefix = 0; p.st[b].cnt_gt = 0; p.st[b].sum_gt = 0.0; p.st[b].n_sel = 0; p.st[b].n_rows = 0; p.st[b].overflow = 0;
        p.st[b].max_conf = 0.f;
    }
    __syncthreads();
    unsigned int loc[8], s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { loc[i] = h[t * 8 + i]; s += loc[i]; }
    part[t] = s;
    __syncthreads();
    if (t == 0) {
        unsigned int run = 0, rank = p.st[b].rank;
        int bin_t = 255;
        for (int i = 0; i < 256; ++i) {
            if (rank < run + part[i]) { bin_t = i; break; }
            run += part[i];
        }
        part[0] = (unsigned int)bin_t; part[1] = run;
    }
    __syncthreads();
    if (t == (int)part[0]) {
        unsigned int run = part[1], rank = p.st[b].rank;
        int bin = t * 8 + 7;
        for (int i = 0; i < 8; ++i) {
            if (rank < run + loc[i]) { bin = t * 8 + i; break; }
            run += loc[i];
        }
        p.st[b].rank = rank - run;
        const int shift = LEVEL == 0 ? 21 : (LEVEL == 1 ? 10 : 0);
        p.st[b].prefix |= (unsigned int)bin << shift;
    }
    __syncthreads();
    for (int i = t; i < 2048; i += 256) h[i] = 0;
}

// sum and count of the values strictly above the selected one (float64 accumulation)
__global__ __launch_bounds__(256) void post_topsum_kernel(PostArgs p) {
    __shared__ double sd[4];
    __shared__ unsigned int sc[4];
    const int b = blockIdx.y, N = p.H * p.W;
    const unsigned int vk = p.st[b].prefix;
    const float* src = p.heat + (long)b * p.page_stride;
    double s = 0.0;
    unsigned int c = 0;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < N; i += (long)gridDim.x * 1024) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + i);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (__float_as_uint(v[e]) > vk) { s += (double)v[e]; ++c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); c += __shfl_xor(c, o, 64); }
    if ((threadIdx.x & 63) == 0) { sd[threadIdx.x >> 6] = s; sc[threadIdx.x >> 6] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {          // per-block partials, summed in block order by post_thresholds_kernel: run-to-run identical
        p.psum[b * 1024 + blockIdx.x] = sd[0] + sd[1] + sd[2] + sd[3];
        p.pcnt[b * 1024 + blockIdx.x] = sc[0] + sc[1] + sc[2] + sc[3];
    }
}

__global__ void post_thresholds_kernel(PostArgs p) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= p.B) return;
    const int N = p.H * p.W, k = (int)((double)N * 0.9);
    const float vk = __uint_as_float(p.st[b].prefix);
    double sum = 0.0;
    unsigned int cnt = 0;
    for (int i = 0; i < p.sweep_blocks; ++i) { sum += p.psum[b * 1024 + i]; cnt += p.pcnt[b * 1024 + i]; }
    p.st[b].sum_gt = sum; p.st[b].cnt_gt = cnt;
    // np.mean of the N - k largest values (heatmap.py:17-18), then float32 arithmetic as numpy does with a float32 scalar
    const float avg = (float)((p.st[b].sum_gt + (double)(N - k - (int)p.st[b].cnt_gt) * (double)vk) / (double)(N - k));
    float sc = avg / 0.7f;
    sc = fminf(fmaxf(sc, 0.f), 1.f);
    sc = sqrtf(sc);
    p.st[b].text_thr = fminf(fmaxf(p.text_threshold * sc, 0.15f), 0.8f);
    p.st[b].low_thr = fminf(fmaxf(p.low_text * sc, 0.1f), 0.6f);
}

// ---------------------------------------------------------------------------------------------------------- 2. labels
// Parent reads during the concurrent merge bypass the (never refreshed) per-CU L1: another CU's atomicMin must become visible.
// A stale read is still a valid ancestor (parents only ever decrease), so this is about progress, not correctness.
__device__ __forceinline__ int uf_load(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int uf_find(const int* lab, int x) {
    int r = uf_load(lab + x);
    while (r != x) { x = r; r = uf_load(lab + x); }
    return x;
}
__device__ __forceinline__ void uf_union(int* lab, int a, int b) {
    // link the larger root under the smaller one; atomicMin makes concurrent links converge to the minimum index
    while (true) {
        a = uf_find(lab, a); b = uf_find(lab, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }           // a > b: a -> b
        const int old = atomicMin(&lab[a], b);
        if (old == a) return;
        a = old;                                                  // someone re-linked a meanwhile: merge its new parent with b
    }
}

// label = index of the first pixel of the pixel's horizontal run INSIDE its 64-lane wave segment (one ballot instead of up to
// 63 atomic links per run); runs that continue across a segment boundary are linked by post_merge_kernel. Statistics reset.
__global__ __launch_bounds__(256) void post_init_kernel(PostArgs p) {
    const int b = blockIdx.y, N = p.H * p.W;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float v = p.heat[(long)b * p.page_stride + i];
    const bool fg = v > p.st[b].low_thr;
    const int x = (int)(i % p.W), lane = threadIdx.x & 63;
    const unsigned long long fgm = __ballot(fg);
    const bool left_fg = lane > 0 && x > 0 && ((fgm >> (lane - 1)) & 1ull);
    const unsigned long long starts = __ballot(fg && !left_fg);
    int lab = -1;
    if (fg) {
        const unsigned long long upto = starts & (lane == 63 ? ~0ull : ((1ull << (lane + 1)) - 1ull));
        lab = (int)(i - lane + (63 - __builtin_clzll(upto)));       // a foreground lane always has a run start at or below it
    }
    const long o = (long)b * N + i;
    p.label[o] = lab;
    p.area[o] = 0; p.minx[o] = 0x7fffffff; p.maxx[o] = -1; p.miny[o] = 0x7fffffff; p.maxy[o] = -1; p.maxv[o] = 0;
}

__global__ __launch_bounds__(256) void post_merge_kernel(PostArgs p) {
    const int b = blockIdx.y, N = p.H * p.W;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    int* lab = p.label + (long)b * N;
    if (lab[i] < 0) return;                                     // the sign of a label never changes: plain read is fine
    const int x = (int)(i % p.W);
    // horizontal: only where a run crosses a wave-segment boundary (inside a segment post_init_kernel linked it)
    if ((threadIdx.x & 63) == 0 && x > 0 && lab[i - 1] >= 0) uf_union(lab, (int)i, (int)i - 1);
    // vertical; redundant when the left neighbours of both pixels are foreground (that thread makes the same connection)
    if (i >= p.W && lab[i - p.W] >= 0 && !(x > 0 && lab[i - 1] >= 0 && lab[i - p.W - 1] >= 0)) uf_union(lab, (int)i, (int)(i - p.W));
}

// ------------------------------------------------------------------------------------------------ 3. flatten + statistics
// A page's text mask is often a few LARGE components (on noise-like maps one component can hold half the page): per-pixel
// atomics on one root's six words serialise in L2 (measured 65 ms for a 16-page batch, r02h profile). Pixels are visited in raster
// order, so a wave usually sees ONE root: it reduces its 64 pixels with cross-lane ops, the workgroup's waves are merged in LDS
// when they agree, and one lane issues the six atomics -- 256x fewer on the hot addresses. Mixed waves fall back to per-lane.
struct WaveAgg { int root, cnt, minx, maxx, miny, maxy; unsigned int maxv; };

__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}

__global__ __launch_bounds__(256) void post_stats_kernel(PostArgs p) {
    __shared__ WaveAgg agg[4];
    const int b = blockIdx.y, N = p.H * p.W;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long base = (long)b * N;
    int* lab = p.label + base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool fg = i < N && lab[i] >= 0;
    int r = -1, x = 0, y = 0;
    unsigned int v = 0;
    if (fg) {
        r = uf_find(lab, (int)i);
        lab[i] = r;
        x = (int)(i % p.W); y = (int)(i / p.W);
        v = __float_as_uint(p.heat[(long)b * p.page_stride + i]);
    }
    const unsigned long long fgm = __ballot(fg);
    const int r0 = fgm ? __shfl(r, __builtin_ctzll(fgm), 64) : -1;
    const bool uniform = fgm && __ballot(fg && r != r0) == 0;
    if (uniform) {
        const int cnt = __builtin_popcountll(fgm);
        const int mnx = wave_min_i(fg ? x : 0x7fffffff), mxx = wave_max_i(fg ? x : -1);
        const int mny = wave_min_i(fg ? y : 0x7fffffff), mxy = wave_max_i(fg ? y : -1);
        const unsigned int mv = (unsigned int)wave_max_i((int)(fg ? v : 0u));          // positive floats: bits order like ints
        if (lane == 0) agg[wave] = WaveAgg{r0, cnt, mnx, mxx, mny, mxy, mv};
    } else {
        if (lane == 0) agg[wave].root = -1;
        if (fg) {
            atomicAdd(&p.area[base + r], 1);
            atomicMin(&p.minx[base + r], x); atomicMax(&p.maxx[base + r], x);
            atomicMin(&p.miny[base + r], y); atomicMax(&p.maxy[base + r], y);
            atomicMax(&p.maxv[base + r], v);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 0; w < 4; ++w) {
            if (agg[w].root < 0) continue;
            WaveAgg a = agg[w];
            for (int u = w + 1; u < 4; ++u)
                if (agg[u].root == a.root) {
                    a.cnt += agg[u].cnt; a.minx = min(a.minx, agg[u].minx); a.maxx = max(a.maxx, agg[u].maxx);
                    a.miny = min(a.miny, agg[u].miny); a.maxy = max(a.maxy, agg[u].maxy); a.maxv = max(a.maxv, agg[u].maxv);
                    agg[u].root = -1;
                }
            atomicAdd(&p.area[base + a.root], a.cnt);
            atomicMin(&p.minx[base + a.root], a.minx); atomicMax(&p.maxx[base + a.root], a.maxx);
            atomicMin(&p.miny[base + a.root], a.miny); atomicMax(&p.maxy[base + a.root], a.maxy);
            atomicMax(&p.maxv[base + a.root], a.maxv);
        }
    }
}

// --------------------------------------------------------------------------------------------------------- 4. select
__device__ __forceinline__ bool post_selected(const PostArgs& p, int b, long base, long i, int& rows) {
    if (p.label[base + i] != (int)i) return false;
    if (p.area[base + i] < 10) return false;
    if (__uint_as_float(p.maxv[base + i]) < p.st[b].text_thr) return false;
    CompStats c{p.minx[base + i], p.maxx[base + i], p.miny[base + i], p.maxy[base + i], 0, 0.f};
    const Dil d = dilation_of(c, p.H);
    rows = d.Y1 - d.Y0 + 1;
    return true;
}

__global__ __launch_bounds__(256) void post_count_kernel(PostArgs p) {
    __shared__ int s0[4], s1[4];
    const int b = blockIdx.y, N = p.H * p.W;
    const long base = (long)b * N, i0 = (long)blockIdx.x * SCAN_PIX;
    int cnt = 0, rows = 0;
    for (int j = 0; j < SCAN_PIX / 256; ++j) {
        const long i = i0 + j * 256 + threadIdx.x;
        int r = 0;
        if (i < N && post_selected(p, b, base, i, r)) { ++cnt; rows += r; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { cnt += __shfl_xor(cnt, o, 64); rows += __shfl_xor(rows, o, 64); }
    if ((threadIdx.x & 63) == 0) { s0[threadIdx.x >> 6] = cnt; s1[threadIdx.x >> 6] = rows; }
    __syncthreads();
    if (threadIdx.x == 0) p.blk[b * p.n_blk + blockIdx.x] = make_int2(s0[0] + s0[1] + s0[2] + s0[3], s1[0] + s1[1] + s1[2] + s1[3]);
}

__global__ __launch_bounds__(256) void post_scan_kernel(PostArgs p) {   // one workgroup per page: exclusive scan of the block counts
    __shared__ int2 carry;
    __shared__ int2 buf[256];
    const int b = blockIdx.x, t = threadIdx.x;
    if (t == 0) carry = make_int2(0, 0);
    __syncthreads();
    for (int base = 0; base < p.n_blk; base += 256) {
        const int i = base + t;
        int2 v = i < p.n_blk ? p.blk[b * p.n_blk + i] : make_int2(0, 0);
        buf[t] = v;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {                      // Hillis-Steele inclusive scan (256 entries, 8 rounds)
            int2 add = t >= o ? buf[t - o] : make_int2(0, 0);
            __syncthreads();
            buf[t].x += add.x; buf[t].y += add.y;
            __syncthreads();
        }
        if (i < p.n_blk) p.blk[b * p.n_blk + i] = make_int2(carry.x + buf[t].x - v.x, carry.y + buf[t].y - v.y);
        __syncthreads();
        if (t == 255) { carry.x += buf[255].x; carry.y += buf[255].y; }
        __syncthreads();
    }
    if (t == 0) {
        p.st[b].n_sel = carry.x; p.st[b].n_rows = carry.y;
        if (carry.x > p.max_boxes || carry.y > p.row_cap) p.st[b].overflow = 1;
    }
}

// scatter in raster order: thread-sequential inside a block (roots are sparse: a few per 4096 pixels)
__global__ __launch_bounds__(256) void post_scatter_kernel(PostArgs p) {
    __shared__ int2 wsum[4];
    __shared__ int2 run;
    const int b = blockIdx.y, N = p.H * p.W, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const long base = (long)b * N, i0 = (long)blockIdx.x * SCAN_PIX;
    if (p.st[b].overflow) return;
    if (t == 0) run = p.blk[b * p.n_blk + blockIdx.x];
    __syncthreads();
    for (int j = 0; j < SCAN_PIX / 256; ++j) {
        const long i = i0 + j * 256 + t;
        int rows = 0;
        const bool sel = i < N && post_selected(p, b, base, i, rows);
        // exclusive prefix of (sel, rows) over the 256 threads of this sweep, raster order = thread order
        int c = sel ? 1 : 0, r = rows;
        int ci = c, ri = r;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int cu = __shfl_up(ci, o, 64), ru = __shfl_up(ri, o, 64);
            if (lane >= o) { ci += cu; ri += ru; }
        }
        if (lane == 63) wsum[wave] = make_int2(ci, ri);
        __syncthreads();
        int2 off = run;
        for (int w = 0; w < wave; ++w) { off.x += wsum[w].x; off.y += wsum[w].y; }
        if (sel) {
            const int idx = off.x + ci - 1, roff = off.y + ri - rows;
            CompStats cs{p.minx[base + i], p.maxx[base + i], p.miny[base + i], p.maxy[base + i], p.area[base + i],
                         __uint_as_float(p.maxv[base + i])};
            p.comp[b * p.max_boxes + idx] = cs;
            p.comp_rowoff[b * p.max_boxes + idx] = roff;
            for (int k = 0; k < rows; ++k) {                     // this component's slice of the row-extreme arrays
                p.rmin[(long)b * p.row_cap + roff + k] = 0x7fffffff;
                p.rmax[(long)b * p.row_cap + roff + k] = -1;
            }
            atomicMax(reinterpret_cast<unsigned int*>(&p.st[b].max_conf), p.maxv[base + i]);     // positive floats order like their bits
        }
        __syncthreads();
        if (t == 0) { run.x += wsum[0].x + wsum[1].x + wsum[2].x + wsum[3].x; run.y += wsum[0].y + wsum[1].y + wsum[2].y + wsum[3].y; }
        __syncthreads();
        // the compact index of a root replaces its area (non-roots are reached through label -> root)
        if (i < N && p.label[base + i] == (int)i) p.area[base + i] = sel ? (off.x + ci - 1) : -1;
    }
}

// --------------------------------------------------------------------------------------------------- 5. row extremes
__global__ __launch_bounds__(256) void post_rows_kernel(PostArgs p) {
    const int b = blockIdx.y, N = p.H * p.W;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (p.st[b].overflow) return;
    const long base = (long)b * N;
    int idx = -1, x = 0, y = 0;
    if (i < N) {
        const int r = p.label[base + i];
        if (r >= 0) idx = p.area[base + r];
        x = (int)(i % p.W); y = (int)(i / p.W);
    }
    // same aggregation as post_stats_kernel: a wave inside one row of one kept component sends one min and one max
    const bool on = idx >= 0;
    const unsigned long long onm = __ballot(on);
    if (!onm) return;
    const int first = __builtin_ctzll(onm);
    const int idx0 = __shfl(idx, first, 64), y0 = __shfl(y, first, 64);
    const bool uniform = __ballot(on && (idx != idx0 || y != y0)) == 0;
    if (uniform) {
        const int mn = wave_min_i(on ? x : 0x7fffffff), mx = wave_max_i(on ? x : -1);
        if ((threadIdx.x & 63) == first) {
            const long o = (long)b * p.row_cap + p.comp_rowoff[b * p.max_boxes + idx0] + (y0 - p.comp[b * p.max_boxes + idx0].y0);
            atomicMin(&p.rmin[o], mn);
            atomicMax(&p.rmax[o], mx);
        }
    } else if (on) {
        const long o = (long)b * p.row_cap + p.comp_rowoff[b * p.max_boxes + idx] + (y - p.comp[b * p.max_boxes + idx].y0);
        atomicMin(&p.rmin[o], x);
        atomicMax(&p.rmax[o], x);
    }
}

// --------------------------------------------------------------------------------------------------------- 6. geometry
// One wave per component. Lanes share the dilated-row sweep and the per-edge rectangle areas; the sequential parts (monotone
// chain, the first-strictly-smaller tie rule of the calipers) run on lane 0. The component's points / hull stack / edge areas
// need 64 bytes per dilated row. Three launches share the components by their row count (each walks the page's component list
// and skips what is not its own; a class that cannot occur on the page is not launched):
//   class 0  rows <= POST_SMALL_ROWS   16 KB of LDS per workgroup -- every text line; many workgroups per CU
//   class 1  rows <= POST_LDS_ROWS     the whole 160 KB LDS -- page-high blobs, vertical rules (one workgroup per CU)
//   class 2  taller                    a per-workgroup slice of the caller's workspace (pages of 2560 rows and more: round 2
//                                      sized ONE launch's LDS by the page height and refused those pages; LDS is ~2x faster
//                                      than the workspace for the lane-0 passes, so it is used wherever it fits)
constexpr int POST_SMALL_ROWS = 255;
constexpr int POST_LDS_ROWS = 2555;                               // 64 * rows + 64 (+ the static word) <= 160 KB
constexpr int POST_BIG_BLOCKS = 8;                                // workgroups (scratch slices) per page of the class-2 launch
static inline size_t post_scratch_bytes(int rows) { return (size_t)64 * rows + 64; }
__device__ __forceinline__ int post_class_of(int rows) { return rows <= POST_SMALL_ROWS ? 0 : (rows <= POST_LDS_ROWS ? 1 : 2); }

__device__ __forceinline__ void post_box_of(const PostArgs& p, int b, int ci, int lane, const CompStats& c, const Dil& d, int rows,
                                            Pt* pts, Pt* stack, double* areas, int* m_sh) {
    const int* rmin = p.rmin + (long)b * p.row_cap + p.comp_rowoff[b * p.max_boxes + ci];
    const int* rmax = p.rmax + (long)b * p.row_cap + p.comp_rowoff[b * p.max_boxes + ci];
    const int n = 2 * rows;
    int l = 0x7fffffff, r = -1;
    for (int k = lane; k < rows; k += 64) {
        int L, R;
        dilated_row(c, d, rmin, rmax, d.Y0 + k, p.W, L, R);
        pts[2 * k] = Pt{L, d.Y0 + k};
        pts[2 * k + 1] = Pt{R, d.Y0 + k};
        l = min(l, L); r = max(r, R);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { l = min(l, __shfl_xor(l, o, 64)); r = max(r, __shfl_xor(r, o, 64)); }
    __syncthreads();
    if (lane == 0) *m_sh = hull_from_rows(pts, n, stack);
    __syncthreads();
    const int m = *m_sh;
    float* out = p.boxes + ((long)b * p.max_boxes + ci) * 8;
    if (m >= 3) {
        double e[6];
        for (int i = lane; i < m; i += 64) areas[i] = edge_rect(stack, m, i, e);
        __syncthreads();
        if (lane == 0) {
            double best_area = 1e300;
            int best = -1;
            for (int i = 0; i < m; ++i) {
                const double a = areas[i];
                if (a >= 0.0 && a < best_area - 1e-9) { best_area = a; best = i; }
            }
            float box[8];
            if (best >= 0) {
                edge_rect(stack, m, best, e);
                rect_corners(e, box);
            } else {
                const float u[8] = {(float)l, (float)d.Y0, (float)r, (float)d.Y0, (float)r, (float)d.Y1, (float)l, (float)d.Y1};
                for (int q = 0; q < 8; ++q) box[q] = u[q];
            }
            finish_box(box, l, r, d.Y0, d.Y1);
            for (int q = 0; q < 8; ++q) out[q] = box[q];
        }
    } else if (lane == 0) {
        float box[8] = {(float)l, (float)d.Y0, (float)r, (float)d.Y0, (float)r, (float)d.Y1, (float)l, (float)d.Y1};
        finish_box(box, l, r, d.Y0, d.Y1);
        for (int q = 0; q < 8; ++q) out[q] = box[q];
    }
    if (lane == 0) {
        const float mc = p.st[b].max_conf;
        p.conf[b * p.max_boxes + ci] = mc > 0.f ? c.maxv / mc : c.maxv;
    }
}

template <int CLASS>
__global__ __launch_bounds__(64) void post_boxes_kernel(PostArgs p, unsigned char* scratch, size_t scratch_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.y, lane = threadIdx.x;
    if (p.st[b].overflow) return;
    __shared__ int m_sh;
    for (int ci = blockIdx.x; ci < p.st[b].n_sel; ci += gridDim.x) {       // grid-stride over the page's components
        const CompStats c = p.comp[b * p.max_boxes + ci];
        const Dil d = dilation_of(c, p.H);
        const int rows = d.Y1 - d.Y0 + 1;
        if (post_class_of(rows) != CLASS) continue;                        // wave-uniform: another launch owns this component
        __syncthreads();
        if constexpr (CLASS == 2) {
            unsigned char* base = scratch + ((size_t)b * gridDim.x + blockIdx.x) * scratch_stride;
            Pt* pts = reinterpret_cast<Pt*>(base);                         // [2 rows]
            Pt* stack = pts + 2 * rows;                                    // [4 rows + 4]
            double* areas = reinterpret_cast<double*>(stack + 4 * rows + 4);   // [hull size <= 2 rows]
            post_box_of(p, b, ci, lane, c, d, rows, pts, stack, areas, &m_sh);
        } else {
            Pt* pts = reinterpret_cast<Pt*>(smem);
            Pt* stack = pts + 2 * rows;
            double* areas = reinterpret_cast<double*>(stack + 4 * rows + 4);
            post_box_of(p, b, ci, lane, c, d, rows, pts, stack, areas, &m_sh);
        }
    }
}

__global__ void post_count_out_kernel(PostArgs p) {             // pages without components / with overflow still report a count
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < p.B) p.count[b] = p.st[b].overflow ? -1 : p.st[b].n_sel;
}

// ------------------------------------------------------------------------------------------------------------ host side
static inline size_t post_align(size_t v) { return (v + 255) & ~(size_t)255; }

struct PostLayout {
    size_t st, hist, psum, pcnt, label, area, minx, maxx, miny, maxy, maxv, blk, comp, rowoff, rmin, rmax, big, big_stride, total;
    int n_blk, row_cap;
};

static inline PostLayout post_layout(int B, int H, int W, int max_boxes) {
    PostLayout L;
    const size_t N = (size_t)H * W;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = post_align(off + bytes); return o; };
    L.n_blk = (int)((N + SCAN_PIX - 1) / SCAN_PIX);
    L.row_cap = (int)std::min<size_t>(N, (size_t)1 << 30);      // dilated rows of all kept components of a page (<= ~1.1 x foreground pixels)
    L.st = take(B * sizeof(PageState));
    L.hist = take((size_t)B * 2048 * 4);
    L.psum = take((size_t)B * 1024 * 8); L.pcnt = take((size_t)B * 1024 * 4);
    L.label = take(B * N * 4); L.area = take(B * N * 4);
    L.minx = take(B * N * 4); L.maxx = take(B * N * 4); L.miny = take(B * N * 4); L.maxy = take(B * N * 4); L.maxv = take(B * N * 4);
    L.blk = take((size_t)B * L.n_blk * sizeof(int2));
    L.comp = take((size_t)B * max_boxes * sizeof(CompStats));
    L.rowoff = take((size_t)B * max_boxes * 4);
    L.rmin = take((size_t)B * L.row_cap * 4); L.rmax = take((size_t)B * L.row_cap * 4);
    L.big_stride = H > POST_LDS_ROWS ? post_align(post_scratch_bytes(H)) : 0;      // tall-component scratch: only pages that can hold one
    L.big = take((size_t)B * POST_BIG_BLOCKS * L.big_stride);
    L.total = off;
    return L;
}

static inline int post_run(const float* heat, long page_stride, int B, int H, int W, float text_threshold, float low_text,
                           int max_boxes, float* boxes, float* conf, int* count, void* workspace, size_t workspace_bytes,
                           hipStream_t s) {
    if (!heat || !boxes || !conf || !count || !workspace || B <= 0 || H <= 0 || W <= 0 || max_boxes <= 0) return SA_ERR_ARG;
    if (W % 4 || (page_stride % 4) || ((uintptr_t)heat % 16)) return SA_ERR_SHAPE;      // 16-byte row sweeps
    const PostLayout L = post_layout(B, H, W, max_boxes);
    if (workspace_bytes < L.total) return SA_ERR_NOMEM;
    char* w = reinterpret_cast<char*>(workspace);
    const int N = H * W;
    PostArgs p;
    p.heat = heat; p.page_stride = page_stride; p.B = B; p.H = H; p.W = W; p.max_boxes = max_boxes; p.row_cap = L.row_cap;
    p.text_threshold = text_threshold; p.low_text = low_text;
    p.st = (PageState*)(w + L.st); p.hist = (unsigned int*)(w + L.hist); p.psum = (double*)(w + L.psum); p.pcnt = (unsigned int*)(w + L.pcnt); p.label = (int*)(w + L.label); p.area = (int*)(w + L.area);
    p.minx = (int*)(w + L.minx); p.maxx = (int*)(w + L.maxx); p.miny = (int*)(w + L.miny); p.maxy = (int*)(w + L.maxy);
    p.maxv = (unsigned int*)(w + L.maxv); p.blk = (int2*)(w + L.blk); p.n_blk = L.n_blk; p.comp = (CompStats*)(w + L.comp);
    p.comp_rowoff = (int*)(w + L.rowoff); p.rmin = (int*)(w + L.rmin); p.rmax = (int*)(w + L.rmax);
    p.boxes = boxes; p.conf = conf; p.count = count;
    SA_HIP(hipMemsetAsync(p.hist, 0, (size_t)B * 2048 * 4, s));
    const dim3 sweep(std::min(1024, cdiv(N, 1024)), B), pix(cdiv(N, 256), B);
    p.sweep_blocks = (int)sweep.x;
    hipLaunchKernelGGL(post_hist_kernel<0>, sweep, dim3(256), 0, s, p);
    hipLaunchKernelGGL(post_pick_kernel<0>, dim3(B), dim3(256), 0, s, p);
    hipLaunchKernelGGL(post_hist_kernel<1>, sweep, dim3(256), 0, s, p);
    hipLaunchKernelGGL(post_pick_kernel<1>, dim3(B), dim3(256), 0, s, p);
    hipLaunchKernelGGL(post_hist_kernel<2>, sweep, dim3(256), 0, s, p);
    hipLaunchKernelGGL(post_pick_kernel<2>, dim3(B), dim3(256), 0, s, p);
    hipLaunchKernelGGL(post_topsum_kernel, sweep, dim3(256), 0, s, p);
    hipLaunchKernelGGL(post_thresholds_kernel, dim3(cdiv(B, 64)), dim3(64), 0, s, p);
    hipLaunchKernelGGL(post_init_kernel, pix, dim3(256), 0, s, p);
    hipLaunchKernelGGL(post_merge_kernel, pix, dim3(256), 0, s, p);
    hipLaunchKernelGGL(post_stats_kernel, pix, dim3(256), 0, s, p);
    const dim3 scan(L.n_blk, B);
    hipLaunchKernelGGL(post_count_kernel, scan, dim3(256), 0, s, p);
    hipLaunchKernelGGL(post_scan_kernel, dim3(B), dim3(256), 0, s, p);
    hipLaunchKernelGGL(post_scatter_kernel, scan, dim3(256), 0, s, p);
    hipLaunchKernelGGL(post_rows_kernel, pix, dim3(256), 0, s, p);
    hipLaunchKernelGGL(post_count_out_kernel, dim3(cdiv(B, 64)), dim3(64), 0, s, p);
    // geometry: three launches by component height (see post_boxes_kernel)
    {
        static AttrOnce a0, a1;
        const size_t lds0 = post_scratch_bytes(std::min(H, POST_SMALL_ROWS));
        a0.ensure(post_boxes_kernel<0>, lds0);
        hipLaunchKernelGGL(post_boxes_kernel<0>, dim3(std::min(max_boxes, 256), B), dim3(64), lds0, s, p, (unsigned char*)nullptr, (size_t)0);
        if (H > POST_SMALL_ROWS) {
            const size_t lds1 = post_scratch_bytes(std::min(H, POST_LDS_ROWS));
            a1.ensure(post_boxes_kernel<1>, lds1);
            hipLaunchKernelGGL(post_boxes_kernel<1>, dim3(32, B), dim3(64), lds1, s, p, (unsigned char*)nullptr, (size_t)0);
        }
        if (H > POST_LDS_ROWS)
            hipLaunchKernelGGL(post_boxes_kernel<2>, dim3(POST_BIG_BLOCKS, B), dim3(64), 0, s, p, (unsigned char*)(w + L.big), L.big_stride);
    }
    return (int)hipGetLastError();
}

}  // namespace post
}  // namespace sa
