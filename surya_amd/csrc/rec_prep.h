// Line-crop pre-processing ON THE DEVICE: page pixels -> image_tiles, replacing the host chain
//   slice_bboxes_from_image / slice_and_pad_poly   (surya/input/processing.py:35-101: crop, outside-polygon = pad value)
//   SuryaOCRProcessor.scale_to_fit                  (processor/__init__.py:141-178: area clamp, cv2.resize LANCZOS4)
//   SuryaOCRProcessor._process_and_tile             (:185-230: round up to x28 with cv2.resize CUBIC, x/255 in fp64,
//                                                    (x - mean) / std in fp32, patchify in merge-block-major order)
// for all lines of a call at once. Pages travel to the device once as uint8 HWC (a quarter of the fp32 bytes; uint8 -> float
// is exact), every line is a descriptor (page, crop rectangle, optional 4-point polygon, sizes, destination rows).
//
// Byte / gather work, HBM-bound: one thread per output pixel (3 channels), 16 (cubic) or 64 (Lanczos) source taps served from
// L1 / L2 -- neighbouring output pixels share taps. Resampling follows surya_amd/common/imageops.py exactly (the restatement
// of cv2's INTER_CUBIC a = -0.75 / INTER_LANCZOS4 geometry used by the host path): half-pixel centres, replicate border, taps
// and sums in float64 with the same association (no FMA contraction), horizontal pass before vertical, result rounded to
// float32 once per stage. Where no resize is needed the tiles are bit-identical to the reference processor's
// (tests/golden/processor_tiles.pt).
#pragma once
#include "common.h"

namespace sa {
namespace prep {

struct LineDesc {
    long page_off;               // byte offset of the page's first pixel in `pages` (uint8 RGB, HWC)
    int page_w, page_h;
    int x0, y0, cw, ch;          // crop rectangle inside the page (already clipped; cw, ch >= 1)
    int has_poly;                // 1: pixels outside the polygon read as `pad`
    float poly[8];               // 4 (x, y) vertices relative to the crop origin (integers stored as float)
    int mid_w, mid_h;            // size after scale_to_fit (== cw, ch when no Lanczos stage)
    int out_w, out_h;            // size after rounding up to multiples of patch * merge (== mid when no cubic stage)
    long mask_off;               // byte offset into the mask arena (has_poly)
    long mid_off;                // float offset into the mid arena (mid != crop)
    long tile_row;               // first row of this line in `tiles`
};

static_assert(sizeof(LineDesc) == 112, "LineDesc layout is part of the C ABI (include/surya_amd.h, preprocess_gpu.py)");

struct PrepArgs {
    const unsigned char* pages;
    const LineDesc* lines;
    int n_lines;
    unsigned char* mask;         // arena: per polygon line [ch][cw] bytes, 1 = inside / on the boundary
    float* mid;                  // arena: per Lanczos line [mid_h][mid_w][3] floats
    float* tiles;                // [sum patches][3 * ps * ps]
    int ps, merge, max_mid_w;    // max_mid_w: widest stage-1 output of the call (grid sizing)
    int pix = 3;                 // bytes per page pixel: 3 (RGB) or 4 (RGBX, the layout PIL keeps in memory: pages then need no repacking on the host)
    float pad;                   // RECOGNITION_PAD_VALUE
    float mean[3], std[3];
};

// ----------------------------------------------------------------------------------------------------------- polygon mask
// fill_poly_mask of surya_amd/common/imageops.py (= cv2.fillPoly's boundary-inclusive convention as restated there):
// interior by the even-odd rule at pixel centres per scanline (ceil / floor of the edge intersections), then the boundary by a
// dense rasterisation of every edge (rint of max(|dx|, |dy|) + 1 equal steps).
__global__ __launch_bounds__(256) void prep_mask_rows_kernel(PrepArgs p) {
#pragma clang fp contract(off)
    const LineDesc& L = p.lines[blockIdx.y];
    if (!L.has_poly) return;
    unsigned char* m = p.mask + L.mask_off;
    for (int y = blockIdx.x * blockDim.x + threadIdx.x; y < L.ch; y += gridDim.x * blockDim.x) {
        double xs[4];
        int n = 0;
        for (int i = 0; i < 4; ++i) {
            const double x0 = L.poly[2 * i], y0 = L.poly[2 * i + 1], x1 = L.poly[2 * ((i + 1) & 3)], y1 = L.poly[2 * ((i + 1) & 3) + 1];
            if (y0 == y1) continue;
            const double lo = y0 < y1 ? y0 : y1, hi = y0 < y1 ? y1 : y0;
            if (lo <= (double)y && (double)y < hi) xs[n++] = x0 + ((double)y - y0) * (x1 - x0) / (y1 - y0);
        }
        for (int i = 1; i < n; ++i)                      // insertion sort of <= 4 intersections
            for (int j = i; j > 0 && xs[j] < xs[j - 1]; --j) { const double t = xs[j]; xs[j] = xs[j - 1]; xs[j - 1] = t; }
        unsigned char* row = m + (long)y * L.cw;
        for (int x = 0; x < L.cw; ++x) row[x] = 0;
        for (int i = 0; i + 1 < n; i += 2) {
            int xa = (int)ceil(xs[i]), xb = (int)floor(xs[i + 1]);
            if (xb < xa) continue;
            xa = xa < 0 ? 0 : xa;
            xb = xb > L.cw - 1 ? L.cw - 1 : xb;
            for (int x = xa; x <= xb; ++x) row[x] = 1;
        }
    }
}

__global__ __launch_bounds__(256) void prep_mask_edges_kernel(PrepArgs p) {
#pragma clang fp contract(off)
    const LineDesc& L = p.lines[blockIdx.y];
    if (!L.has_poly) return;
    unsigned char* m = p.mask + L.mask_off;
    for (int e = 0; e < 4; ++e) {
        const double x0 = L.poly[2 * e], y0 = L.poly[2 * e + 1], x1 = L.poly[2 * ((e + 1) & 3)], y1 = L.poly[2 * ((e + 1) & 3) + 1];
        const double ax = fabs(x1 - x0), ay = fabs(y1 - y0);
        const int steps = (int)(ax > ay ? ax : ay) + 1;                   // np.linspace(a, b, steps + 1)
        const double sx = (x1 - x0) / (double)steps, sy = (y1 - y0) / (double)steps;
        for (int k = blockIdx.x * blockDim.x + threadIdx.x; k <= steps; k += gridDim.x * blockDim.x) {
            const double fx = k == steps ? x1 : x0 + (double)k * sx, fy = k == steps ? y1 : y0 + (double)k * sy;
            const long xi = (long)rint(fx), yi = (long)rint(fy);
            if (xi >= 0 && xi < L.cw && yi >= 0 && yi < L.ch) m[yi * L.cw + xi] = 1;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------ resampling
__device__ __forceinline__ void cubic_w(double t, double* w) {            // imageops._cubic_weights, a = -0.75
#pragma clang fp contract(off)
    const double a = -0.75;
    w[0] = ((a * (t + 1) - 5 * a) * (t + 1) + 8 * a) * (t + 1) - 4 * a;
    w[1] = ((a + 2) * t - (a + 3)) * t * t + 1;
    w[2] = ((a + 2) * (1 - t) - (a + 3)) * (1 - t) * (1 - t) + 1;
    w[3] = 1.0 - w[0] - w[1] - w[2];
}
__device__ __forceinline__ double sinc_pi(double x) {                     // np.sinc
    if (x == 0.0) return 1.0;
    const double y = 3.141592653589793238462643383279502884 * x;
    return sin(y) / y;
}
__device__ __forceinline__ void lanczos4_w(double t, double* w) {         // imageops._lanczos4_weights
#pragma clang fp contract(off)
    double s = 0.0;
    for (int i = 0; i < 8; ++i) {
        const double x = t - (double)(i - 3);
        double v = fabs(x) < 1e-12 ? 1.0 : sinc_pi(x) * sinc_pi(x / 4.0);
        if (fabs(x) >= 4.0) v = 0.0;
        w[i] = v;
    }
    // np.sum over 8 contiguous doubles: pairwise is not used below 8 elements' unrolled block -> numpy adds them with its
    // 8-accumulator loop only for n >= 8: r[0..7] each one element, combined as ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7))
    s = ((w[0] + w[1]) + (w[2] + w[3])) + ((w[4] + w[5]) + (w[6] + w[7]));
    for (int i = 0; i < 8; ++i) w[i] = w[i] / s;
}

template <int TAPS>
__device__ __forceinline__ void axis_setup(int out_i, int in_len, int out_len, int* idx, double* w) {
#pragma clang fp contract(off)
    const double scale = (double)in_len / (double)out_len;
    const double src = ((double)out_i + 0.5) * scale - 0.5;
    const double base = floor(src), t = src - base;
    if (TAPS == 4) cubic_w(t, w); else lanczos4_w(t, w);
    const int first = TAPS == 4 ? -1 : -3;
    for (int k = 0; k < TAPS; ++k) {
        long j = (long)base + first + k;
        idx[k] = (int)(j < 0 ? 0 : (j > in_len - 1 ? in_len - 1 : j));   // replicate border
    }
}

// source pixel of a line's (masked) crop, channel c
__device__ __forceinline__ float crop_px(const PrepArgs& p, const LineDesc& L, int y, int x, int c) {
    if (L.has_poly && !p.mask[L.mask_off + (long)y * L.cw + x]) return p.pad;
    return (float)p.pages[L.page_off + ((long)(L.y0 + y) * L.page_w + (L.x0 + x)) * p.pix + c];
}

// out(y, x, :) of resampling a [in_h][in_w][3] source to [out_h][out_w]: horizontal pass first (per tap row), then vertical,
// float64 products and sequential sums like (gathered * w).sum(1) in imageops._resample_axis; an axis that keeps its length
// is passed through untouched (no taps), as there.
template <int TAPS, typename SRC>
__device__ __forceinline__ void resample_px(SRC src, int in_h, int in_w, int out_h, int out_w, int oy, int ox, float* out3) {
#pragma clang fp contract(off)
    int xi[TAPS], yi[TAPS];
    double wx[TAPS], wy[TAPS];
    const bool rx = in_w != out_w, ry = in_h != out_h;
    if (rx) axis_setup<TAPS>(ox, in_w, out_w, xi, wx);
    if (ry) axis_setup<TAPS>(oy, in_h, out_h, yi, wy);
    for (int c = 0; c < 3; ++c) {
        double acc = 0.0;
        const int ny = ry ? TAPS : 1;
        for (int j = 0; j < ny; ++j) {
            const int sy = ry ? yi[j] : oy;
            double h;
            if (rx) {
                h = (double)src(sy, xi[0], c) * wx[0];
                for (int k = 1; k < TAPS; ++k) h = h + (double)src(sy, xi[k], c) * wx[k];
            } else {
                h = (double)src(sy, ox, c);
            }
            if (ry) acc = j == 0 ? h * wy[0] : acc + h * wy[j];
            else acc = h;
        }
        out3[c] = (float)acc;
    }
}

// the three channels of a source pixel of a line's (masked) crop: one 32-bit load for RGBX pages
__device__ __forceinline__ void crop_px3(const PrepArgs& p, const LineDesc& L, int y, int x, double (&v)[3]) {
    if (L.has_poly && !p.mask[L.mask_off + (long)y * L.cw + x]) { v[0] = v[1] = v[2] = (double)p.pad; return; }
    const unsigned char* px = p.pages + L.page_off + ((long)(L.y0 + y) * L.page_w + (L.x0 + x)) * p.pix;
    if (p.pix == 4) {                                       // page_off and the pixel offset are multiples of 4
        const uint32_t w = *reinterpret_cast<const uint32_t*>(px);
        v[0] = (double)(w & 0xff); v[1] = (double)((w >> 8) & 0xff); v[2] = (double)((w >> 16) & 0xff);
    } else {
        v[0] = (double)px[0]; v[1] = (double)px[1]; v[2] = (double)px[2];
    }
}

// stage 1 (scale_to_fit): Lanczos-4 of the masked crop into the line's slice of the mid arena. A thread owns an output column
// and walks down the rows. Values are those of resample_px (horizontal pass per source row, then the vertical taps, float64
// products summed in tap order), computed without its redundancy:
//   * the 8 horizontal taps (16 double-precision sin() per axis position) once per column, the vertical ones once per output
//     row by the first lanes of the block, all rows up front (the first version re-derived them per pixel: 19.7 ms per call);
//   * the horizontal pass h(sy, ox, c) of a SOURCE row once, kept in an 8-slot LDS ring indexed by sy & 7: consecutive output
//     rows share 7 of their 8 source rows when a line is scaled up, and every thread of the block needs the same rows (the second
//     version recomputed it for each of the 8 vertical taps: 192 byte loads per output pixel, 25 ms per 2842-line call).
constexpr int S1_ROWS = 128;      // output rows whose vertical taps are resident at a time (61 KB of LDS per block with the ring)
__global__ __launch_bounds__(256) void prep_stage1_kernel(PrepArgs p) {
#pragma clang fp contract(off)
    __shared__ double wy_s[S1_ROWS][8];
    __shared__ int yi_s[S1_ROWS][8];
    __shared__ double ring[8][3][256];
    __shared__ int tag[8];
    const LineDesc& L = p.lines[blockIdx.y];
    if (L.mid_w == L.cw && L.mid_h == L.ch) return;
    const int tid = threadIdx.x, ox = blockIdx.x * blockDim.x + tid;
    if (blockIdx.x * blockDim.x >= L.mid_w) return;                          // whole block beyond the line's width
    const bool rx = L.cw != L.mid_w, ry = L.ch != L.mid_h, live = ox < L.mid_w;
    int xi[8];
    double wx[8];
    if (rx && live) axis_setup<8>(ox, L.cw, L.mid_w, xi, wx);
    if (tid < 8) tag[tid] = -1;
    // horizontal pass of source row sy for this thread's column
    auto hrow = [&](int sy, double (&h)[3]) {
        if (rx) {
            double v[3];
            crop_px3(p, L, sy, xi[0], v);
            h[0] = v[0] * wx[0]; h[1] = v[1] * wx[0]; h[2] = v[2] * wx[0];
            for (int k = 1; k < 8; ++k) {
                crop_px3(p, L, sy, xi[k], v);
                h[0] = h[0] + v[0] * wx[k]; h[1] = h[1] + v[1] * wx[k]; h[2] = h[2] + v[2] * wx[k];
            }
        } else {
            crop_px3(p, L, sy, ox, h);
        }
    };
    for (int r0 = 0; r0 < L.mid_h; r0 += S1_ROWS) {
        const int nr = min(S1_ROWS, L.mid_h - r0);
        __syncthreads();
        if (ry && tid < nr) axis_setup<8>(r0 + tid, L.ch, L.mid_h, yi_s[tid], wy_s[tid]);
        __syncthreads();
        for (int r = 0; r < nr; ++r) {
            const int oy = r0 + r;
            float* dst = p.mid + L.mid_off + ((long)oy * L.mid_w + ox) * 3;
            if (!ry) {                                       // height unchanged: the horizontal pass is the result
                if (live) {
                    double h[3];
                    hrow(oy, h);
                    dst[0] = (float)h[0]; dst[1] = (float)h[1]; dst[2] = (float)h[2];
                }
                continue;
            }
            // make the 8 source rows resident (block-uniform decisions: every thread sees the same tags)
            for (int j = 0; j < 8; ++j) {
                const int sy = yi_s[r][j], slot = sy & 7;
                if (tag[slot] != sy) {
                    __syncthreads();                         // everyone has finished reading the row this slot held
                    if (live) {
                        double h[3];
                        hrow(sy, h);
                        ring[slot][0][tid] = h[0]; ring[slot][1][tid] = h[1]; ring[slot][2][tid] = h[2];
                    }
                    if (tid == 0) tag[slot] = sy;
                    __syncthreads();
                }
            }
            if (live) {
                for (int c = 0; c < 3; ++c) {
                    double acc = ring[yi_s[r][0] & 7][c][tid] * wy_s[r][0];
                    for (int j = 1; j < 8; ++j) acc = acc + ring[yi_s[r][j] & 7][c][tid] * wy_s[r][j];
                    dst[c] = (float)acc;
                }
            }
        }
    }
}

// stage 2 (_process_and_tile): cubic to the x28 size, normalise, scatter into merge-block-major patch rows
__global__ __launch_bounds__(256) void prep_tiles_kernel(PrepArgs p) {
#pragma clang fp contract(off)
    const LineDesc& L = p.lines[blockIdx.y];
    const bool staged = !(L.mid_w == L.cw && L.mid_h == L.ch);
    const int gw = L.out_w / p.ps, m = p.merge, ps = p.ps, pd = 3 * ps * ps;
    const long n = (long)L.out_w * L.out_h;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int oy = (int)(i / L.out_w), ox = (int)(i % L.out_w);
        float v[3];
        if (staged) {
            const float* mid = p.mid + L.mid_off;
            const int mw = L.mid_w;
            resample_px<4>([&](int y, int x, int c) { return mid[((long)y * mw + x) * 3 + c]; }, L.mid_h, L.mid_w, L.out_h, L.out_w, oy, ox, v);
        } else {
            resample_px<4>([&](int y, int x, int c) { return crop_px(p, L, y, x, c); }, L.ch, L.cw, L.out_h, L.out_w, oy, ox, v);
        }
        // patch (gy, gx) -> row ((gy / m) * (gw / m) + gx / m) * m * m + (gy % m) * m + gx % m; element [c][py][px]
        const int gy = oy / ps, gx = ox / ps, py = oy % ps, px = ox % ps;
        const long row = L.tile_row + ((long)(gy / m) * (gw / m) + gx / m) * (m * m) + (gy % m) * m + (gx % m);
        float* dst = p.tiles + row * pd + py * ps + px;
        for (int c = 0; c < 3; ++c) {
            const float r = (float)((double)v[c] * (1.0 / 255.0));         // x * rescale_factor in fp64, cast (:181-182)
            dst[c * ps * ps] = (r - p.mean[c]) / p.std[c];
        }
    }
}

static inline int prep_run(const PrepArgs& p, int any_poly, int any_stage1, hipStream_t s) {
    if (p.n_lines <= 0) return SA_OK;
    if (!p.pages || !p.lines || !p.tiles) return SA_ERR_ARG;
    if (any_poly) {
        if (!p.mask) return SA_ERR_ARG;
        hipLaunchKernelGGL(prep_mask_rows_kernel, dim3(2, p.n_lines), dim3(256), 0, s, p);
        hipLaunchKernelGGL(prep_mask_edges_kernel, dim3(2, p.n_lines), dim3(256), 0, s, p);
    }
    if (any_stage1) {
        if (!p.mid) return SA_ERR_ARG;
        hipLaunchKernelGGL(prep_stage1_kernel, dim3(cdiv(p.max_mid_w, 256), p.n_lines), dim3(256), 0, s, p);
    }
    hipLaunchKernelGGL(prep_tiles_kernel, dim3(32, p.n_lines), dim3(256), 0, s, p);
    return (int)hipGetLastError();
}

}  // namespace prep
}  // namespace sa
