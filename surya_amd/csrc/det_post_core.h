// Heat-map -> text-box geometry shared by the HIP kernels (det_post.h) and a plain-C++ build used by the CPU tests
// (tests/native/det_post_host.cpp): everything here is a pure per-component function with no launch or memory-model code.
//
// What it computes, for ONE connected component of `heat > low_text` (surya/detection/heatmap.py:44-98):
//   * the component is dilated by a (niter + 1)^2 rectangle anchored at its centre (cv2.dilate semantics: dst(y, x) =
//     max over src(y + dy, x + dx), dx, dy in [-a, k - 1 - a], a = k / 2), clipped to the image;
//   * minAreaRect of the dilated pixels: the convex hull only needs each dilated row's leftmost / rightmost pixel, and a
//     dilated row's extremes follow from the source rows' extremes (min over the k rows that reach it, shifted), so the
//     pixels of the dilated mask are never materialised;
//   * rotating calipers over the hull edges in float64, the same vertex order / tie rule as surya_amd/detection/heatmap.py
//     (min_area_rect_points), so the chosen rectangle is the same one;
//   * near-square -> upright box, clockwise order starting at the smallest x + y.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define SA_HD __host__ __device__ __forceinline__
#else
#define SA_HD inline
#endif

namespace sa {
namespace post {

struct Pt { int x, y; };

struct CompStats {          // of one connected component, in page pixel coordinates
    int x0, x1, y0, y1;     // inclusive bounding box
    int area;
    float maxv;
};

// dilation geometry of a component: k x k rectangle, reach `lo` towards smaller and `hi` towards larger coordinates
struct Dil { int k, lo, hi, Y0, Y1; };

SA_HD Dil dilation_of(const CompStats& c, int H) {
    const int w = c.x1 - c.x0 + 1, h = c.y1 - c.y0 + 1;
    const int niter = (int)sqrt((double)(w < h ? w : h));       // int(np.sqrt(min(w, h)))
    Dil d;
    d.k = niter + 1;
    d.hi = d.k / 2;                // a source pixel lights dst x in [xs - (k - 1 - a), xs + a]
    d.lo = d.k - 1 - d.hi;
    d.Y0 = c.y0 - d.lo < 0 ? 0 : c.y0 - d.lo;
    d.Y1 = c.y1 + d.hi > H - 1 ? H - 1 : c.y1 + d.hi;
    return d;
}

// Extremes of dilated row Y from the source-row extremes rmin / rmax (indexed by ys - c.y0).
SA_HD void dilated_row(const CompStats& c, const Dil& d, const int* rmin, const int* rmax, int Y, int W, int& L, int& R) {
    int a = Y - d.hi, b = Y + d.lo;                             // source rows that reach Y
    if (a < c.y0) a = c.y0;
    if (b > c.y1) b = c.y1;
    int l = 0x7fffffff, r = -1;
    for (int ys = a; ys <= b; ++ys) {
        const int mn = rmin[ys - c.y0], mx = rmax[ys - c.y0];
        l = mn < l ? mn : l;
        r = mx > r ? mx : r;
    }
    L = l - d.lo < 0 ? 0 : l - d.lo;
    R = r + d.hi > W - 1 ? W - 1 : r + d.hi;
}

SA_HD long cross3(const Pt& a, const Pt& b, const Pt& p) {     // (b - a) x (p - a)
    return (long)(b.x - a.x) * (p.y - a.y) - (long)(b.y - a.y) * (p.x - a.x);
}

// Convex hull of points given sorted by (y, x) (rows ascending, left before right), duplicates allowed. Andrew's monotone
// chain run with the roles of x and y swapped; the result is returned in the order heatmap.py's _convex_hull produces:
// counter-clockwise in (x, y), starting at the lexicographically smallest (x, y) vertex. `stack` needs 2 n + 2 entries, the
// hull is written to stack[0 .. m). Returns m (vertices; collinear points dropped, like `<= 0` pops do there).
SA_HD int hull_from_rows(const Pt* pts, int n, Pt* stack) {
    // de-duplicate consecutive equal points on the fly (np.unique in the host version)
    int m = 0;
    // "lower" chain in swapped coordinates: cross' = (b.y - a.y) (p.x - a.x) - (b.x - a.x) (p.y - a.y) = -cross3
    for (int i = 0; i < n; ++i) {
        const Pt p = pts[i];
        if (i > 0 && p.x == pts[i - 1].x && p.y == pts[i - 1].y) continue;
        while (m >= 2 && -cross3(stack[m - 2], stack[m - 1], p) <= 0) --m;
        stack[m++] = p;
    }
    const int lower = m;                    // stack[0 .. lower): includes both end points
    if (lower < 2) return lower;            // one distinct point
    int t = lower - 1;                      // last point of the lower chain is the start of the upper chain
    for (int i = n - 2; i >= 0; --i) {
        const Pt p = pts[i];
        if (p.x == pts[i + 1].x && p.y == pts[i + 1].y) continue;
        while (m - t >= 2 && -cross3(stack[m - 2], stack[m - 1], p) <= 0) --m;
        stack[m++] = p;
    }
    --m;                                    // the first point was appended again at the end
    // stack[0 .. m) is counter-clockwise in swapped coordinates = clockwise in (x, y): reverse, then rotate so the
    // lexicographically smallest (x, y) vertex comes first
    for (int i = 0, j = m - 1; i < j; ++i, --j) { const Pt tp = stack[i]; stack[i] = stack[j]; stack[j] = tp; }
    int s = 0;
    for (int i = 1; i < m; ++i)
        if (stack[i].x < stack[s].x || (stack[i].x == stack[s].x && stack[i].y < stack[s].y)) s = i;
    if (s) {                                // rotate left by s using the free upper half of the stack as scratch
        for (int i = 0; i < m; ++i) stack[m + i] = stack[(i + s) % m];
        for (int i = 0; i < m; ++i) stack[i] = stack[m + i];
    }
    return m;
}

// Area of the rectangle aligned with hull edge i (float64, the arithmetic of min_area_rect_points); < 0 for a zero edge.
SA_HD double edge_rect(const Pt* hull, int m, int i, double* uvext /* u0,u1 (dir), umin, umax, vmin, vmax */) {
#if defined(__clang__)
#pragma clang fp contract(off)              // same products and sums on the CPU test build and on the device
#endif
    const Pt a = hull[i], b = hull[(i + 1) % m];
    const double ex = (double)(b.x - a.x), ey = (double)(b.y - a.y);
    const double nrm = hypot(ex, ey);
    if (nrm == 0.0) return -1.0;
    const double ux = ex / nrm, uy = ey / nrm, vx = -uy, vy = ux;
    double umin = 1e300, umax = -1e300, vmin = 1e300, vmax = -1e300;
    for (int j = 0; j < m; ++j) {
        const double pu = (double)hull[j].x * ux + (double)hull[j].y * uy;
        const double pv = (double)hull[j].x * vx + (double)hull[j].y * vy;
        umin = pu < umin ? pu : umin; umax = pu > umax ? pu : umax;
        vmin = pv < vmin ? pv : vmin; vmax = pv > vmax ? pv : vmax;
    }
    uvext[0] = ux; uvext[1] = uy; uvext[2] = umin; uvext[3] = umax; uvext[4] = vmin; uvext[5] = vmax;
    return (umax - umin) * (vmax - vmin);
}

SA_HD void rect_corners(const double* e, float* box /* [4][2] */) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const double ux = e[0], uy = e[1], vx = -e[1], vy = e[0];
    const double cu[4] = {e[2], e[3], e[3], e[2]}, cv[4] = {e[4], e[4], e[5], e[5]};
    for (int i = 0; i < 4; ++i) {
        box[2 * i] = (float)(ux * cu[i] + vx * cv[i]);
        box[2 * i + 1] = (float)(uy * cu[i] + vy * cv[i]);
    }
}

// detect_boxes' tail (heatmap.py: near-square test, clockwise order, roll to the smallest x + y). l, r, t, b = extent of
// the dilated pixels. box is rewritten in place.
SA_HD void finish_box(float* box, int l, int r, int t, int b) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const float d01x = box[0] - box[2], d01y = box[1] - box[3], d12x = box[2] - box[4], d12y = box[3] - box[5];
    const float bw = sqrtf(d01x * d01x + d01y * d01y), bh = sqrtf(d12x * d12x + d12y * d12y);
    const float mx = bw > bh ? bw : bh, mn = bw > bh ? bh : bw;
    if (fabsf(1.0f - mx / (mn + 1e-5f)) <= 0.1f) {
        const float u[8] = {(float)l, (float)t, (float)r, (float)t, (float)r, (float)b, (float)l, (float)b};
        for (int i = 0; i < 8; ++i) box[i] = u[i];
    }
    const float cx = (((box[0] + box[2]) + box[4]) + box[6]) / 4.0f, cy = (((box[1] + box[3]) + box[5]) + box[7]) / 4.0f;
    float ang[4];
    int ord[4] = {0, 1, 2, 3};
    for (int i = 0; i < 4; ++i) ang[i] = atan2f(box[2 * i + 1] - cy, box[2 * i] - cx);
    for (int i = 1; i < 4; ++i)             // insertion sort (stable) by angle
        for (int j = i; j > 0 && ang[ord[j]] < ang[ord[j - 1]]; --j) { const int tp = ord[j]; ord[j] = ord[j - 1]; ord[j - 1] = tp; }
    float s[8];
    for (int i = 0; i < 4; ++i) { s[2 * i] = box[2 * ord[i]]; s[2 * i + 1] = box[2 * ord[i] + 1]; }
    int st = 0;
    float best = s[0] + s[1];
    for (int i = 1; i < 4; ++i) {
        const float v = s[2 * i] + s[2 * i + 1];
        if (v < best) { best = v; st = i; }
    }
    for (int i = 0; i < 4; ++i) { box[2 * i] = s[2 * ((i + st) & 3)]; box[2 * i + 1] = s[2 * ((i + st) & 3) + 1]; }
}

// Whole per-component geometry, sequential form (one thread; the HIP kernel spreads the row and edge loops over a wave and
// calls the same pieces). pts: 2 * rows entries, stack: 4 * rows + 4 entries. Returns false for a degenerate component.
SA_HD bool component_box(const CompStats& c, const int* rmin, const int* rmax, int H, int W, Pt* pts, Pt* stack, float* box) {
    const Dil d = dilation_of(c, H);
    int n = 0, l = 0x7fffffff, r = -1;
    for (int Y = d.Y0; Y <= d.Y1; ++Y) {
        int L, R;
        dilated_row(c, d, rmin, rmax, Y, W, L, R);
        pts[n].x = L; pts[n].y = Y; ++n;
        pts[n].x = R; pts[n].y = Y; ++n;
        l = L < l ? L : l; r = R > r ? R : r;
    }
    const int m = hull_from_rows(pts, n, stack);
    if (m < 3) {                            // heatmap.py min_area_rect_points: len(hull) < 3 -> the points' upright box
        const float u[8] = {(float)l, (float)d.Y0, (float)r, (float)d.Y0, (float)r, (float)d.Y1, (float)l, (float)d.Y1};
        for (int i = 0; i < 8; ++i) box[i] = u[i];
    } else {
        double best[6], cur[6], best_area = 1e300;
        bool have = false;
        for (int i = 0; i < m; ++i) {
            const double area = edge_rect(stack, m, i, cur);
            if (area < 0.0) continue;
            if (area < best_area - 1e-9) {
                best_area = area; have = true;
                for (int q = 0; q < 6; ++q) best[q] = cur[q];
            }
        }
        if (!have) return false;
        rect_corners(best, box);
    }
    finish_box(box, l, r, d.Y0, d.Y1);
    return true;
}

}  // namespace post
}  // namespace sa
