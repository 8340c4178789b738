// CPU build of the heat-map -> boxes pipeline used by tests/test_det_post_cpu.py (g++, no GPU): the per-component geometry is
// surya_amd/csrc/det_post_core.h VERBATIM (the same header the HIP kernels include); thresholds, labelling, statistics and
// compaction are restated sequentially here with the same definitions the kernels use (rank-select of the top-10 % mean,
// 4-connected union-find with the smallest raster index as root, components ordered by root). Test infrastructure only.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../surya_amd/csrc/det_post_core.h"

using namespace sa::post;

static int find_root(std::vector<int>& lab, int x) {
    while (lab[x] != x) { lab[x] = lab[lab[x]]; x = lab[x]; }
    return x;
}

extern "C" int det_post_host(const float* heat, int H, int W, float text_threshold, float low_text, int max_boxes, float* boxes,
                             float* conf, int* count, float* thr_out) {
    const int N = H * W;
    // get_dynamic_thresholds (heatmap.py:14-24): mean of the N - k largest values, k = int(N * 0.9)
    const int k = (int)((double)N * 0.9);
    std::vector<float> v(heat, heat + N);
    std::nth_element(v.begin(), v.begin() + k, v.end());
    const float vk = v[k];
    double sum_gt = 0.0; long cnt_gt = 0;
    for (int i = 0; i < N; ++i) if (heat[i] > vk) { sum_gt += heat[i]; ++cnt_gt; }
    const float avg = (float)((sum_gt + (double)(N - k - cnt_gt) * (double)vk) / (double)(N - k));
    float sc = avg / 0.7f;
    sc = sc < 0.f ? 0.f : (sc > 1.f ? 1.f : sc);
    sc = sqrtf(sc);
    float tt = text_threshold * sc, lt = low_text * sc;
    tt = tt < 0.15f ? 0.15f : (tt > 0.8f ? 0.8f : tt);
    lt = lt < 0.1f ? 0.1f : (lt > 0.6f ? 0.6f : lt);
    if (thr_out) { thr_out[0] = tt; thr_out[1] = lt; thr_out[2] = avg; }
    // labels: union-find, root = smallest raster index of the component
    std::vector<int> lab(N);
    for (int i = 0; i < N; ++i) lab[i] = heat[i] > lt ? i : -1;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const int p = y * W + x;
            if (lab[p] < 0) continue;
            const int nb[2] = {x > 0 ? p - 1 : -1, y > 0 ? p - W : -1};
            for (int q : nb) {
                if (q < 0 || lab[q] < 0) continue;
                int a = find_root(lab, p), b = find_root(lab, q);
                if (a == b) continue;
                if (a < b) lab[b] = a; else lab[a] = b;
            }
        }
    std::vector<CompStats> st(N);
    std::vector<int> roots;
    for (int p = 0; p < N; ++p) {
        if (lab[p] < 0) continue;
        const int r = find_root(lab, p);
        lab[p] = r;
        const int x = p % W, y = p / W;
        if (r == p) { st[r] = CompStats{x, x, y, y, 0, heat[p]}; roots.push_back(r); }   // root is the first pixel in raster order
        CompStats& c = st[r];
        c.x0 = std::min(c.x0, x); c.x1 = std::max(c.x1, x); c.y0 = std::min(c.y0, y); c.y1 = std::max(c.y1, y);
        c.area += 1; c.maxv = std::max(c.maxv, heat[p]);
    }
    int n = 0;
    float max_conf = 0.f;
    std::vector<int> rmin, rmax;
    std::vector<Pt> pts, stack;
    for (int r : roots) {
        const CompStats& c = st[r];
        if (c.area < 10 || c.maxv < tt) continue;
        if (n >= max_boxes) return -1;
        const int h = c.y1 - c.y0 + 1;
        rmin.assign(h, 0x7fffffff); rmax.assign(h, -1);
        for (int y = c.y0; y <= c.y1; ++y)
            for (int x = c.x0; x <= c.x1; ++x)
                if (lab[y * W + x] == r) { rmin[y - c.y0] = std::min(rmin[y - c.y0], x); rmax[y - c.y0] = std::max(rmax[y - c.y0], x); }
        const Dil d = dilation_of(c, H);
        const int rows = d.Y1 - d.Y0 + 1;
        pts.resize(2 * rows); stack.resize(4 * rows + 4);
        if (!component_box(c, rmin.data(), rmax.data(), H, W, pts.data(), stack.data(), boxes + 8 * n)) continue;
        conf[n] = c.maxv;
        max_conf = std::max(max_conf, c.maxv);
        ++n;
    }
    if (max_conf > 0.f)
        for (int i = 0; i < n; ++i) conf[i] = conf[i] / max_conf;
    *count = n;
    return 0;
}
