// Where does a chunk iteration of mbconv_kernel (csrc/det_mbconv.h) go? Instrumented build (-DSA_MBC_TIMING=1): wave 0 (an X wave) and wave 4 (a D
// wave) of every workgroup stamp s_memtime four times per iteration; this program launches the two shipped shapes on random operands (16 pages) and
// prints the launch time (hipEvents, plain build semantics otherwise) and the mean segment lengths.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSA_MBC_TIMING=1 -Iinclude tools/microbench/mbconv_timing.hip -o /tmp/mbconv_timing && /tmp/mbconv_timing
// X segments: [0,1] W1 requests + MFMA loop   [1,2] bias, Hardswish, E stores   [2,3] vmcnt(0)      (then the barrier)
// D segments: [0,1] taps, W2 requests, depthwise, D store   [1,2] projection   [2,3] barrier wait
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../surya_amd/csrc/det_mbconv.h"
using namespace sa;

__global__ void fill_kernel(bf16_t* p, long n, unsigned seed, float scale) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned x = (unsigned)i * 2654435761u + seed;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[i] = f2bf(((float)(x & 0xffff) / 32768.f - 1.f) * scale);
}
static bf16_t* rnd(long n, unsigned seed, float scale) {
    bf16_t* p; hipMalloc(&p, n * 2);
    fill_kernel<<<(int)((n + 255) / 256), 256>>>(p, n, seed, scale);
    return p;
}

static void run(int B, int H, int Cin, int Cm, int Cout) {
    const int Ho = H / 2;
    bf16_t *in = rnd((long)B * H * H * Cin, 1, 1.f), *w1 = rnd((long)Cm * Cin, 2, 0.08f), *b1 = rnd(Cm, 3, 0.5f), *wd = rnd(9L * Cm, 4, 0.3f),
           *bd = rnd(Cm, 5, 0.5f), *w2 = rnd((long)Cout * Cm, 6, 0.03f), *b2 = rnd(Cout, 7, 0.5f), *out = rnd((long)B * Ho * Ho * Cout, 8, 0.f);
#if SA_MBC_TIMING
    const int th = Cin == 128 ? 8 : 4, grid = B * (Ho / th) * (Ho / 8);
    const size_t dbg_n = (size_t)grid * 2 * 40 * 8;
    hipMalloc(&g_mbc_dbg, dbg_n * 8); hipMemset(g_mbc_dbg, 0, dbg_n * 8);
#endif
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0, 0);
        int rc = launch_mbconv(in, w1, b1, wd, bd, w2, b2, nullptr, out, B, H, H, Cin, Cm, Ho, Ho, Cout, 2, 0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        if (rc) { printf("launch rc %d\n", rc); return; }
        float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
    }
    const double gf = 2.0 * B * ((double)H * H * Cin * Cm + (double)Ho * Ho * Cm * (9 + Cout)) * 1e-9;
    printf("mbconv %d -> %d -> %d, %d pages %dx%d -> %dx%d: %.1f us (%.0f TF/s)\n", Cin, Cm, Cout, B, H, H, Ho, Ho, best * 1e3, gf / best);
#if SA_MBC_TIMING
    std::vector<long long> h(dbg_n);
    hipMemcpy(h.data(), g_mbc_dbg, dbg_n * 8, hipMemcpyDeviceToHost);
    const int nch = Cm / 64;
    for (int role = 0; role < 2; ++role) {
        double seg[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long cnt = 0;
        for (int g = 0; g < grid; ++g)
            for (int it = 4; it < (nch + 1 < 39 ? nch + 1 : 39) - 2; ++it) {
                const long long* a = &h[(((size_t)g * 2 + role) * 40 + it) * 8];
                const long long* nx = &h[(((size_t)g * 2 + role) * 40 + it + 1) * 8];
                seg[0] += a[1] - a[0]; seg[1] += a[2] - a[1]; seg[2] += a[3] - a[2]; seg[3] += nx[0] - a[0]; seg[4] += a[4] - a[0]; seg[5] += a[5] - a[4]; seg[6] += a[6] - a[5]; seg[7] += a[1] - a[6]; ++cnt;
            }
        if (role) printf("  D detail: requests %.0f  first reads + bias %.0f  nine taps %.0f  Hardswish + D store %.0f\n", seg[4] / cnt, seg[5] / cnt, seg[6] / cnt, seg[7] / cnt);
        printf("  %s wave: [0,1] %.0f  [1,2] %.0f  [2,3] %.0f  iteration %.0f  (s_memtime ticks, 100 MHz: x10 ns)\n", role ? "D" : "X", seg[0] / cnt, seg[1] / cnt, seg[2] / cnt, seg[3] / cnt);
    }
    hipFree(g_mbc_dbg);
#endif
    hipFree(in); hipFree(w1); hipFree(b1); hipFree(wd); hipFree(bd); hipFree(w2); hipFree(b2); hipFree(out);
}

int main() {
    run(16, 128, 128, 2048, 256);
    run(16, 64, 256, 6144, 512);
    return 0;
}
