// Where does a tile of the PERSISTENT 8-phase GEMM loop spend its time? Instrumented build (-DSA_P8_TIMING=1): waves 0 and 4 of every
// workgroup stamp s_memtime at the segment boundaries of their first 24 tiles; this program launches the kernel directly on random
// operands and prints the mean segment lengths in cycles and microseconds.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSA_P8_TIMING=1 -Iinclude tools/microbench/p8_timing.hip -o /tmp/p8_timing && /tmp/p8_timing
// Segments (stamps 0..8): [0,1] phases 0-3 of the tile (no counted wait)  [1,2] phase 4 (first counted wait behind the previous tile's stores)
//   [2,3] phases 5-7   [3,4] middle K-tiles   [4,5] last two K-tiles   [5,6] epilogue pass 1 (+ residual requests)   [6,7] vmcnt(0)
//   [7,8] staging + stores
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "../../surya_amd/csrc/gemm.h"
using namespace sa;

__global__ void fill_kernel(bf16_t* p, long n, unsigned seed, float scale) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned x = (unsigned)i * 2654435761u + seed;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[i] = f2bf(((float)(x & 0xffff) / 32768.f - 1.f) * scale);
}

template <int EPI>
static void run(const char* name, int M, int N, int K) {
    bf16_t *X, *W, *C, *B, *R = nullptr;
    const int No = (EPI == EPI_SWIGLU) ? N / 2 : N;
    hipMalloc(&X, (size_t)M * K * 2); hipMalloc(&W, (size_t)N * K * 2); hipMalloc(&C, (size_t)M * No * 2); hipMalloc(&B, (size_t)N * 2);
    if (EPI == EPI_RESIDUAL) hipMalloc(&R, (size_t)M * No * 2);
    fill_kernel<<<(int)(((long)M * K + 255) / 256), 256>>>(X, (long)M * K, 1u, 1.f);
    fill_kernel<<<(int)(((long)N * K + 255) / 256), 256>>>(W, (long)N * K, 2u, 0.03f);
    fill_kernel<<<(N + 255) / 256, 256>>>(B, N, 3u, 1.f);
    if (R) fill_kernel<<<(int)(((long)M * No + 255) / 256), 256>>>(R, (long)M * No, 4u, 1.f);
    const int grid = 256, TILES = 24;
    long long* dbg;
    const size_t dbg_n = (size_t)grid * 2 * TILES * 12;
    hipMalloc(&dbg, dbg_n * 8);
    GemmArgs<bf16_t, bf16_t> a{X, (long)K, W, (long)K, C, (long)No, B, R, (long)No, M, N, K};
    const int tm = cdiv(M, 256), tn = cdiv(N, 256);
    a.swz_n = cdiv(tn, cdiv(tn, 8));
    a.swz_m = std::max(1, 32 / a.swz_n);
    auto kern = gemm_nt_p8p_kernel<bf16_t, bf16_t, EPI>;
    const size_t lds = (size_t)2 * 512 * 128 + 8 * 4096;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(dbg, 0, dbg_n * 8);
        a.dbg = dbg;
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, a);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<long long> h(dbg_n);
    hipMemcpy(h.data(), dbg, dbg_n * 8, hipMemcpyDeviceToHost);
    const int tiles_per_wg = cdiv(tm * tn, grid);
    printf("%s M=%d N=%d K=%d: %.1f us, %.1f TFLOP/s, %d tiles, ~%d per workgroup, %.1f us per tile-round\n", name, M, N, K, ms * 1e3,
           2.0 * M * N * K / ms / 1e9, tm * tn, tiles_per_wg, ms * 1e3 / tiles_per_wg);
    const char* seg[8] = {"phases 0-3 (no wait)", "phase 4 (first counted wait)", "phases 5-7", "middle K-tiles", "last two K-tiles", "epilogue pass 1",
                          "vmcnt(0)", "staging + stores"};
    for (int g = 0; g < 2; ++g) {
        double sum[8] = {0}; long cnt = 0; double tile_sum = 0;
        for (int wg = 0; wg < grid; ++wg)
            for (int t = 1; t < std::min(TILES, tiles_per_wg - 1); ++t) {          // skip the first tile (prologue) and the last
                const long long* s = &h[(((size_t)wg * 2 + g) * TILES + t) * 12];
                if (!s[0] || !s[8]) continue;
                for (int i = 0; i < 8; ++i) sum[i] += (double)(s[i + 1] - s[i]);
                tile_sum += (double)(s[8] - s[0]);
                ++cnt;
            }
        if (!cnt) continue;
        // microseconds by proportion: a tile-round of the launch (event time / tiles per workgroup) over the mean stamped tile
        const double us_per_cycle = (ms * 1e3 / tiles_per_wg) / (tile_sum / cnt);
        printf("  wave group %d (%ld tiles): tile %.0f cycles (~%.0f cycles per us)\n", g, cnt, tile_sum / cnt, 1.0 / us_per_cycle);
        for (int i = 0; i < 8; ++i) printf("    %-30s %8.0f cycles  %6.2f us\n", seg[i], sum[i] / cnt, sum[i] / cnt * us_per_cycle);
    }
    hipFree(X); hipFree(W); hipFree(C); hipFree(B); if (R) hipFree(R); hipFree(dbg);
}

// The detector's 3 x 3 convolutions on the one-tile-per-workgroup 2-stage kernel (implicit GEMM, direct-to-LDS gather): stamps at kernel
// entry, in front of the K loop, behind it, and at the end.
static void run_conv(const char* name, int B, int H, int W, int Cin, int Cout, int stride) {
    const int Ho = H / stride, Wo = W / stride, M = B * Ho * Wo, K = ((9 * Cin + 63) / 64) * 64;
    bf16_t *X, *Wt, *C, *Bi, *Z;
    hipMalloc(&X, (size_t)B * H * W * Cin * 2); hipMalloc(&Wt, (size_t)Cout * K * 2); hipMalloc(&C, (size_t)M * Cout * 2); hipMalloc(&Bi, (size_t)Cout * 2);
    hipMalloc(&Z, 256); hipMemset(Z, 0, 256);
    fill_kernel<<<(int)(((long)B * H * W * Cin + 255) / 256), 256>>>(X, (long)B * H * W * Cin, 5u, 1.f);
    fill_kernel<<<(int)(((long)Cout * K + 255) / 256), 256>>>(Wt, (long)Cout * K, 6u, 0.05f);
    fill_kernel<<<(Cout + 255) / 256, 256>>>(Bi, Cout, 7u, 1.f);
    GemmArgs<bf16_t, bf16_t> a{nullptr, 0, Wt, (long)K, C, (long)Cout, Bi, nullptr, (long)Cout, M, Cout, K};
    a.conv_in = X; a.conv_zero = Z; a.cH = H; a.cW = W; a.cCin = Cin; a.cHo = Ho; a.cWo = Wo; a.cKW = 3; a.cStride = stride; a.cPad = 1; a.cTaps = 9;
    a.fd_hw = make_fastdiv((unsigned)(Ho * Wo)); a.fd_wo = make_fastdiv((unsigned)Wo);
    a.cv_m1 = Cin >= 64 ? (65536u + Cin / 64 - 1) / (Cin / 64) : 0u; a.cv_m2 = (65536u + 2) / 3; a.cv_rowskip = (W - 3) * Cin * 2;
    const int tm = cdiv(M, 256), tn = cdiv(Cout, 256);
    a.swz_n = cdiv(tn, cdiv(tn, 8)); a.swz_m = std::max(1, 32 / a.swz_n);
    const int tiles = cdiv(tm * tn, 256) * 256;
    long long* dbg;
    hipMalloc(&dbg, (size_t)tiles * 4 * 8);
    auto kern = gemm_nt_kernel<bf16_t, bf16_t, 256, 256, 4, 2, EPI_HARDSWISH, false, 2, true>;
    const size_t lds = (size_t)2 * 512 * 128;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(dbg, 0, (size_t)tiles * 4 * 8);
        a.dbg = dbg;
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), lds, 0, a);
        hipEventRecord(e1); hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<long long> h((size_t)tiles * 4);
    hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost);
    double seg[3] = {0, 0, 0}; long cnt = 0;
    for (int t = 0; t < tiles; ++t) {
        const long long* s = &h[(size_t)t * 4];
        if (!s[0] || !s[3]) continue;
        for (int i = 0; i < 3; ++i) seg[i] += (double)(s[i + 1] - s[i]);
        ++cnt;
    }
    const int rounds = cdiv(tm * tn, 256);
    printf("%s: M=%d N=%d K=%d: %.1f us, %.1f TFLOP/s, %d tiles = %d rounds, %.1f us per round\n", name, M, Cout, K, ms * 1e3,
           2.0 * M * Cout * 9 * Cin / ms / 1e9, tm * tn, rounds, ms * 1e3 / rounds);
    if (cnt) printf("    per workgroup (cycles): setup %.0f   K loop %.0f (%d K-tiles, %.0f per K-tile)   epilogue %.0f   total %.0f\n", seg[0] / cnt, seg[1] / cnt,
                    K / 64, seg[1] / cnt / (K / 64), seg[2] / cnt, (seg[0] + seg[1] + seg[2]) / cnt);
    hipFree(X); hipFree(Wt); hipFree(C); hipFree(Bi); hipFree(Z); hipFree(dbg);
}

// The same convolutions on the persistent 8-phase loop (gather in the request stream; odd K-tile counts get a virtual zero K-tile).
template <int MODE = 1>      // 1: Cin % 64 == 0 (a K-tile is one tap), 2: Cin == 32 (two taps per K-tile)
static void run_conv_p8p(const char* name, int B, int H, int W, int Cin, int Cout, int stride) {
    const int Ho = H / stride, Wo = W / stride, M = B * Ho * Wo, K = ((9 * Cin + 63) / 64) * 64;
    bf16_t *X, *Wt, *C, *Bi, *Z;
    hipMalloc(&X, (size_t)B * H * W * Cin * 2); hipMalloc(&Wt, (size_t)Cout * K * 2); hipMalloc(&C, (size_t)M * Cout * 2); hipMalloc(&Bi, (size_t)Cout * 2);
    hipMalloc(&Z, 256); hipMemset(Z, 0, 256);
    fill_kernel<<<(int)(((long)B * H * W * Cin + 255) / 256), 256>>>(X, (long)B * H * W * Cin, 5u, 1.f);
    fill_kernel<<<(int)(((long)Cout * K + 255) / 256), 256>>>(Wt, (long)Cout * K, 6u, 0.05f);
    fill_kernel<<<(Cout + 255) / 256, 256>>>(Bi, Cout, 7u, 1.f);
    GemmArgs<bf16_t, bf16_t> a{nullptr, 0, Wt, (long)K, C, (long)Cout, Bi, nullptr, (long)Cout, M, Cout, K};
    a.conv_in = X; a.conv_zero = Z; a.cH = H; a.cW = W; a.cCin = Cin; a.cHo = Ho; a.cWo = Wo; a.cKW = 3; a.cStride = stride; a.cPad = 1; a.cTaps = 9;
    a.fd_hw = make_fastdiv((unsigned)(Ho * Wo)); a.fd_wo = make_fastdiv((unsigned)Wo);
    a.cv_m1 = Cin >= 64 ? (65536u + Cin / 64 - 1) / (Cin / 64) : 0u; a.cv_m2 = (65536u + 2) / 3; a.cv_rowskip = (W - 3) * Cin * 2;
    const int tm = cdiv(M, 256), tn = cdiv(Cout, 256);
    a.swz_n = cdiv(tn, cdiv(tn, 8)); a.swz_m = std::max(1, 32 / a.swz_n);
    const int grid = 256, TILES = 24;
    long long* dbg;
    const size_t dbg_n = (size_t)grid * 2 * TILES * 12;
    hipMalloc(&dbg, dbg_n * 8);
    auto kern = gemm_nt_p8p_kernel<bf16_t, bf16_t, EPI_HARDSWISH, MODE>;
    const size_t lds = (size_t)2 * 512 * 128 + 8 * 4096;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(dbg, 0, dbg_n * 8);
        a.dbg = dbg;
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, a);
        hipEventRecord(e1); hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<long long> h(dbg_n);
    hipMemcpy(h.data(), dbg, dbg_n * 8, hipMemcpyDeviceToHost);
    const int tiles_per_wg = cdiv(tm * tn, grid);
    printf("%s [persistent]: M=%d N=%d K=%d: %.1f us, %.1f TFLOP/s, %d tiles, %.1f us per tile-round\n", name, M, Cout, K, ms * 1e3,
           2.0 * M * Cout * 9 * Cin / ms / 1e9, tm * tn, ms * 1e3 / tiles_per_wg);
    const char* seg[8] = {"phases 0-3 (no wait)", "phase 4 (first counted wait)", "phases 5-7", "middle K-tiles", "last two K-tiles", "epilogue pass 1",
                          "vmcnt(0)", "staging + stores"};
    for (int g = 0; g < 2; ++g) {
        double sum[8] = {0}; long cnt = 0; double tile_sum = 0;
        for (int wg = 0; wg < grid; ++wg)
            for (int t = 1; t < std::min(TILES, tiles_per_wg - 1); ++t) {
                const long long* st = &h[(((size_t)wg * 2 + g) * TILES + t) * 12];
                if (!st[0] || !st[8]) continue;
                for (int i = 0; i < 8; ++i) sum[i] += (double)(st[i + 1] - st[i]);
                tile_sum += (double)(st[8] - st[0]); ++cnt;
            }
        if (!cnt) continue;
        printf("  wave group %d (%ld tiles): tile %.0f cycles:", g, cnt, tile_sum / cnt);
        for (int i = 0; i < 8; ++i) printf("  %s %.0f", seg[i], sum[i] / cnt);
        // inside "last two K-tiles": its first two phases, the next tile's request state (set_req + set_conv), the phase behind it, the rest
        double sub[4] = {0}; long sc = 0;
        for (int wg = 0; wg < grid; ++wg)
            for (int t = 1; t < std::min(TILES, tiles_per_wg - 1); ++t) {
                const long long* st = &h[(((size_t)wg * 2 + g) * TILES + t) * 12];
                if (!st[4] || !st[5] || !st[9] || !st[10] || !st[11]) continue;
                sub[0] += (double)(st[9] - st[4]); sub[1] += (double)(st[10] - st[9]); sub[2] += (double)(st[11] - st[10]); sub[3] += (double)(st[5] - st[11]); ++sc;
            }
        if (sc) printf("  | tail: phases 0-1 %.0f, next tile's request state %.0f, phase 2 %.0f, phases 3-7 %.0f", sub[0] / sc, sub[1] / sc, sub[2] / sc, sub[3] / sc);
        printf("\n");
    }
    hipFree(X); hipFree(Wt); hipFree(C); hipFree(Bi); hipFree(Z); hipFree(dbg);
}

int main() {
    run_conv_p8p<2>("det op 4: 3x3 s2, 32 -> 512 ch, 512^2 -> 256^2 x 16 pages", 16, 512, 512, 32, 512, 2);
    run_conv("det op 4: 3x3 s2, 32 -> 512 ch, 512^2 -> 256^2 x 16 pages", 16, 512, 512, 32, 512, 2);
    run_conv_p8p("det op 6: 3x3 s1, 64 -> 256 ch, 256^2 x 16 pages", 16, 256, 256, 64, 256, 1);
    run_conv_p8p("det op 10: 3x3 s1, 128 -> 512 ch, 128^2 x 16 pages", 16, 128, 128, 128, 512, 1);
    run_conv("det op 6: 3x3 s1, 64 -> 256 ch, 256^2 x 16 pages", 16, 256, 256, 64, 256, 1);
    run_conv("det op 10: 3x3 s1, 128 -> 512 ch, 128^2 x 16 pages", 16, 128, 128, 128, 512, 1);
    run<EPI_BIAS>("enc qkv (bias)", 46460, 3840, 1280);
    run<EPI_SWIGLU>("enc gate|up (SwiGLU)", 46460, 6912, 1280);
    run<EPI_RESIDUAL>("enc proj (+ residual)", 46460, 1280, 1280);
    run<EPI_BIAS>("square 8k", 8192, 8192, 8192);
    return 0;
}
