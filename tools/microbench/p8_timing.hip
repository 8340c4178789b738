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

int main() {
    run<EPI_BIAS>("enc qkv (bias)", 46460, 3840, 1280);
    run<EPI_SWIGLU>("enc gate|up (SwiGLU)", 46460, 6912, 1280);
    run<EPI_RESIDUAL>("enc proj (+ residual)", 46460, 1280, 1280);
    run<EPI_BIAS>("square 8k", 8192, 8192, 8192);
    return 0;
}
