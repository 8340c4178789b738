// What bounds the decode-regime GEMMs (M = 256)? Load-only emulation of the NT GEMM's global traffic for C[256, N] =
// X[256, K] W[N, K]^T with N = 10240, K = 1280 (the decoder gate|up projection, 26 MB of weights): every workgroup walks K
// in 128-byte steps and loads BM rows of X and BN rows of W per step into registers (no LDS, no MFMA), tile_n fastest
// like the real kernel. Compared per tile shape: time, GB/s of unique bytes, GB/s of bytes actually requested by the CUs.
// A second table streams W alone row-major vs tile-major (one contiguous run per workgroup) to test DRAM-page locality.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int BM, int BN>
__global__ __launch_bounds__(256) void tile_loads(const unsigned char* __restrict__ x, const unsigned char* __restrict__ w,
                                                  unsigned int* __restrict__ out, int K2, int nk, int tiles_n) {
    constexpr int XL = BM / 32, WL = BN / 32;          // 16-byte loads per thread per k-step
    const int tid = threadIdx.x, tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    const unsigned char* xp = x + ((long)tm * BM + (tid >> 3)) * K2 + (tid & 7) * 16;
    const unsigned char* wp = w + ((long)tn * BN + (tid >> 3)) * K2 + (tid & 7) * 16;
    u32x4 acc = {0, 0, 0, 0};
#pragma unroll 2
    for (int k = 0; k < nk; ++k) {
#pragma unroll
        for (int i = 0; i < XL; ++i) acc += *reinterpret_cast<const u32x4*>(xp + (long)i * 32 * K2 + k * 128);
#pragma unroll
        for (int i = 0; i < WL; ++i) acc += *reinterpret_cast<const u32x4*>(wp + (long)i * 32 * K2 + k * 128);
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[blockIdx.x] = acc[0];
}

template <int MODE>
__global__ __launch_bounds__(256) void stream(const unsigned char* __restrict__ w, unsigned int* __restrict__ out, int K2, int nk) {
    const int tid = threadIdx.x, blk = blockIdx.x;
    u32x4 acc = {0, 0, 0, 0};
    if (MODE == 0) {
        const unsigned char* p0 = w + ((long)blk * 64 + (tid >> 3)) * K2 + (tid & 7) * 16;
        const unsigned char* p1 = p0 + 32L * K2;
#pragma unroll 4
        for (int k = 0; k < nk; ++k) { acc += *reinterpret_cast<const u32x4*>(p0 + k * 128); acc += *reinterpret_cast<const u32x4*>(p1 + k * 128); }
    } else {
        const unsigned char* p0 = w + (long)blk * 64 * K2 + tid * 16;
#pragma unroll 4
        for (int k = 0; k < nk; ++k) { acc += *reinterpret_cast<const u32x4*>(p0 + k * 8192); acc += *reinterpret_cast<const u32x4*>(p0 + k * 8192 + 4096); }
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[blockIdx.x] = acc[0];
}

static const int M = 256, N = 10240, K = 1280, K2 = K * 2, NK = K / 64;
static unsigned char *g_w, *g_x; static unsigned int* g_out; static hipEvent_t e0, e1;

template <int BM, int BN> void run_tile() {
    const int tiles_n = N / BN, blocks = (M / BM) * tiles_n;
    float best = 1e9;
    for (int it = 0; it < 20; ++it) {
        const unsigned char* wp = g_w + (size_t)(it % 8) * N * K2;      // rotate copies: weights come from HBM, not MALL
        hipEventRecord(e0);
        hipLaunchKernelGGL((tile_loads<BM, BN>), dim3(blocks), dim3(256), 0, 0, g_x, wp, g_out, K2, NK, tiles_n);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (it >= 4 && ms < best) best = ms;
    }
    const double uniq = (double)(M + N) * K2, req = (double)blocks * (BM + BN) * K2;
    printf("tile %3dx%-3d  %4d workgroups: %6.2f us  unique %5.0f GB/s  requested %6.1f MB -> %5.0f GB/s\n", BM, BN, blocks, best * 1e3,
           uniq / best / 1e6, req / 1e6, req / best / 1e6);
}

int main() {
    hipMalloc(&g_w, (size_t)N * K2 * 8); hipMalloc(&g_x, (size_t)M * K2); hipMalloc(&g_out, 1 << 20);
    hipMemset(g_w, 1, (size_t)N * K2 * 8); hipMemset(g_x, 1, (size_t)M * K2);
    hipEventCreate(&e0); hipEventCreate(&e1);
    run_tile<64, 64>(); run_tile<128, 64>(); run_tile<64, 128>(); run_tile<128, 128>(); run_tile<256, 64>(); run_tile<256, 32>();
    run_tile<256, 128>(); run_tile<64, 256>(); run_tile<32, 128>();
    for (int mode = 0; mode < 2; ++mode) {
        float best = 1e9;
        for (int it = 0; it < 20; ++it) {
            const unsigned char* wp = g_w + (size_t)(it % 8) * N * K2;
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(stream<0>, dim3(N / 64), dim3(256), 0, 0, wp, g_out, K2, NK);
            else hipLaunchKernelGGL(stream<1>, dim3(N / 64), dim3(256), 0, 0, wp, g_out, K2, NK);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (it >= 4 && ms < best) best = ms;
        }
        printf("W only, %s: %.2f us  %.0f GB/s\n", mode ? "tile-major" : "row-major ", best * 1e3, (double)N * K2 / best / 1e6);
    }
    return 0;
}
