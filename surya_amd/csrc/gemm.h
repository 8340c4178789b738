// MFMA "NT" GEMM for gfx950:  C[M,N] = X[M,K] . W[N,K]^T  (+ fused epilogue), fp32 accumulate.
//
// Both operands are K-contiguous (activations row-major, weights in nn.Linear layout), so both MFMA
// fragments are 16-byte `ds_read_b128` reads of the same LDS geometry. Tiles are expressed in 128-byte
// K-rows: 64 bf16 or 32 fp32 elements, which makes the bf16 path (v_mfma_f32_16x16x32_bf16) and the exact
// fp32 path (4 x v_mfma_f32_16x16x4_f32 per 16-byte chunk, k-permuted identically on both operands) share
// every address computation.
//
// The weight fragment is the FIRST MFMA operand, so the 16x16 result is D[n][m]: a lane owns 4 CONSECUTIVE
// output columns n of one row m -> 8/16-byte stores, float4 bias loads, and the (gate, up) pairs of the
// interleaved SwiGLU weight land in one lane.
//
// LDS: [rows][8 chunks of 16 B], chunk index XOR-swizzled with (row & 7); register-staged double buffer,
// one barrier per K-tile.
#pragma once
#include "common.h"

namespace sa {

enum GemmEpi { EPI_BIAS = 0, EPI_RESIDUAL = 1, EPI_GELU = 2, EPI_SWIGLU = 3, EPI_HARDSWISH = 4, EPI_RELU = 5 };

template <typename TI, typename TO>
struct GemmArgs {
    const TI* X; long ldx;       // [M, K]
    const TI* W; long ldw;       // [N, K]
    TO* C; long ldc;             // [M, N] (SwiGLU: [M, N/2])
    const TI* bias;              // [N] or nullptr
    const TO* R; long ldr;       // residual [M, N] (EPI_RESIDUAL), may alias C
    int M, N, K;                 // K % KE == 0, N % 4 == 0
};

template <typename TI> struct Mfma;
template <> struct Mfma<bf16_t> {
    __device__ static __forceinline__ void run(f32x4& acc, const uint4& w, const uint4& x) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), acc, 0, 0, 0);
    }
};
template <> struct Mfma<float> {
    __device__ static __forceinline__ void run(f32x4& acc, const uint4& w, const uint4& x) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.x), __uint_as_float(x.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.y), __uint_as_float(x.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.z), __uint_as_float(x.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.w), __uint_as_float(x.w), acc, 0, 0, 0);
    }
};

// BM x BN output tile per 256-thread workgroup; waves arranged WM x WN (WM*WN == 4).
template <typename TI, typename TO, int BM, int BN, int WM, int WN, int EPI>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmArgs<TI, TO> p) {
    constexpr int KE = Ty<TI>::KE;            // elements per 128-byte row
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int FM = WTM / 16, FN = WTN / 16;
    constexpr int XCH = BM * 8 / 256, WCH = BN * 8 / 256;   // 16-byte chunks per thread per tile
    static_assert(WM * WN == 4 && FM >= 1 && FN >= 1 && XCH >= 1 && WCH >= 1, "tile");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // layout: buf b: X tile [BM][128B] then W tile [BN][128B]
    constexpr int XBYTES = BM * 128, WBYTES = BN * 128, BUF = XBYTES + WBYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // global source pointers for this thread's staging chunks (rows clamped into range)
    const unsigned char* xsrc[XCH];
    const unsigned char* wsrc[WCH];
    int xdst[XCH], wdst[WCH];
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
        int id = tid + i * 256, row = id >> 3, c = id & 7;
        int gr = min(m0 + row, p.M - 1);
        xsrc[i] = reinterpret_cast<const unsigned char*>(p.X + (long)gr * p.ldx) + c * 16;
        xdst[i] = row * 128 + ((c ^ (row & 7)) << 4);
    }
#pragma unroll
    for (int i = 0; i < WCH; ++i) {
        int id = tid + i * 256, row = id >> 3, c = id & 7;
        int gr = min(n0 + row, p.N - 1);
        wsrc[i] = reinterpret_cast<const unsigned char*>(p.W + (long)gr * p.ldw) + c * 16;
        wdst[i] = XBYTES + row * 128 + ((c ^ (row & 7)) << 4);
    }

    f32x4 acc[FN][FM];
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int i = 0; i < FM; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / KE;
    uint4 xr[XCH], wr[WCH];
#pragma unroll
    for (int i = 0; i < XCH; ++i) xr[i] = *reinterpret_cast<const uint4*>(xsrc[i]);
#pragma unroll
    for (int i = 0; i < WCH; ++i) wr[i] = *reinterpret_cast<const uint4*>(wsrc[i]);
#pragma unroll
    for (int i = 0; i < XCH; ++i) *reinterpret_cast<uint4*>(smem + xdst[i]) = xr[i];
#pragma unroll
    for (int i = 0; i < WCH; ++i) *reinterpret_cast<uint4*>(smem + wdst[i]) = wr[i];
    __syncthreads();

    // fragment read offsets: row = base + (lane & 15), chunk = kk*4 + (lane >> 4)
    const int frow = lane & 15, fch = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const unsigned char* cur = smem + (kt & 1) * BUF;
        unsigned char* nxt = smem + ((kt + 1) & 1) * BUF;
        const bool more = (kt + 1) < nk;
        if (more) {
            const long koff = (long)(kt + 1) * 128;
#pragma unroll
            for (int i = 0; i < XCH; ++i) xr[i] = *reinterpret_cast<const uint4*>(xsrc[i] + koff);
#pragma unroll
            for (int i = 0; i < WCH; ++i) wr[i] = *reinterpret_cast<const uint4*>(wsrc[i] + koff);
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            uint4 xf[FM], wf[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                int row = wm * WTM + i * 16 + frow;
                xf[i] = *reinterpret_cast<const uint4*>(cur + row * 128 + (((kk * 4 + fch) ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                int row = wn * WTN + j * 16 + frow;
                wf[j] = *reinterpret_cast<const uint4*>(cur + XBYTES + row * 128 + (((kk * 4 + fch) ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int i = 0; i < FM; ++i) Mfma<TI>::run(acc[j][i], wf[j], xf[i]);
        }
        if (more) {
#pragma unroll
            for (int i = 0; i < XCH; ++i) *reinterpret_cast<uint4*>(nxt + xdst[i]) = xr[i];
#pragma unroll
            for (int i = 0; i < WCH; ++i) *reinterpret_cast<uint4*>(nxt + wdst[i]) = wr[i];
        }
        __syncthreads();
    }

    // epilogue: lane owns row m = .. + (lane & 15), columns n = .. + (lane >> 4) * 4 + {0..3}
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int m = m0 + wm * WTM + i * 16 + (lane & 15);
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int n = n0 + wn * WTN + j * 16 + (lane >> 4) * 4;
            if (n >= p.N) continue;
            float v[4] = {acc[j][i][0], acc[j][i][1], acc[j][i][2], acc[j][i][3]};
            if (p.bias) {
                float b[4];
                load4(p.bias + n, b);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += b[r];
            }
            if constexpr (EPI == EPI_SWIGLU) {
                // weight rows interleaved (gate_j, up_j): columns n..n+3 = g0,u0,g1,u1 -> outputs n/2, n/2+1
                store2(p.C + (long)m * p.ldc + (n >> 1), silu_f(v[0]) * v[1], silu_f(v[2]) * v[3]);
            } else {
                if constexpr (EPI == EPI_RESIDUAL) {
                    float r4[4];
                    load4(p.R + (long)m * p.ldr + n, r4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += r4[r];
                } else if constexpr (EPI == EPI_GELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = gelu_erf_f(v[r]);
                } else if constexpr (EPI == EPI_HARDSWISH) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = hardswish_f(v[r]);
                } else if constexpr (EPI == EPI_RELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                }
                store4(p.C + (long)m * p.ldc + n, v[0], v[1], v[2], v[3]);
            }
        }
    }
}

// Optional per-launch timing with HIP events on the launch stream (bench.py's `roofline` object): one record per
// tile configuration, accumulating launches, algorithmic FLOPs / bytes and event-measured milliseconds.
struct GemmProfiler {
    static constexpr int NCFG = 4, POOL = 8192;
    bool enabled = false;
    hipEvent_t ev[2 * POOL];
    bool have_events = false;
    int n = 0;
    int cfg_of[POOL];
    double flops_of[POOL], bytes_of[POOL];
    void start() {
        if (!have_events) {
            for (int i = 0; i < 2 * POOL; ++i) (void)hipEventCreate(&ev[i]);
            have_events = true;
        }
        n = 0; enabled = true;
    }
};
inline GemmProfiler& gemm_profiler() { static GemmProfiler p; return p; }
inline int gemm_cfg_id(int BM, int BN) { return BM == 128 ? 0 : (BM == 64 ? 1 : (BM == 32 ? 2 : 3)); }

template <typename TI, typename TO, int BM, int BN, int WM, int WN, int EPI>
static inline int launch_gemm_cfg(const GemmArgs<TI, TO>& a, hipStream_t s) {
    const int tiles = cdiv(a.M, BM) * cdiv(a.N, BN);
    constexpr size_t lds = (size_t)(BM + BN) * 128 * 2;
    auto kern = gemm_nt_kernel<TI, TO, BM, BN, WM, WN, EPI>;
    static bool attr_set = false;   // >64 KiB dynamic LDS needs the opt-in attribute; harmless below
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    GemmProfiler& pf = gemm_profiler();
    const bool prof = pf.enabled && pf.n < GemmProfiler::POOL;
    if (prof) (void)hipEventRecord(pf.ev[2 * pf.n], s);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), lds, s, a);
    if (prof) {
        (void)hipEventRecord(pf.ev[2 * pf.n + 1], s);
        pf.cfg_of[pf.n] = gemm_cfg_id(BM, BN);
        pf.flops_of[pf.n] = 2.0 * a.M * a.N * a.K;
        const double outn = (EPI == EPI_SWIGLU) ? a.N / 2 : a.N;
        pf.bytes_of[pf.n] = ((double)a.M * a.K + (double)a.N * a.K) * sizeof(TI) + (double)a.M * outn * sizeof(TO) +
                            (EPI == EPI_RESIDUAL ? (double)a.M * a.N * sizeof(TO) : 0.0);
        ++pf.n;
    }
    return (int)hipGetLastError();
}

// Tile choice: big tiles when the grid still fills 256 CUs, smaller ones for skinny problems (decode).
template <typename TI, typename TO, int EPI>
static inline int launch_gemm(const GemmArgs<TI, TO>& a, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0) return SA_OK;
    if (a.K % Ty<TI>::KE != 0 || a.N % 4 != 0) return SA_ERR_SHAPE;
    const long big = (long)cdiv(a.M, 128) * cdiv(a.N, 128);
    if (big >= 384) return launch_gemm_cfg<TI, TO, 128, 128, 2, 2, EPI>(a, s);
    const long mid = (long)cdiv(a.M, 64) * cdiv(a.N, 64);
    if (mid >= 256 || a.M > 32) return launch_gemm_cfg<TI, TO, 64, 64, 2, 2, EPI>(a, s);
    return launch_gemm_cfg<TI, TO, 32, 64, 1, 4, EPI>(a, s);
}

}  // namespace sa
