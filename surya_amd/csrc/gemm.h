// MFMA "NT" GEMM for gfx950:  C[M,N] = X[M,K] . W[N,K]^T  (+ fused epilogue), fp32 accumulate.
//
// Both operands are K-contiguous (activations row-major, weights in nn.Linear layout), so both MFMA
// fragments are 16-byte `ds_read_b128` reads of the same LDS geometry. Tiles are expressed in 128-byte
// K-rows: 64 bf16 or 32 fp32 elements, which makes the bf16 path (v_mfma_f32_32x32x16_bf16) and the exact
// fp32 path (4 x v_mfma_f32_32x32x2_f32 per 16-byte chunk, k-permuted identically on both operands) share
// every address computation.
//
// The weight fragment is the FIRST MFMA operand, so the 32x32 result is D[n][m]: a lane owns 4 CONSECUTIVE
// output columns n of one row m -> 8/16-byte stores, float4 bias loads, and the (gate, up) pairs of the
// interleaved SwiGLU weight / the rotate_half pairs of the pair-interleaved vision q,k rows land in one lane.
//
// Staging, two generations (template parameter GLDS):
//   GLDS = 0  K-tiles are fetched TWO tiles ahead into alternating register sets and written to a double-buffered,
//             XOR-swizzled LDS image one iteration before use (one barrier per K-tile); kept for tile shapes whose
//             staging does not divide into whole 8-row groups per wave.
//   GLDS >= 2 global_load_lds_dwordx4 moves 8 tile rows per instruction straight into LDS (swizzle on the source
//             address): 2 stages unrolled for the big tiles, a ring of GLDS stages with raw s_barrier + vmcnt(N) for the
//             split-K decode tiles.
// Main loop: the ds_read_b128 fragment reads of K-step kk+1 are interleaved with the MFMAs of step kk.
//
// Tile menu (launch_gemm): 256x256 (8 waves 4x2, 64x128 per wave) when whole rounds of 256 workgroups are cheaper than
// rounds of 512 128x128 tiles; 128x128 (4 waves 2x2); 64x64 / 128x64 / 64x32 for the decode regime M <= 256 (many light
// workgroups beat tall 256-row tiles there, tools/microbench); split-K for the skinny decode projections.
// Epilogues: bias / residual / GELU / SwiGLU / Hardswish / ReLU / greedy-argmax partials / vision RoPE, staged through LDS
// into coalesced 16-byte row stores.
#pragma once
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "fastdiv.h"

namespace sa {

enum GemmEpi { EPI_BIAS = 0, EPI_RESIDUAL = 1, EPI_GELU = 2, EPI_SWIGLU = 3, EPI_HARDSWISH = 4, EPI_RELU = 5, EPI_ARGMAX = 6, EPI_ROPE = 7,
               EPI_GEGLU = 8 /* gelu_tanh(gate) * up, rows interleaved like SWIGLU (ADETR decoder MLP, adetr/decoder.py:331-344) */ };

template <typename TI, typename TO>
struct GemmArgs {
    const TI* X; long ldx;       // [M, K]
    const TI* W; long ldw;       // [N, K]
    TO* C; long ldc;             // [M, N] (SwiGLU: [M, N/2])
    const TI* bias;              // [N] or nullptr
    const TO* R; long ldr;       // residual [M, N] (EPI_RESIDUAL), may alias C
    int M, N, K;                 // K % KE == 0, N % 4 == 0
    int splitk = 1;              // > 1: K is cut into `splitk` slices, raw fp32 partial sums go to `part`
    float* part = nullptr;       // [splitk][M][N] fp32 (no bias / epilogue applied; the consumer kernel combines them)
    int swz_m = 0, swz_n = 0;    // > 0: XCD-aware rasterisation in super-tiles of swz_m x swz_n output tiles (set by the launcher)
    int mgroup = 0;              // 1: the M-tiles of one W tile take consecutive places in ONE XCD's queue (lm_head above 256 rows: round 6)
    // EPI_ARGMAX (TO = float): C is not written; instead one float4 {max, bits of the first argmax column, sum exp(v - max), 0}
    // per (row, tile column) goes to amax[m * cdiv(N, bn_used) + tile_n]; the launcher reports the tile width it chose.
    float4* amax = nullptr;
    mutable int bn_used = 0;
    // EPI_ROPE (vision qkv projection): rotary embedding of the q and k columns (n < rope_cols) in the epilogue. The weight
    // rows of every head are stored PAIR-INTERLEAVED (new column 2j = old j, 2j + 1 = old j + D/2), so the lane that owns
    // four consecutive columns holds two complete rotate_half pairs; rope[m * (D/2) + j] = (cos, sin) of token m, pair j.
    const float2* rope = nullptr;
    int rope_cols = 0, rope_D = 0;
    // Implicit-GEMM convolution (direct-to-LDS tiles, template parameter CONV): X is not a matrix but an NHWC tensor
    // conv_in[B][cH][cW][cCin]; row m of the GEMM is output pixel (b, oy, ox), K index k = (ky * cKW + kx) * cCin + ci (the weight
    // rows are laid out the same way, zero padded to K), and the 16-byte chunk a lane stages comes from input pixel
    // (oy * cStride - cPad + ky, ox * cStride - cPad + kx) -- or from the 16 zero bytes at conv_zero outside the image / past the
    // last tap: global_load_lds takes a per-lane source address, so the gather costs address arithmetic only (no im2col buffer,
    // no staging registers). M = B * cHo * cWo, N = Cout, C / R = the NHWC output / residual.
    const TI* conv_in = nullptr;
    const TI* conv_zero = nullptr;
    int cH = 0, cW = 0, cCin = 0, cHo = 0, cWo = 0, cKW = 0, cStride = 0, cPad = 0, cTaps = 0;
    int conv_lean = 1;           // 1: per-(row, tap) gather state precomputed once per workgroup when K-tiles align with taps (0 = round-4 gather, for A/B)
    FastDiv fd_hw, fd_wo;        // CONV on the persistent loop: per-row divisors cHo * cWo and cWo (set by the launcher)
    // ... and its wave-uniform tap math, per request and therefore on the K loop's critical path: K-tile kt covers channels of tap
    // kt / (cCin / 64) = (kt * cv_m1) >> 16, filter row tap / cKW = (tap * cv_m2) >> 16 (cv_m = ceil(2^16 / d): exact for kt < 4096 and
    // d <= 16, tools checked in Python); the K-tile's byte offset from the row's first tap is kt * 128 + ky * cv_rowskip (a filter row's
    // cKW * cCin channels are contiguous in NHWC; cv_rowskip = (cW - cKW) * cCin * sizeof(TI) steps to the next image row)
    unsigned cv_m1 = 0, cv_m2 = 0;
    int cv_rowskip = 0;
#ifndef SA_P8_TIMING
#define SA_P8_TIMING 0                // tools/microbench/p8_timing.hip: s_memtime stamps of the persistent 8-phase loop's segments into `dbg`
#endif
#if SA_P8_TIMING
    long long* dbg = nullptr;        // [workgroup][2 waves (0, 4)][SA_P8_TIMING_TILES][12] cycle stamps
#endif
};

// 32x32 MFMA tiles: per flop they need half the LDS fragment traffic of 16x16 tiles (the 16x16 version of this kernel
// was LDS-bound at ~690 TF/s, r01 profile). Fragment of either operand: lane l holds 16 bytes = K-chunk (ks*2 + (l >> 5))
// of row (l & 31); bf16: one v_mfma_f32_32x32x16_bf16; fp32: four v_mfma_f32_32x32x2_f32 on the chunk's 4 floats
// (k-permuted identically on both operands, so the sum is unchanged and exact).
template <typename TI> struct Mfma;
template <> struct Mfma<bf16_t> {
    __device__ static __forceinline__ void run(f32x16& acc, const u32x4& w, const u32x4& x) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), acc, 0, 0, 0);
    }
};
template <> struct Mfma<float> {
    __device__ static __forceinline__ void run(f32x16& acc, const u32x4& w, const u32x4& x) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w[0]), __uint_as_float(x[0]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w[1]), __uint_as_float(x[1]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w[2]), __uint_as_float(x[2]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w[3]), __uint_as_float(x[3]), acc, 0, 0, 0);
    }
};

// BM x BN output tile per workgroup of WM x WN waves.
template <typename TI, typename TO, int BM, int BN, int WM, int WN, int EPI, bool SPLIT = false, int GLDS = 0, bool CONV = false, int WAUX = 0>
__global__ __launch_bounds__(64 * WM * WN) void gemm_nt_kernel(GemmArgs<TI, TO> p) {
    static_assert(!CONV || (GLDS == 2 && !SPLIT), "the convolution gather exists for the 2-stage direct-to-LDS loop");
    constexpr int NT = 64 * WM * WN;
    constexpr int KE = Ty<TI>::KE;            // elements per 128-byte row
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int FM = WTM / 32, FN = WTN / 32;
    // 16-byte chunks per thread per tile. When a tile has fewer chunks than threads (256x32: 256 W chunks, 512 threads)
    // the upper threads duplicate the lower ones' chunk (same data to the same LDS slot) instead of being predicated:
    // predicated loads cost an exec-mask branch + vmcnt(0) each and demote the staging registers to scratch.
    constexpr int XCH = (BM * 8 + NT - 1) / NT, WCH = (BN * 8 + NT - 1) / NT;
    constexpr int XMOD = XCH * NT > BM * 8 ? BM * 8 : XCH * NT, WMOD = WCH * NT > BN * 8 ? BN * 8 : WCH * NT;
    static_assert(FM >= 1 && FN >= 1 && WTM % 32 == 0 && WTN % 32 == 0, "tile");
    static_assert((XCH == 1 || (BM * 8) % NT == 0) && (WCH == 1 || (BN * 8) % NT == 0), "staging split");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // layout: buf b: X tile [BM][128B] then W tile [BN][128B]
    constexpr int XBYTES = BM * 128, WBYTES = BN * 128, BUF = XBYTES + WBYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int tiles_n = (p.N + BN - 1) / BN;
    int tile_m = (int)blockIdx.x / tiles_n, tile_n = (int)blockIdx.x % tiles_n, ks = 0;
    if constexpr (SPLIT) {
        // The tiles_m workgroups that stream the same (W tile, K slice) must sit on ONE XCD so that slice crosses the fabric
        // once: workgroup b runs on XCD b % 8, so (tile_n, ks) pairs are dealt round-robin to XCDs and the M-tiles of a pair
        // take consecutive positions in that XCD's queue. With tile_m slowest the four M-tiles landed on four XCDs and the
        // split-K GEMMs fetched 2.7x their unique bytes (r01 FETCH_SIZE profile).
        const int tiles_m = (p.M + BM - 1) / BM;
        const int x = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3;
        const int pair = (j / tiles_m) * 8 + x;
        if (pair >= tiles_n * p.splitk) return;
        tile_m = j % tiles_m;
        tile_n = pair / p.splitk;
        ks = pair % p.splitk;
    }
    if (!SPLIT && p.mgroup) {
        // decode regime above 256 rows (round 6): every 256-row block of the batch multiplies the same W tile, so the blocks of a tile are dealt
        // to one XCD back to back (the split-K mapping with one slice): the tile crosses the fabric once per step, not once per row block
        const int tiles_m = (p.M + BM - 1) / BM;
        const int x = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3;
        const int pair = (j / tiles_m) * 8 + x;
        if (pair >= tiles_n) return;
        tile_m = j % tiles_m;
        tile_n = pair;
    }
    if (!SPLIT && p.swz_n > 0) {
        // Workgroup b runs on XCD b % 8 (observed dispatch order; used for speed only). Each XCD has a private 4 MiB L2,
        // so the ~64 workgroups resident on one XCD should form a compact patch of output tiles: they then share a few
        // X-slabs and W-slabs through that L2 instead of each pulling its own pair from MALL/HBM (n-fastest order gave ~55
        // distinct slabs per 64 tiles and capped the 128x128 GEMM at ~0.65 PF/s, r01 profile).
        // Tiles are numbered in super-tile order (super-rows of swz_m tile rows, cut into super-columns of swz_n tile
        // columns, edge super-tiles shrunk to what exists, so only valid tiles are numbered); consecutive runs of GRP
        // numbers are dealt to the XCDs round-robin. Every XCD gets the same number of tiles: an earlier version dealt
        // whole super-tiles including the clipped ones, and with N = 1280 (10 tile columns = one full + one quarter
        // super-column) the odd XCDs drew only quarter super-tiles and the launch ran at 63 % (r01 microbench).
        const int x = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3;
        constexpr int GRP = ((BM + BN) * 128 * (GLDS > 2 && GLDS != 8 ? GLDS : 2) > 80 * 1024) ? 32 : 64;     // workgroups resident on one XCD (32 CUs x 1 or 2, by LDS)
        const int idx = ((j / GRP) * 8 + x) * GRP + (j % GRP);
        const int tiles_m = (p.M + BM - 1) / BM;
        if (idx >= tiles_m * tiles_n) return;
        const int per_row = p.swz_m * tiles_n;                          // tiles in a full super-row
        const int sr = idx / per_row, rem = idx - sr * per_row;
        const int h = min(p.swz_m, tiles_m - sr * p.swz_m);
        const int sc = rem / (h * p.swz_n), rem2 = rem - sc * h * p.swz_n;
        const int w = min(p.swz_n, tiles_n - sc * p.swz_n);
        tile_m = sr * p.swz_m + rem2 / w;
        tile_n = sc * p.swz_n + rem2 % w;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
#if SA_P8_TIMING
#define SA_STAMP1(IDX) { if (p.dbg && tid == 0) p.dbg[(long)blockIdx.x * 4 + (IDX)] = (long long)__builtin_readcyclecounter(); }
#else
#define SA_STAMP1(IDX) {}
#endif
    SA_STAMP1(0);
    const int nk_all = p.K / KE;
    const int kt_begin = SPLIT ? (int)((long)ks * nk_all / p.splitk) : 0;
    const int kt_end = SPLIT ? (int)((long)(ks + 1) * nk_all / p.splitk) : nk_all;

    f32x16 acc[FN][FM];
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

    const int nk = kt_end - kt_begin;
    const int last = nk - 1;
    const int pairs = nk >> 1;
    const int frow = lane & 31, fch = lane >> 5;     // fragment: row = base + (lane & 31), chunk = ks*2 + (lane >> 5)
#define SA_FRAGS(XF, WF, KK)                                                                                   \
    {                                                                                                          \
        _Pragma("unroll") for (int i = 0; i < FM; ++i) {                                                       \
            const int row = wm * WTM + i * 32 + frow;                                                          \
            XF[i] = *reinterpret_cast<const u32x4*>(cur_ + row * 128 + ((((KK) * 2 + fch) ^ ((row >> 1) & 7)) << 4)); \
        }                                                                                                      \
        _Pragma("unroll") for (int j = 0; j < FN; ++j) {                                                       \
            const int row = wn * WTN + j * 32 + frow;                                                          \
            WF[j] = *reinterpret_cast<const u32x4*>(cur_ + XBYTES + row * 128 + ((((KK) * 2 + fch) ^ ((row >> 1) & 7)) << 4)); \
        }                                                                                                      \
    }
#define SA_MFMAS(XF, WF)                                   \
    _Pragma("unroll") for (int j = 0; j < FN; ++j)         \
        _Pragma("unroll") for (int i = 0; i < FM; ++i) Mfma<TI>::run(acc[j][i], WF[j], XF[i]);
    // Fragments of K-step kk+1 are read from LDS before the MFMAs of step kk are issued, so one wave keeps the matrix
    // pipe busy without relying on a second resident wave to cover its ds_read latency.
#ifndef SA_INTERLEAVE
#define SA_INTERLEAVE 1
#endif
    // Accumulator-heavy tiles (the 256x320 lm_head tile: 160 accumulator registers per lane of 256) cannot afford two fragment sets:
    // one set, hipcc schedules; the second wave of each SIMD covers the fragment-read latency.
    constexpr bool LEAN = FM * FN * 16 > 128;
#define SA_COMPUTE_LEAN(CURP)                              \
    {                                                      \
        const unsigned char* cur_ = (CURP);                \
        _Pragma("unroll") for (int kk_ = 0; kk_ < 4; ++kk_) { \
            u32x4 xf_[FM], wf_[FN];                        \
            SA_FRAGS(xf_, wf_, kk_);                       \
            __builtin_amdgcn_sched_barrier(0);             \
            SA_MFMAS(xf_, wf_);                            \
            __builtin_amdgcn_sched_barrier(0);             \
        }                                                  \
    }
#if SA_INTERLEAVE
    // The LDS fragment reads of step kk+1 are placed BETWEEN the MFMAs of step kk (one read per MFMA) instead of in a burst
    // ahead of them: sched_group_barrier(mask, count, id), mask 0x100 = DS read, 0x8 = MFMA. +2-5 % on every shape over the
    // burst order below (-DSA_INTERLEAVE=0), e.g. 8k^3 1151 -> 1205 TF/s (r01 microbench).
#define SA_SGB_PAIRS()                                                                                  \
    _Pragma("unroll") for (int i_ = 0; i_ < (FM + FN < FM * FN ? FM + FN : FM * FN); ++i_) {            \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                              \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                              \
    }                                                                                                   \
    if constexpr (FM * FN > FM + FN) __builtin_amdgcn_sched_group_barrier(0x008, FM * FN - FM - FN, 0); \
    if constexpr (FM + FN > FM * FN) __builtin_amdgcn_sched_group_barrier(0x100, FM + FN - FM * FN, 0);
#define SA_COMPUTE_FULL(CURP)                              \
    {                                                      \
        const unsigned char* cur_ = (CURP);                \
        u32x4 xfa[FM], wfa[FN], xfb[FM], wfb[FN];          \
        SA_FRAGS(xfa, wfa, 0);                             \
        SA_FRAGS(xfb, wfb, 1);                             \
        SA_MFMAS(xfa, wfa);                                \
        SA_FRAGS(xfa, wfa, 2);                             \
        SA_MFMAS(xfb, wfb);                                \
        SA_FRAGS(xfb, wfb, 3);                             \
        SA_MFMAS(xfa, wfa);                                \
        SA_MFMAS(xfb, wfb);                                \
        __builtin_amdgcn_sched_group_barrier(0x100, FM + FN, 0); \
        SA_SGB_PAIRS();                                    \
        SA_SGB_PAIRS();                                    \
        SA_SGB_PAIRS();                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, FM * FN, 0); \
    }
#else
#define SA_COMPUTE_FULL(CURP)                              \
    {                                                      \
        const unsigned char* cur_ = (CURP);                \
        u32x4 xfa[FM], wfa[FN], xfb[FM], wfb[FN];          \
        SA_FRAGS(xfa, wfa, 0);                             \
        SA_FRAGS(xfb, wfb, 1);                             \
        __builtin_amdgcn_sched_barrier(0);                 \
        SA_MFMAS(xfa, wfa);                                \
        __builtin_amdgcn_sched_barrier(0);                 \
        SA_FRAGS(xfa, wfa, 2);                             \
        __builtin_amdgcn_sched_barrier(0);                 \
        SA_MFMAS(xfb, wfb);                                \
        __builtin_amdgcn_sched_barrier(0);                 \
        SA_FRAGS(xfb, wfb, 3);                             \
        __builtin_amdgcn_sched_barrier(0);                 \
        SA_MFMAS(xfa, wfa);                                \
        __builtin_amdgcn_sched_barrier(0);                 \
        SA_MFMAS(xfb, wfb);                                \
    }
#endif
#define SA_COMPUTE(CURP)                                   \
    {                                                      \
        if constexpr (LEAN) SA_COMPUTE_LEAN(CURP)          \
        else SA_COMPUTE_FULL(CURP)                         \
    }
    if constexpr (GLDS > 0) {
        // Direct-to-LDS staging (global_load_lds_dwordx4): one instruction moves 64 lanes x 16 bytes = 8 consecutive
        // 128-byte tile rows from global memory into LDS at (wave-uniform M0 base) + lane * 16, with no staging VGPRs and
        // no ds_write traffic (the register pipeline below is LDS-write-bound on big tiles, r01 profile). The LDS image is
        // a plain linear copy, so the XOR swizzle is applied on the SOURCE side: lane l fills physical chunk (l & 7) of
        // row r = base + (l >> 3) and therefore fetches logical chunk (l & 7) ^ ((r >> 1) & 7) of that row.
        constexpr int NW = WM * WN, XI = BM / 8 / NW, WI = BN / 8 / NW;
        static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "glds: whole 8-row groups per wave");
        const unsigned char* xg[XI];
        const unsigned char* wg[WI];
        [[maybe_unused]] long cbase[XI];                     // CONV: element offset of image b of the lane's row
        [[maybe_unused]] int ciy[XI], cix[XI], cch[XI];      // CONV: top-left input coordinates of the row's window, the lane's K offset
        // CONV, K-tiles aligned with filter taps (round 5): everything about a (row, tap) pair that does not depend on the K-tile is
        // computed ONCE per workgroup -- the byte address of the window's top-left pixel (+ the lane's channel chunk) and one validity
        // bit per tap -- so a request costs a bit test, one 64-bit add and the select against the zero page (~5 VALU) instead of the
        // window arithmetic, five compares and a 64-bit multiply-add per request (~20: r04 counters, VALU issue 47 % beside MFMA 26 %).
        // (they take the place of cbase / ciy in that case: both sets live side by side cost four spills)
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const int row = (wave * XI + i) * 8 + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
            if constexpr (CONV) {
                const int m = min(m0 + row, p.M - 1), hw = p.cHo * p.cWo;
                const int b = m / hw, r = m - b * hw, oy = r / p.cWo;
                cbase[i] = (long)b * p.cH * p.cW * p.cCin;
                ciy[i] = oy * p.cStride - p.cPad;
                cix[i] = (r - oy * p.cWo) * p.cStride - p.cPad;
                cch[i] = c * Ty<TI>::V16;
                xg[i] = nullptr;
            } else {
                xg[i] = reinterpret_cast<const unsigned char*>(p.X + (long)min(m0 + row, p.M - 1) * p.ldx) + c * 16 + (long)kt_begin * 128;
            }
        }
        // K-tiles aligned with filter taps (Cin a multiple of the K-tile): the tap of a K-tile is a scalar
        [[maybe_unused]] const bool conv_uniform = CONV && (p.cCin % KE == 0) && p.cTaps <= 32 && p.conv_lean;
        if constexpr (CONV) {
            if (conv_uniform) {
#pragma unroll
                for (int i = 0; i < XI; ++i) {
                    unsigned m_ = 0;
                    for (int t_ = 0; t_ < p.cTaps; ++t_) {
                        const int iy_ = ciy[i] + t_ / p.cKW, ix_ = cix[i] + t_ % p.cKW;
                        m_ |= (unsigned)((iy_ >= 0) & (iy_ < p.cH) & (ix_ >= 0) & (ix_ < p.cW)) << t_;
                    }
                    cbase[i] = (cbase[i] + ((long)ciy[i] * p.cW + cix[i]) * p.cCin + cch[i]) * (long)sizeof(TI);   // now: BYTE offset of the window's corner + chunk
                    ciy[i] = (int)m_;                                                                               // now: validity bit per tap
                }
            }
        }
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            const int row = (wave * WI + i) * 8 + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
            wg[i] = reinterpret_cast<const unsigned char*>(p.W + (long)min(n0 + row, p.N - 1) * p.ldw) + c * 16 + (long)kt_begin * 128;
        }
        // Weight stream cache policy: default. Non-temporal weight loads (aux = 2, compile with -DSA_NT_W=1) were measured
        // on the decode-regime tiles and LOSE here (small-tile GEMMs 41.7 -> 46.9 ms per recognition step): with 64-row
        // tiles the four M-tiles of a weight slab run on one XCD and share it through that L2, which nt defeats.
#ifndef SA_NT_W
#define SA_NT_W 0
#endif
        // WAUX = 2 (template parameter): non-temporal weight stream for tiles whose W rows are read by exactly ONE workgroup (the
        // 256x320 lm_head tile spans all rows of the batch, so nothing shares its weight slab through L2).
        constexpr int W_AUX = WAUX ? WAUX : ((SA_NT_W && (SPLIT || BM * BN <= 128 * 64)) ? 2 : 0);
        typedef const __attribute__((address_space(1))) void* gptr_t;
        typedef __attribute__((address_space(3))) void* lptr_t;
#define SA_ISSUE(BUFOFF, KT)                                                                                            \
    {                                                                                                                   \
        const long koff_ = (long)(KT) * 128;                                                                            \
        if constexpr (CONV) {                                                                                           \
            const int kb_ = (KT) * KE;                                                                                  \
            int tap_u_ = 0, ci_u_ = 0;                                                                                  \
            if (conv_uniform) { tap_u_ = kb_ / p.cCin; ci_u_ = kb_ - tap_u_ * p.cCin; }                                 \
            if (conv_uniform) {                                                                                         \
                const int ky_ = tap_u_ / p.cKW, kx_ = tap_u_ - ky_ * p.cKW;                                             \
                const long toff_ = (((long)ky_ * p.cW + kx_) * p.cCin + ci_u_) * (long)sizeof(TI);   /* scalar */       \
                const unsigned bit_ = tap_u_ < 32 ? 1u << tap_u_ : 0u;                                                  \
                _Pragma("unroll") for (int i = 0; i < XI; ++i) {                                                        \
                    const unsigned char* src_ = ((unsigned)ciy[i] & bit_) ? reinterpret_cast<const unsigned char*>(p.conv_in) + cbase[i] + toff_ \
                                                                          : reinterpret_cast<const unsigned char*>(p.conv_zero); \
                    __builtin_amdgcn_global_load_lds((gptr_t)src_, (lptr_t)(smem + (BUFOFF) + (wave * XI + i) * 1024), 16, 0, 0); \
                }                                                                                                       \
            } else                                                                                                      \
            _Pragma("unroll") for (int i = 0; i < XI; ++i) {                                                            \
                int tap_, ci_;                                                                                          \
                if (conv_uniform) { tap_ = tap_u_; ci_ = ci_u_ + cch[i]; }                                              \
                else { const int k0_ = kb_ + cch[i]; tap_ = k0_ / p.cCin; ci_ = k0_ - tap_ * p.cCin; }                  \
                const int ky_ = tap_ / p.cKW, kx_ = tap_ - ky_ * p.cKW;                                                 \
                const int iy_ = ciy[i] + ky_, ix_ = cix[i] + kx_;                                                       \
                const bool ok_ = (tap_ < p.cTaps) & (iy_ >= 0) & (iy_ < p.cH) & (ix_ >= 0) & (ix_ < p.cW);              \
                const TI* src_ = ok_ ? p.conv_in + cbase[i] + ((long)iy_ * p.cW + ix_) * p.cCin + ci_ : p.conv_zero;    \
                __builtin_amdgcn_global_load_lds((gptr_t)src_, (lptr_t)(smem + (BUFOFF) + (wave * XI + i) * 1024), 16, 0, 0); \
            }                                                                                                           \
        } else {                                                                                                        \
            _Pragma("unroll") for (int i = 0; i < XI; ++i) __builtin_amdgcn_global_load_lds(                            \
                (gptr_t)(xg[i] + koff_), (lptr_t)(smem + (BUFOFF) + (wave * XI + i) * 1024), 16, 0, 0);                 \
        }                                                                                                               \
        _Pragma("unroll") for (int i = 0; i < WI; ++i) __builtin_amdgcn_global_load_lds(                                \
            (gptr_t)(wg[i] + koff_), (lptr_t)(smem + (BUFOFF) + XBYTES + (wave * WI + i) * 1024), 16, 0, W_AUX);        \
    }
        if constexpr (GLDS == 8) {
            // ---- 8-phase schedule (round 5) for the 256x256 bf16 tile: 8 waves = 4 (M) x 2 (N), 64 x 128 per wave, 2 waves per SIMD.
            // The 2-stage loop below issues a K-tile, multiplies a whole K-tile, then drains vmcnt(0) behind a full barrier: every wave
            // of the CU stalls once per K-tile for the youngest load (r04: 50 % matrix-pipe occupancy inside the loop). Here
            //   * a K-tile is split in FOUR half-tiles of 16 KiB -- X0 / X1 = the first / second 32 rows of every wave row (4 x 32 rows),
            //     W0 / W1 = the first / second 64 columns of every wave column (2 x 64 rows of W) -- and a PHASE multiplies one 32 x 64
            //     quadrant of every wave's output over the K-tile: 8 MFMAs (256 matrix-pipe cycles) from 4 X fragments and 8 W fragments;
            //   * the two waves of a SIMD (wave w and w + 4) run ONE barrier apart: while one multiplies a quadrant, the other reads the
            //     next quadrant's fragments from LDS and issues one half-tile of global_load_lds; raw s_barrier on both sides of the
            //     MFMA block, nothing drains;
            //   * LDS holds 8 half-tile slots (2 K-tiles, 128 KiB); the half-tile read in phase P + 5 is requested in phase P (4 half-tiles
            //     = 64 KiB in flight per CU, ~4 phases ~ 2000 cycles of latency cover) and the counted s_waitcnt vmcnt(8) of phase P only
            //     asks for the half-tile that phase P + 1 reads -- requested four phases ago.
            // Half-tile sequence h = 4 t + u, u: 0 = X0, 1 = W0, 2 = X1, 3 = W1 of K-tile t; read in phase h - 1, requested in phase h - 6:
            //   phase (t, 0): read W0(t)                  multiply (X0, W0)   request X1(t+1)
            //   phase (t, 1): read X1(t)                  multiply (X1, W0)   request W1(t+1)
            //   phase (t, 2): read W1(t)                  multiply (X1, W1)   request X0(t+2)
            //   phase (t, 3): read X0(t+1) (other regs)   multiply (X0, W1)   request W0(t+2)
            // Ordering (MI355X: nothing orders a ds_read behind a pending LDS-DMA but the issuer's vmcnt + a barrier the reader passed):
            //   RAW  the issuer waits in the load segment of phase R - 1 (before that phase's first barrier), readers read in phase R:
            //        at least one barrier between, also across the one-barrier stagger of the two wave groups.
            //   WAR  a slot is re-requested three phases after its last read (>= 5 barriers; the reads retire at the first MFMA).
            // K order, MFMA and accumulator assignment are the 2-stage loop's: results are bit-identical (tools/microbench/bigtile_ab.py 1 3).
            static_assert(BM == 256 && BN == 256 && WM == 4 && WN == 2 && !SPLIT && !CONV && sizeof(TI) == 2 && !LEAN, "8-phase schedule: the 256x256 bf16 tile");
            constexpr int HT = 16384;                                    // bytes of a half-tile slot; slot (b, u) at b * 4 * HT + u * HT
            const int wv = __builtin_amdgcn_readfirstlane(wave);
            // request side: instruction i of wave wv fills local rows (wv * 2 + i) * 8 + (lane >> 3) of a half-tile. Addresses are a
            // wave-uniform base (tile origin + K offset: scalar registers, advanced on the SALU) + a 32-bit lane offset (tile-local row
            // x row pitch + swizzled chunk), so a request is one global_load_lds with an SGPR base and costs no VALU.
            const unsigned char* xbase = reinterpret_cast<const unsigned char*>(p.X + (long)m0 * p.ldx);
            const unsigned char* wbase = reinterpret_cast<const unsigned char*>(p.W + (long)n0 * p.ldw);
            unsigned xq[2][2], wq[2][2];                                  // [half][i]
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int lr = (wv * 2 + i) * 8 + (lane >> 3), c = (lane & 7) ^ ((lr >> 1) & 7);
                    const int xr = (lr >> 5) * 64 + hh * 32 + (lr & 31), wr_ = (lr >> 6) * 128 + hh * 64 + (lr & 63);
                    xq[hh][i] = (unsigned)(min(xr, p.M - 1 - m0) * (int)(p.ldx * sizeof(TI)) + c * 16);
                    wq[hh][i] = (unsigned)(min(wr_, p.N - 1 - n0) * (int)(p.ldw * sizeof(TI)) + c * 16);
                }
#define SA8_REQ(BASE, OFFS, SLOT, KT)                                                                             \
    {                                                                                                             \
        const unsigned char* b_ = (BASE) + (long)(KT) * 128;                                                      \
        asm volatile("" : "+s"(b_));      /* keeps (uniform base) + zext(lane offset) visible to instruction selection: */ \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                        \
            unsigned o_ = OFFS[i_];       /* left alone, hipcc hoists base + offset as 64-bit VGPR pairs and adds K on the VALU */ \
            asm volatile("" : "+v"(o_));                                                                          \
            __builtin_amdgcn_global_load_lds((gptr_t)(b_ + o_), (lptr_t)(smem + (SLOT) * HT + (wv * 2 + i_) * 1024), 16, 0, 0); \
        }                                                                                                         \
    }
            // read side: lane offsets inside a half-tile (the XOR swizzle depends on the fragment row only)
            int fo[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) fo[kk] = frow * 128 + (((kk * 2 + fch) ^ ((frow >> 1) & 7)) << 4);
            // ds_read offsets are 16-bit immediates: one address set per K-tile buffer (64 KiB each), slot / fragment offsets folded
            int xl0[4], wl0[4], xl1[4], wl1[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                xl0[kk] = wm * 32 * 128 + fo[kk];              // + u * HT
                wl0[kk] = wn * 64 * 128 + fo[kk];              // + u * HT + j * 32 * 128
                xl1[kk] = xl0[kk] + 4 * HT;
                wl1[kk] = wl0[kk] + 4 * HT;
            }
            u32x4 xa[4], xb[4], wf[2][4];
#define SA8_RX(XR, SLOT) { _Pragma("unroll") for (int kk_ = 0; kk_ < 4; ++kk_) XR[kk_] = *reinterpret_cast<const u32x4*>(smem + ((SLOT) < 4 ? xl0[kk_] : xl1[kk_]) + ((SLOT) & 3) * HT); }
#define SA8_RW(SLOT)                                                                                              \
    {                                                                                                             \
        _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                                          \
            _Pragma("unroll") for (int kk_ = 0; kk_ < 4; ++kk_)                                                   \
                wf[j_][kk_] = *reinterpret_cast<const u32x4*>(smem + ((SLOT) < 4 ? wl0[kk_] : wl1[kk_]) + ((SLOT) & 3) * HT + j_ * 4096); \
    }
#define SA8_MMA(XR, JH, I)                                                                                        \
    {                                                                                                             \
        __builtin_amdgcn_s_setprio(1);                                                                            \
        _Pragma("unroll") for (int kk_ = 0; kk_ < 4; ++kk_)                                                       \
            _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) Mfma<TI>::run(acc[(JH) * 2 + j_][I], wf[j_][kk_], XR[kk_]); \
        __builtin_amdgcn_s_setprio(0);                                                                            \
    }
#define SA8_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
            // one phase: READ; REQ; counted wait | barrier | 8 MFMAs | barrier
#define SA8_PHASE(READ, REQ, VMN, XR, JH, I)          \
    {                                                 \
        READ;                                         \
        __builtin_amdgcn_sched_barrier(0);            \
        REQ;                                          \
        if constexpr ((VMN) >= 0) SA8_VM((VMN) < 0 ? 0 : (VMN)); \
        __builtin_amdgcn_sched_barrier(0);            \
        __builtin_amdgcn_s_barrier();                 \
        __builtin_amdgcn_sched_barrier(0);            \
        SA8_MMA(XR, JH, I);                           \
        __builtin_amdgcn_sched_barrier(0);            \
        __builtin_amdgcn_s_barrier();                 \
        __builtin_amdgcn_sched_barrier(0);            \
    }
            // slots: buffer 0 = 0..3, buffer 1 = 4..7; u: 0 X0, 1 W0, 2 X1, 3 W1
            SA8_REQ(xbase, xq[0], 0, 0); SA8_REQ(wbase, wq[0], 1, 0); SA8_REQ(xbase, xq[1], 2, 0); SA8_REQ(wbase, wq[1], 3, 0);
            SA8_REQ(xbase, xq[0], 4, 1); SA8_REQ(wbase, wq[0], 5, 1);
            SA8_VM(8);                                                   // X0(0), W0(0) of this wave have landed
            __builtin_amdgcn_s_barrier();
            if (wv >= 4) __builtin_amdgcn_s_barrier();                   // the second wave of every SIMD runs one barrier behind
            __builtin_amdgcn_sched_barrier(0);
            SA8_RX(xa, 0);
            for (int pi = 0; pi < pairs - 1; ++pi) {
                const int t = 2 * pi;
                SA8_PHASE(SA8_RW(1),     SA8_REQ(xbase, xq[1], 6, t + 1), 8, xa, 0, 0);
                SA8_PHASE(SA8_RX(xb, 2), SA8_REQ(wbase, wq[1], 7, t + 1), 8, xb, 0, 1);
                SA8_PHASE(SA8_RW(3),     SA8_REQ(xbase, xq[0], 0, t + 2), 8, xb, 1, 1);
                SA8_PHASE(SA8_RX(xb, 4), SA8_REQ(wbase, wq[0], 1, t + 2), 8, xa, 1, 0);
                SA8_PHASE(SA8_RW(5),     SA8_REQ(xbase, xq[1], 2, t + 2), 8, xb, 0, 0);
                SA8_PHASE(SA8_RX(xa, 6), SA8_REQ(wbase, wq[1], 3, t + 2), 8, xa, 0, 1);
                SA8_PHASE(SA8_RW(7),     SA8_REQ(xbase, xq[0], 4, t + 3), 8, xa, 1, 1);
                SA8_PHASE(SA8_RX(xa, 0), SA8_REQ(wbase, wq[0], 5, t + 3), 8, xb, 1, 0);
            }
            {   // last two K-tiles: nothing left to request after W1(nk - 1); the counted waits shrink with what is still in flight
                const int t = nk - 2;
                SA8_PHASE(SA8_RW(1),     SA8_REQ(xbase, xq[1], 6, t + 1), 8, xa, 0, 0);
                SA8_PHASE(SA8_RX(xb, 2), SA8_REQ(wbase, wq[1], 7, t + 1), 8, xb, 0, 1);
                SA8_PHASE(SA8_RW(3),     {},                       6, xb, 1, 1);
                SA8_PHASE(SA8_RX(xb, 4), {},                       4, xa, 1, 0);
                SA8_PHASE(SA8_RW(5),     {},                       2, xb, 0, 0);
                SA8_PHASE(SA8_RX(xa, 6), {},                       0, xa, 0, 1);
                SA8_PHASE(SA8_RW(7),     {},                      -1, xa, 1, 1);
                SA8_PHASE({},            {},                      -1, xb, 1, 0);
            }
            if (wv < 4) __builtin_amdgcn_s_barrier();                    // the leading half meets the trailing half's last barrier
#undef SA8_PHASE
#undef SA8_VM
#undef SA8_MMA
#undef SA8_RW
#undef SA8_RX
#undef SA8_REQ
        } else if constexpr (GLDS == 2) {
            // Two buffers, loop unrolled over both so every LDS offset is an immediate. Tile kt+1 streams into the other
            // buffer while tile kt is multiplied; the vmcnt(0) + barrier at the end of the iteration both publishes tile
            // kt+1 and retires every wave's reads of tile kt before its buffer is refilled. Big tiles run best this way:
            // 64 KiB of LDS keeps two workgroups per CU, which hides more latency than a deeper ring with one (r01 microbench:
            // 8k^3 985 TF/s with 2 stages, 830-860 with 3-4).
#define SA_LANDED()                                      \
    {                                                    \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
        __syncthreads();                                 \
    }
            // (The refill after the last tile is skipped: round 2 re-fetched the last K-tile there, clamped, "so the loads in flight
            // stay constant" -- a rule of the register-staged loop below that this loop, with its explicit vmcnt(0), never needed;
            // it cost every workgroup one more K-tile of L2 traffic and the wait for it, ~0.8 us of a 20-step decode gate|up launch.)
            SA_STAMP1(1);
            SA_ISSUE(0, 0);
            SA_LANDED();
            for (int pi = 0; pi < pairs; ++pi) {
                const int kt = 2 * pi;
                SA_ISSUE(BUF, kt + 1);                      // kt + 1 <= last inside the pair loop
                __builtin_amdgcn_sched_barrier(0);
                SA_COMPUTE(smem);
                __builtin_amdgcn_sched_barrier(0);
                SA_LANDED();
                if (kt + 2 <= last) SA_ISSUE(0, kt + 2);    // wave-uniform
                __builtin_amdgcn_sched_barrier(0);
                SA_COMPUTE(smem + BUF);
                __builtin_amdgcn_sched_barrier(0);
                SA_LANDED();
            }
            if constexpr (!LEAN) {                      // (the accumulator-heavy tiles are launched for even K-tile counts only)
                if (nk & 1) SA_COMPUTE(smem);
            }
#undef SA_LANDED
        } else {
        // GLDS-stage ring: tiles kt .. kt+GLDS-2 are in flight or resident while tile kt is multiplied. Per iteration:
        // wait until this wave's loads of tile kt have landed (vmcnt leaves the GLDS-2 younger tiles outstanding), one
        // raw s_barrier (all waves' parts of tile kt are in LDS, and every wave is done with tile kt-1), refill the
        // buffer tile kt-1 occupied with tile kt+GLDS-1, multiply tile kt. __syncthreads() is avoided inside the loop: its
        // fence makes hipcc drain vmcnt to 0, which would collapse the prefetch distance to one tile.
        constexpr int LPT = XI + WI;                       // loads per tile per wave
        static_assert(GLDS <= 4 && (GLDS - 2) * LPT <= 63, "ring depth / vmcnt range");
        // Only tiles that exist are fetched. The wait in front of tile kt leaves min(GLDS - 2, last - kt) younger tiles in flight, so
        // the count is an immediate chosen by a (wave-uniform) branch on the tiles that remain. Round 2 kept the count constant by
        // re-fetching the last tile GLDS - 1 times at the tail and waiting for all of it before the epilogue: 3 extra K-tiles of
        // traffic and their latency per workgroup -- on the o-projection's 6-7 K-tiles per slice, +45 % bytes.
#pragma unroll
        for (int st = 0; st < GLDS - 1; ++st)
            if (st <= last) SA_ISSUE(st * BUF, st);
        int rd = 0, wr = (GLDS - 1) * BUF;                 // byte offsets of the buffer to multiply / to refill
        for (int kt = 0; kt < nk; ++kt) {
            const int rem = last - kt;                     // younger tiles already requested: min(rem, GLDS - 2)
            if (rem >= GLDS - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((GLDS - 2) * LPT) : "memory");
            else if (rem == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (kt + GLDS - 1 <= last) SA_ISSUE(wr, kt + GLDS - 1);
            __builtin_amdgcn_sched_barrier(0);
            SA_COMPUTE(smem + rd);
            __builtin_amdgcn_sched_barrier(0);
            rd = rd + BUF == GLDS * BUF ? 0 : rd + BUF;
            wr = wr + BUF == GLDS * BUF ? 0 : wr + BUF;
        }
        }
#undef SA_ISSUE
    } else {
    // global source pointers for this thread's staging chunks (rows clamped into range)
    const unsigned char* xsrc[XCH];
    const unsigned char* wsrc[WCH];
    int xdst[XCH], wdst[WCH];
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
        const int id = (tid + i * NT) % XMOD, row = id >> 3, c = id & 7;
        const int gr = min(m0 + row, p.M - 1);
        xsrc[i] = reinterpret_cast<const unsigned char*>(p.X + (long)gr * p.ldx) + c * 16 + (long)kt_begin * 128;
        xdst[i] = row * 128 + ((c ^ ((row >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int i = 0; i < WCH; ++i) {
        const int id = (tid + i * NT) % WMOD, row = id >> 3, c = id & 7;
        const int gr = min(n0 + row, p.N - 1);
        wsrc[i] = reinterpret_cast<const unsigned char*>(p.W + (long)gr * p.ldw) + c * 16 + (long)kt_begin * 128;
        wdst[i] = XBYTES + row * 128 + ((c ^ ((row >> 1) & 7)) << 4);
    }

    // Two statically named register sets (runtime-indexed arrays would be demoted to scratch memory).
    u32x4 xr0[XCH], wr0[WCH], xr1[XCH], wr1[WCH];
#define SA_FETCH(XR, WR, KT)                                                                   \
    {                                                                                          \
        const long koff_ = (long)(KT) * 128;                                                   \
        _Pragma("unroll") for (int i = 0; i < XCH; ++i)                                        \
            XR[i] = *reinterpret_cast<const u32x4*>(xsrc[i] + koff_);              \
        _Pragma("unroll") for (int i = 0; i < WCH; ++i)                                        \
            WR[i] = *reinterpret_cast<const u32x4*>(wsrc[i] + koff_);              \
    }
#define SA_STASH(XR, WR, BUFP)                                                                 \
    {                                                                                          \
        unsigned char* b_ = (BUFP);                                                            \
        _Pragma("unroll") for (int i = 0; i < XCH; ++i)                                        \
            *reinterpret_cast<u32x4*>(b_ + xdst[i]) = XR[i];                       \
        _Pragma("unroll") for (int i = 0; i < WCH; ++i)                                        \
            *reinterpret_cast<u32x4*>(b_ + wdst[i]) = WR[i];                       \
    }
    // iteration kt: the set that held tile kt is free (tile kt already sits in LDS) -> refill it with tile kt+2;
    // compute tile kt; publish tile kt+1 (fetched one iteration ago into the other set) to the other LDS buffer.
    // Fetches and publishes are UNCONDITIONAL (tile index clamped to the last tile, redundant at the tail): with a
    // conditional fetch hipcc cannot count the loads in flight at the merge point and falls back to vmcnt(0) before
    // every fetch, which collapses the prefetch distance to one tile (seen in the r01 ISA: ~1 us per K-iteration).
    SA_FETCH(xr0, wr0, 0);
    SA_FETCH(xr1, wr1, min(1, last));
    SA_STASH(xr0, wr0, smem);
    __syncthreads();
    for (int pi = 0; pi < pairs; ++pi) {
        const int kt = 2 * pi;
        // sched_barrier(0) pins fetch -> compute -> publish: left alone, hipcc hoists the publish (and its vmcnt wait for
        // loads issued a few instructions earlier) above the MFMAs, exposing the full global latency every iteration.
        SA_FETCH(xr0, wr0, min(kt + 2, last));
        __builtin_amdgcn_sched_barrier(0);
        SA_COMPUTE(smem);
        __builtin_amdgcn_sched_barrier(0);
        SA_STASH(xr1, wr1, smem + BUF);
        __syncthreads();
        SA_FETCH(xr1, wr1, min(kt + 3, last));
        __builtin_amdgcn_sched_barrier(0);
        SA_COMPUTE(smem + BUF);
        __builtin_amdgcn_sched_barrier(0);
        SA_STASH(xr0, wr0, smem);
        __syncthreads();
    }
    if (nk & 1) SA_COMPUTE(smem);          // odd tail: tile nk-1 was published to buffer 0 by the last pair
#undef SA_FETCH
#undef SA_STASH
    }

    SA_STAMP1(2);
    // Epilogue through LDS. 32x32 result D[n][m]: a lane owns row m = .. + (lane & 31) and, per register group g, four
    // consecutive columns -- written straight to HBM that is 64 scattered 8-byte pieces per store instruction (the
    // N = 1280 residual GEMMs ran at 400 TF/s on it). Instead the tile is staged in the (now idle) staging LDS with bias /
    // activation applied, then stored as whole 16-byte chunks of contiguous rows; the residual is added on the way out.
    //
    // Every global operand of the epilogue (bias, rotary table, residual) is fetched in BATCHES of unconditional loads with
    // clamped addresses, one wait per batch. Round 2 loaded them where they were used, under `if (p.bias)` / `if (n < rope_cols)` /
    // loop-`continue` conditions: hipcc then branches around every load and waits vmcnt(0) behind each -- 32 dependent L2 round
    // trips for the bias of a 256x256 tile and 16 for its residual (r03 ISA), ~15-20 us of a ~50 us tile at K = 1280.
#ifndef SA_ABL
#define SA_ABL 0      // ablation builds of the 8-phase tile (tools/microbench): 1 = epilogue without its global stores, 2 = no epilogue at all
#endif
    if constexpr (SA_ABL == 2 && GLDS == 8) {
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int i = 0; i < FM; ++i) asm volatile("" ::"v"(acc[j][i]));
        return;
    }
    __syncthreads();
    // Greedy-head partials straight from the accumulators, for tiles whose fp32 image does not fit LDS (the 256x320 lm_head tile):
    // a lane holds, for each of its FM rows, FN x 16 of the wave's columns in ascending order (the other FN x 16 sit in lane ^ 32),
    // so the row's (max, first argmax, sum exp) over the wave's columns is a serial pass + one exchange; the WN waves of a row meet
    // in 16 bytes of LDS each.
    constexpr bool ARGMAX_DIRECT = (EPI == EPI_ARGMAX) && ((size_t)BM * BN * 4 > 160 * 1024);
    if constexpr (ARGMAX_DIRECT) {
        static_assert(!SPLIT && std::is_same<TO, float>::value && NT >= BM, "direct argmax epilogue");
        const bool with_bias = p.bias != nullptr;                             // wave-uniform
        float4* rec = reinterpret_cast<float4*>(smem);                        // [BM][WN] per-wave records of a row
        float* bias_s = reinterpret_cast<float*>(smem + (size_t)BM * WN * 16);   // [BN] the tile's bias in fp32 (LDS: 20 float4 per lane would not fit beside the accumulators)
        if (tid < BN / 4) {
            float b[4] = {0.f, 0.f, 0.f, 0.f};
            if (with_bias) load4(p.bias + min(n0 + tid * 4, p.N - 4), b);
            *reinterpret_cast<float4*>(bias_s + tid * 4) = make_float4(b[0], b[1], b[2], b[3]);
        }
        __syncthreads();
        // columns of this lane: base + off with off = j * 32 + g * 8 + r a compile-time constant; off < lim are inside N
        const int base = n0 + wn * WTN + (lane >> 5) * 4, lim = p.N - base;
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 b4 = *reinterpret_cast<const float4*>(bias_s + wn * WTN + j * 32 + g * 8 + (lane >> 5) * 4);
                const float b[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r)          // masked once, here: every later pass sees -inf past N
                        acc[j][i][4 * g + r] = (j * 32 + g * 8 + r < lim) ? acc[j][i][4 * g + r] + b[r] : -INFINITY;
            }
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            __builtin_amdgcn_sched_barrier(0);                                    // one row block at a time (register pressure)
            float best = -INFINITY;
            int bo = 0;
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = acc[j][i][4 * g + r];
                        if (v > best) { best = v; bo = j * 32 + g * 8 + r; }       // strict >: first maximum wins (offsets ascend)
                    }
            int bi = (best == -INFINITY) ? 0x7fffffff : base + bo;
            {
                const float ob = __shfl_xor(best, 32, 64);
                const int oi = __shfl_xor(bi, 32, 64);
                if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
            }
            const float bsafe = (best == -INFINITY) ? 0.f : best;                 // exp(-inf - 0) = 0 for a row block wholly past N
            float se = 0.f;
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if constexpr (std::is_same<TI, float>::value) se += expf(acc[j][i][4 * g + r] - bsafe);      // reference mode
                        else se += __expf(acc[j][i][4 * g + r] - bsafe);
                    }
            se += __shfl_xor(se, 32, 64);
            if (lane < 32) rec[(wm * WTM + i * 32 + lane) * WN + wn] = make_float4(best, __int_as_float(bi), se, 0.f);
        }
        __syncthreads();
        if (tid < BM && m0 + tid < p.M) {
            float best = -INFINITY;
            int bi = 0x7fffffff;
#pragma unroll
            for (int w = 0; w < WN; ++w) {
                const float4 r4 = rec[tid * WN + w];
                const int ri = __float_as_int(r4.y);
                if (r4.x > best || (r4.x == best && ri < bi)) { best = r4.x; bi = ri; }
            }
            float se = 0.f;
#pragma unroll
            for (int w = 0; w < WN; ++w) {
                const float4 r4 = rec[tid * WN + w];
                se += r4.z * expf(r4.x - best);                                      // a wave past N holds (-inf, -, 0): contributes 0
            }
            p.amax[(long)(m0 + tid) * tiles_n + tile_n] = make_float4(best, __int_as_float(bi), se, 0.f);
        }
        return;
    }
    constexpr bool GLU = (EPI == EPI_SWIGLU || EPI == EPI_GEGLU) && !SPLIT;     // gated epilogues halve the output width
    constexpr int OW = GLU ? BN / 2 : BN;                                    // output columns of this tile
    using TS = typename std::conditional<SPLIT, float, TO>::type;            // staged / stored element type
    constexpr int ROWB = OW * (int)sizeof(TS), CPR = ROWB / 16;              // bytes and 16-byte chunks per tile row
    constexpr int XM = CPR >= 8 ? 7 : CPR - 1;                               // chunk XOR mask (conflict-free b128 writes)
    static_assert(ARGMAX_DIRECT || (BM * ROWB <= 160 * 1024 && CPR >= 1), "output tile must fit LDS (launcher sizes it)");
    const bool has_bias = !SPLIT && p.bias != nullptr;                       // wave-uniform
    // the lane's 4 bias columns of (j, g), the same for every i; kept as loaded (packed bf16: 2 registers) until they are used
    using BiasRaw = typename std::conditional<std::is_same<TI, float>::value, float4, uint2>::type;
    [[maybe_unused]] BiasRaw bias_raw[FN][4];
    if constexpr (!SPLIT) {
        if (has_bias) {
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    bias_raw[j][g] = *reinterpret_cast<const BiasRaw*>(p.bias + min(n0 + wn * WTN + j * 32 + g * 8 + (lane >> 5) * 4, p.N - 4));
        }
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int row = wm * WTM + i * 32 + (lane & 31);
        constexpr int JG = FN >= 2 ? 2 : 1;                                  // rotary entries are fetched for JG column blocks at a time
        [[maybe_unused]] float4 rope_v[JG][4];                               // (8 x 16 bytes in flight; all FN at once spilled at 256x256)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            if constexpr (EPI == EPI_ROPE && !SPLIT) {
                if (j % JG == 0) {
                    // (cos j, sin j, cos j+1, sin j+1) of this lane's row; columns past rope_cols read a clamped (unused) entry
                    const int m = min(m0 + row, p.M - 1), half = p.rope_D >> 1;
#pragma unroll
                    for (int jj = 0; jj < JG; ++jj)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int n = min(min(n0 + wn * WTN + (j + jj) * 32 + g * 8 + (lane >> 5) * 4, p.N - 4), p.rope_cols - 4);
                            rope_v[jj][g] = *reinterpret_cast<const float4*>(p.rope + (long)m * half + ((n % p.rope_D) >> 1));
                        }
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ncol = wn * WTN + j * 32 + g * 8 + (lane >> 5) * 4;   // tile-local column of v[0]
                float v[4] = {acc[j][i][4 * g], acc[j][i][4 * g + 1], acc[j][i][4 * g + 2], acc[j][i][4 * g + 3]};
                if constexpr (!SPLIT) {
                    if (has_bias) {
                        float b[4];
                        load4(reinterpret_cast<const TI*>(&bias_raw[j][g]), b);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += b[r];
                    }
                    if constexpr (EPI == EPI_GELU) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = gelu_epi<TI>(v[r]);
                    } else if constexpr (EPI == EPI_HARDSWISH) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = hardswish_f(v[r]);
                    } else if constexpr (EPI == EPI_RELU) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                    }
                }
                if constexpr (EPI == EPI_ROPE && !SPLIT) {
                    const int n = min(n0 + ncol, p.N - 4);
                    if (n < p.rope_cols) {
                        // reference order (encoder/__init__.py:188-199): the projection output is rounded to the storage
                        // dtype, rotated in fp32, rounded once more (the store below)
                        const float4 cs = rope_v[j % JG][g];                                         // (cos j, sin j, cos j+1, sin j+1)
                        const float x0 = Ty<TO>::rnd(v[0]), x1 = Ty<TO>::rnd(v[1]), x2 = Ty<TO>::rnd(v[2]), x3 = Ty<TO>::rnd(v[3]);
                        // products and sums pinned (one rounded product, one fused multiply-add): left to the compiler, the
                        // 128x128 and 256x256 instantiations contracted these differently and the SAME patch got encoder features
                        // one bf16 ulp apart depending on how many other lines were in the batch (r02 batch-invariance test)
                        v[0] = __fmaf_rn(x0, cs.x, -__fmul_rn(x1, cs.y)); v[1] = __fmaf_rn(x1, cs.x, __fmul_rn(x0, cs.y));
                        v[2] = __fmaf_rn(x2, cs.z, -__fmul_rn(x3, cs.w)); v[3] = __fmaf_rn(x3, cs.z, __fmul_rn(x2, cs.w));
                    }
                }
                if constexpr (GLU) {
                    // weight rows interleaved (gate_j, up_j): columns ncol..+3 = g0,u0,g1,u1 -> outputs ncol/2, ncol/2+1
                    const int boff = (ncol >> 1) * (int)sizeof(TS);
                    TS* dst = reinterpret_cast<TS*>(smem + row * ROWB + ((((boff >> 4) ^ (row & XM)) << 4) | (boff & 15)));
                    if constexpr (EPI == EPI_GEGLU) {
                        // the reference rounds gate and up to the storage dtype, applies gelu there, then multiplies (two Linear outputs)
                        const float g0 = Ty<TO>::rnd(gelu_tanh_f(Ty<TO>::rnd(v[0]))), g1 = Ty<TO>::rnd(gelu_tanh_f(Ty<TO>::rnd(v[2])));
                        store2(dst, g0 * Ty<TO>::rnd(v[1]), g1 * Ty<TO>::rnd(v[3]));
                    } else {
                        store2(dst, silu_epi<TI>(v[0]) * v[1], silu_epi<TI>(v[2]) * v[3]);
                    }
                } else {
                    const int boff = ncol * (int)sizeof(TS);
                    TS* dst = reinterpret_cast<TS*>(smem + row * ROWB + ((((boff >> 4) ^ (row & XM)) << 4) | (boff & 15)));
                    store4(dst, v[0], v[1], v[2], v[3]);
                }
            }
        }
    }
    __syncthreads();
    if constexpr (EPI == EPI_ARGMAX) {
        // Greedy-head partials straight from the staged tile: the [M, N] fp32 logits never travel to HBM (84 MB written and
        // read twice per decode step at V = 81920, M = 256 -- more than the lm_head weights themselves).
        static_assert(!SPLIT && std::is_same<TO, float>::value, "argmax epilogue works on fp32 tiles");
        constexpr int TPR = NT / BM;                  // threads per tile row (adjacent lanes)
        static_assert(NT % BM == 0 && (TPR & (TPR - 1)) == 0 && TPR <= 8 && CPR % TPR == 0, "argmax epilogue split");
        constexpr int SEG = CPR / TPR;                // 16-byte chunks per thread
        const int row = tid / TPR, part = tid % TPR;
        const unsigned char* rowp = smem + row * ROWB;
        float best = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll 4
        for (int cc = 0; cc < SEG; ++cc) {
            const int c = part * SEG + cc, n = n0 + c * 4;
            const f32x4 v = *reinterpret_cast<const f32x4*>(rowp + ((c ^ (row & XM)) << 4));
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (n + i < p.N && v[i] > best) { best = v[i]; bi = n + i; }     // strict >: first maximum wins
        }
#pragma unroll
        for (int o = 1; o < TPR; o <<= 1) {
            const float ob = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        float se = 0.f;
#pragma unroll 4
        for (int cc = 0; cc < SEG; ++cc) {
            const int c = part * SEG + cc, n = n0 + c * 4;
            const f32x4 v = *reinterpret_cast<const f32x4*>(rowp + ((c ^ (row & XM)) << 4));
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (n + i < p.N) se += expf(v[i] - best);
        }
#pragma unroll
        for (int o = 1; o < TPR; o <<= 1) se += __shfl_xor(se, o, 64);
        if (part == 0 && m0 + row < p.M) p.amax[(long)(m0 + row) * tiles_n + tile_n] = make_float4(best, __int_as_float(bi), se, 0.f);
        return;
    }
    constexpr int EPC = 16 / (int)sizeof(TS);                                // elements per 16-byte chunk
    const int n_out = GLU ? p.N / 2 : p.N;
    const int n0_out = GLU ? n0 / 2 : n0;
    constexpr int ITERS = (BM * CPR + NT - 1) / NT;                          // 16-byte chunks per thread
    [[maybe_unused]] u32x4 res[ITERS];
    if constexpr (EPI == EPI_RESIDUAL && !SPLIT) {                           // the whole tile's residual in one batch of loads
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int id = min(tid + it * NT, BM * CPR - 1), row = id / CPR, c = id % CPR;
            const int m = min(m0 + row, p.M - 1), n = min(n0_out + c * EPC, n_out - EPC);
            res[it] = *reinterpret_cast<const u32x4*>(p.R + (long)m * p.ldr + n);
        }
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int id = tid + it * NT, row = id / CPR, c = id % CPR;
        const int m = m0 + row, n = n0_out + c * EPC;
        if (id >= BM * CPR || m >= p.M || n >= n_out) continue;              // stores only: nothing below waits on memory
        u32x4 raw = *reinterpret_cast<const u32x4*>(smem + row * ROWB + ((c ^ (row & XM)) << 4));
        if constexpr (SA_ABL == 1 && GLDS == 8) { asm volatile("" ::"v"(raw)); continue; }
        if constexpr (SPLIT) {
            *reinterpret_cast<u32x4*>(p.part + ((long)ks * p.M + m) * p.N + n) = raw;
        } else {
            if constexpr (EPI == EPI_RESIDUAL) {
                float a[EPC], r[EPC];
                const uint4 raw4 = make_uint4(raw[0], raw[1], raw[2], raw[3]);
                unpack16(raw4, a, (TO*)nullptr);
                unpack16(make_uint4(res[it][0], res[it][1], res[it][2], res[it][3]), r, (TO*)nullptr);
                TO* dst = p.C + (long)m * p.ldc + n;
#pragma unroll
                for (int e = 0; e < EPC; e += 4) store4(dst + e, a[e] + r[e], a[e + 1] + r[e + 1], a[e + 2] + r[e + 2], a[e + 3] + r[e + 3]);
            } else {
                *reinterpret_cast<u32x4*>(p.C + (long)m * p.ldc + n) = raw;
            }
        }
    }
    SA_STAMP1(3);
#undef SA_STAMP1
}

// ------------------------------------------------------------------------------------------------------------------------
// Persistent 8-phase tile loop (round 5; replaces round 4's persistent 2-stage loop, which was bit-identical and not faster).
// A 256x256 tile of K = 1280 spends ~28 us in the 8-phase K loop above and ~9 us outside it (tools/microbench/p8_abl.py,
// profiles/r05_c_*): workgroup launch, the first half-tiles' round trip, an epilogue that needs the operand LDS back (two
// __syncthreads, nothing of the next tile in flight) and the drain of its stores before the CU takes the next workgroup. Here ONE
// workgroup per CU walks the XCD-swizzled tile order and never leaves the K-loop's schedule:
//   * the half-tile ring runs on ACROSS tiles: the last six phases of a tile request the first six half-tiles of the next tile
//     (same slots, same distances, same counted waits);
//   * the epilogue is wave-private: each wave stages its 64 x 128 results block by block (32 rows x 64 columns = 4 KiB) through its
//     own 4 KiB of the 32 KiB of LDS the ring leaves free, so it needs no workgroup barrier and no operand slot. The two wave groups
//     are re-ALIGNED behind a tile's last phase and staggered again at the top of the next tile: carried through the epilogue, the
//     stagger made each group wait at a barrier for the other group's whole epilogue (tools/microbench/p8_timing.hip);
//   * vmcnt counts stores as well as loads: every load in flight is retired (one vmcnt(0)) before a wave's first store, the first four
//     phases of the next tile run without a counted wait (everything they need was requested before the stores and has been retired),
//     and the counted waits resume at phase 4 -- about 1 us after the last store was issued;
//   * the accumulators restart from the MFMA's zero C operand in the first K-tile (no 128-register clear);
//   * the bias slice of a wave (128 columns) is ONE register per lane, requested in the tile's last phase, and reaches the lanes that
//     need it through ds_bpermute (no LDS bytes).
// Everything that is a function of the lane id alone (request offsets, fragment offsets, epilogue addresses) is recomputed where it is
// needed from a fresh v_mbcnt: kept live across the tile loop beside 128 accumulators it is spilled, and a scratch reload waits vmcnt(0).
// K order, MFMA, accumulator assignment and every epilogue formula are the one-tile kernel's: bit-identical (tests/test_gpu_round4.py).
// Requires an even K-tile count >= 4 and a 2-byte output type.
template <typename TI, typename TO, int EPI, int CONV = 0>
__global__ __launch_bounds__(512) void gemm_nt_p8p_kernel(GemmArgs<TI, TO> p) {
    constexpr int BM = 256, BN = 256, GRP = 32, KE = Ty<TI>::KE, HT = 16384;
    static_assert(sizeof(TI) == 2 && sizeof(TO) == 2 && EPI != EPI_ARGMAX, "persistent 8-phase loop: bf16 operands, 2-byte outputs");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wv >> 1, wn = wv & 1;
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN, total = tiles_m * tiles_n;
    const int padded = ((total + 8 * GRP - 1) / (8 * GRP)) * (8 * GRP);
    // CONV (implicit-GEMM convolution, K-tiles aligned with filter taps: Cin % 64 == 0): an odd K-tile count gets one VIRTUAL K-tile of
    // zeros behind the last (its X requests fall past the last tap and take the zero page like any padded tap, its W requests take the
    // zero page too): + 0 * 0 leaves every accumulator bit unchanged and the phase structure stays two K-tiles per round.
    const int nk_real = p.K / KE, nk = CONV ? nk_real + (nk_real & 1) : nk_real, pairs = nk >> 1;

    // virtual block id -> output tile (the one-tile kernel's XCD-aware super-tile order)
    auto map_tile = [&](int vb, int& tile_m, int& tile_n) -> bool {
        const int x = vb & 7, j = vb >> 3;
        const int idx = ((j / GRP) * 8 + x) * GRP + (j % GRP);
        if (idx >= total) return false;
        const int per_row = p.swz_m * tiles_n;
        const int sr = idx / per_row, rem = idx - sr * per_row;
        const int h = min(p.swz_m, tiles_m - sr * p.swz_m);
        const int sc = rem / (h * p.swz_n), rem2 = rem - sc * h * p.swz_n;
        const int w = min(p.swz_n, tiles_n - sc * p.swz_n);
        tile_m = sr * p.swz_m + rem2 / w;
        tile_n = sc * p.swz_n + rem2 % w;
        return true;
    };
    int vb_next = (int)blockIdx.x;
    auto next_tile = [&](int& tile_m, int& tile_n) -> bool {      // wave-uniform
        while (vb_next < padded) {
            const int vb = vb_next;
            vb_next += (int)gridDim.x;
            if (map_tile(vb, tile_m, tile_n)) return true;
        }
        return false;
    };

    // The lane id, recomputed where it is needed (v_mbcnt on an opaque zero: hipcc can neither hoist nor merge it). Left to a register
    // that lives across the tile loop it is spilled beside the 128 accumulators, and a scratch reload waits vmcnt(0): it drains the
    // request stream -- or, at the top of a tile, waits for the previous tile's stores.
    auto fresh_lane = []() -> int {
        unsigned z = 0;
        asm volatile("" : "+v"(z));
        return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z));
    };
    // ---- request side (see the 8-phase loop of gemm_nt_kernel): scalar tile base + 32-bit lane offset
    const unsigned char* xbase;
    const unsigned char* wbase;
    unsigned xq[2][2], wq[2][2];
    auto set_req = [&](int tile_m, int tile_n) {
        // (the lane id is made opaque per call: left visible, hipcc hoists the lane-dependent halves of these eight offsets out of the tile
        // loop, keeps them live beside 128 accumulators and spills -- and every scratch reload waits vmcnt(0), i.e. drains the request stream)
        const int lane = fresh_lane();
        const int m0 = tile_m * BM, n0 = tile_n * BN;
        xbase = reinterpret_cast<const unsigned char*>(p.X + (long)m0 * p.ldx);
        wbase = reinterpret_cast<const unsigned char*>(p.W + (long)n0 * p.ldw);
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int lr = (wv * 2 + i) * 8 + (lane >> 3), c = (lane & 7) ^ ((lr >> 1) & 7);
                const int xr = (lr >> 5) * 64 + hh * 32 + (lr & 31), wr_ = (lr >> 6) * 128 + hh * 64 + (lr & 63);
                xq[hh][i] = (unsigned)(min(xr, p.M - 1 - m0) * (int)(p.ldx * sizeof(TI)) + c * 16);
                wq[hh][i] = (unsigned)(min(wr_, p.N - 1 - n0) * (int)(p.ldw * sizeof(TI)) + c * 16);
            }
    };
    // CONV request state per (half, instruction): signed byte offset of the window's top-left pixel (+ the lane's channel chunk) from conv_in,
    // and one validity bit per filter tap (the launcher guarantees the tensor spans < 2 GiB and <= 32 taps)
    [[maybe_unused]] int cb[2][2];
    [[maybe_unused]] unsigned cm[2][2];
    [[maybe_unused]] auto set_conv = [&](int tile_m, int hh, int i) {
        // cm: bit ky = filter row ky lies inside the image for this output pixel, bit 16 + kx = filter column kx does (closed form, no tap
        // loop; no division instruction sequences: this runs once per tile inside a phase's load segment, and the first version -- `/` and
        // `%` per tap and per row -- cost thousands of cycles per tile, tools/microbench/p8_timing.hip)
        const int lane = fresh_lane();
        const int hw = p.cHo * p.cWo, KH = p.cTaps / p.cKW;
        {
            const int lr = (wv * 2 + i) * 8 + (lane >> 3), c = (lane & 7) ^ ((lr >> 1) & 7);
            const int xr = (lr >> 5) * 64 + hh * 32 + (lr & 31);
            const unsigned m = (unsigned)min(tile_m * BM + xr, p.M - 1);
            const unsigned bi = fast_div(m, p.fd_hw), r = m - bi * (unsigned)hw, oy = fast_div(r, p.fd_wo), ox = r - oy * (unsigned)p.cWo;
            const int iy0 = (int)oy * p.cStride - p.cPad, ix0 = (int)ox * p.cStride - p.cPad;
            cb[hh][i] = ((((int)bi * p.cH + iy0) * p.cW + ix0) * p.cCin) * (int)sizeof(TI) + c * 16;
            const int rlo = max(0, -iy0), rhi = max(rlo, min(KH, p.cH - iy0)), clo = max(0, -ix0), chi = max(clo, min(p.cKW, p.cW - ix0));
            cm[hh][i] = ((1u << rhi) - (1u << rlo)) | (((1u << chi) - (1u << clo)) << 16) | (CONV == 2 ? (unsigned)(c >> 2) << 31 : 0u);
        }
    };
    // X request of half HH for K-tile KT: plain rows, or the convolution gather (tap of the K-tile: scalar; ~8 VALU per request)
#define SP_REQX(HH, SLOT, KT)                                                                                     \
    {                                                                                                             \
        if constexpr (CONV == 1) {                                                                                \
            const unsigned tap_ = ((unsigned)(KT) * p.cv_m1) >> 16, ky_ = (tap_ * p.cv_m2) >> 16, kx_ = tap_ - ky_ * (unsigned)p.cKW;   /* scalar */ \
            const int toff_ = (KT) * 128 + (int)ky_ * p.cv_rowskip;                                               \
            const unsigned bit_ = (1u << ky_) | (0x10000u << kx_);    /* the virtual K-tile: ky = KH, no such row bit */ \
            _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                    \
                int c_ = cb[HH][i_];                                                                              \
                asm volatile("" : "+v"(c_));          /* the 64-bit address is formed HERE, not hoisted as a register pair per request */ \
                const unsigned char* src_ = ((cm[HH][i_] & bit_) == bit_) ? reinterpret_cast<const unsigned char*>(p.conv_in) + (long)(c_ + toff_) \
                                                                : reinterpret_cast<const unsigned char*>(p.conv_zero); \
                __builtin_amdgcn_global_load_lds((gptr_t)src_, (lptr_t)(smem + (SLOT) * HT + (wv * 2 + i_) * 1024), 16, 0, 0); \
            }                                                                                                     \
        } else if constexpr (CONV == 2) {                                                                         \
            /* Cin = 32: a K-tile row is taps 2 kt (chunks 0..3) and 2 kt + 1 (chunks 4..7). Both taps' scalars, selected per lane by */ \
            /* the chunk's half (bit 31 of cm); the byte offset tap * 64 = kt * 128 + half * 64 is already in cb (c * 16). Taps past */ \
            /* the filter (the odd ninth tap's partner, the virtual K-tile) have ky >= KH: no such row bit, zero page.               */ \
            const unsigned ta_ = 2u * (unsigned)(KT), kya_ = (ta_ * p.cv_m2) >> 16, kxa_ = ta_ - kya_ * (unsigned)p.cKW;          \
            const unsigned tb_ = ta_ + 1u, kyb_ = (tb_ * p.cv_m2) >> 16, kxb_ = tb_ - kyb_ * (unsigned)p.cKW;                      \
            const int toa_ = (KT) * 128 + (int)kya_ * p.cv_rowskip, tob_ = (KT) * 128 + (int)kyb_ * p.cv_rowskip;                  \
            const unsigned bita_ = (1u << kya_) | (0x10000u << kxa_), bitb_ = (1u << kyb_) | (0x10000u << kxb_);                   \
            _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                    \
                int c_ = cb[HH][i_];                                                                              \
                asm volatile("" : "+v"(c_));                                                                      \
                const bool hi_ = (int)cm[HH][i_] < 0;                                                             \
                const unsigned bit_ = hi_ ? bitb_ : bita_;                                                        \
                const int toff_ = hi_ ? tob_ : toa_;                                                              \
                const unsigned char* src_ = ((cm[HH][i_] & bit_) == bit_) ? reinterpret_cast<const unsigned char*>(p.conv_in) + (long)(c_ + toff_) \
                                                                : reinterpret_cast<const unsigned char*>(p.conv_zero); \
                __builtin_amdgcn_global_load_lds((gptr_t)src_, (lptr_t)(smem + (SLOT) * HT + (wv * 2 + i_) * 1024), 16, 0, 0); \
            }                                                                                                     \
        } else {                                                                                                  \
            SP_REQ(xbase, xq[HH], SLOT, KT);                                                                      \
        }                                                                                                         \
    }
    // W request: the virtual K-tile of an odd CONV K-tile count reads the zero page
#define SP_REQW(HH, SLOT, KT)                                                                                     \
    {                                                                                                             \
        if constexpr (CONV) {                                                                                     \
            const bool real_ = (KT) < nk_real;                                                                    \
            const unsigned char* b_ = real_ ? wbase + (long)(KT) * 128 : reinterpret_cast<const unsigned char*>(p.conv_zero); \
            asm volatile("" : "+s"(b_));                                                                          \
            _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                    \
                unsigned o_ = real_ ? wq[HH][i_] : 0u;                                                            \
                asm volatile("" : "+v"(o_));                                                                      \
                __builtin_amdgcn_global_load_lds((gptr_t)(b_ + o_), (lptr_t)(smem + (SLOT) * HT + (wv * 2 + i_) * 1024), 16, 0, 0); \
            }                                                                                                     \
        } else {                                                                                                  \
            SP_REQ(wbase, wq[HH], SLOT, KT);                                                                      \
        }                                                                                                         \
    }
#define SP_REQ(BASE, OFFS, SLOT, KT)                                                                              \
    {                                                                                                             \
        const unsigned char* b_ = (BASE) + (long)(KT) * 128;                                                      \
        asm volatile("" : "+s"(b_));                                                                              \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                        \
            unsigned o_ = OFFS[i_];                                                                               \
            asm volatile("" : "+v"(o_));                                                                          \
            __builtin_amdgcn_global_load_lds((gptr_t)(b_ + o_), (lptr_t)(smem + (SLOT) * HT + (wv * 2 + i_) * 1024), 16, 0, 0); \
        }                                                                                                         \
    }
    // ---- read side: fragment offsets (set per tile, from a fresh lane id: sixteen registers that do not live across the epilogue)
    int xl0[4], wl0[4], xl1[4], wl1[4];
    auto set_read = [&]() {
        const int ln = fresh_lane(), frow = ln & 31, fch = ln >> 5;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int fo = frow * 128 + (((kk * 2 + fch) ^ ((frow >> 1) & 7)) << 4);
            xl0[kk] = wm * 32 * 128 + fo;
            wl0[kk] = wn * 64 * 128 + fo;
            xl1[kk] = xl0[kk] + 4 * HT;
            wl1[kk] = wl0[kk] + 4 * HT;
            // (CONV has eight registers of gather state more: keep hipcc from folding slot offsets past the 16-bit immediate into further
            // address registers -- it spilled one, and its reload is a vmcnt(0) in the tile's tail)
            if constexpr (CONV) asm volatile("" : "+v"(xl0[kk]), "+v"(wl0[kk]), "+v"(xl1[kk]), "+v"(wl1[kk]));
        }
    };
    u32x4 xa[4], xb[4], wf[2][4];
    f32x16 acc[4][2];
#define SP_RX(XR, SLOT) { _Pragma("unroll") for (int kk_ = 0; kk_ < 4; ++kk_) XR[kk_] = *reinterpret_cast<const u32x4*>(smem + ((SLOT) < 4 ? xl0[kk_] : xl1[kk_]) + ((SLOT) & 3) * HT); }
#define SP_RW(SLOT)                                                                                               \
    {                                                                                                             \
        _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                                          \
            _Pragma("unroll") for (int kk_ = 0; kk_ < 4; ++kk_)                                                   \
                wf[j_][kk_] = *reinterpret_cast<const u32x4*>(smem + ((SLOT) < 4 ? wl0[kk_] : wl1[kk_]) + ((SLOT) & 3) * HT + j_ * 4096); \
    }
    // ZERO: the quadrant's first MFMAs of the tile take the constant-zero C operand (0 * anything + 0: same bits as a cleared register)
#define SP_MMA(XR, JH, I, ZERO)                                                                                   \
    {                                                                                                             \
        __builtin_amdgcn_s_setprio(1);                                                                            \
        _Pragma("unroll") for (int kk_ = 0; kk_ < 4; ++kk_)                                                       \
            _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) {                                                    \
                if ((ZERO) && kk_ == 0) {                                                                         \
                    f32x16 z_;                                                                                    \
                    _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) z_[r_] = 0.f;                               \
                    acc[(JH) * 2 + j_][I] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[j_][kk_]), __builtin_bit_cast(bf16x8, XR[kk_]), z_, 0, 0, 0); \
                } else {                                                                                          \
                    Mfma<TI>::run(acc[(JH) * 2 + j_][I], wf[j_][kk_], XR[kk_]);                                   \
                }                                                                                                 \
            }                                                                                                     \
        __builtin_amdgcn_s_setprio(0);                                                                            \
    }
#define SP_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
#define SP_PHASE(READ, REQ, VMN, XR, JH, I, ZERO)     \
    {                                                 \
        READ;                                         \
        __builtin_amdgcn_sched_barrier(0);            \
        REQ;                                          \
        if constexpr ((VMN) >= 0) SP_VM((VMN) < 0 ? 0 : (VMN)); \
        __builtin_amdgcn_sched_barrier(0);            \
        __builtin_amdgcn_s_barrier();                 \
        __builtin_amdgcn_sched_barrier(0);            \
        SP_MMA(XR, JH, I, ZERO);                      \
        __builtin_amdgcn_sched_barrier(0);            \
        __builtin_amdgcn_s_barrier();                 \
        __builtin_amdgcn_sched_barrier(0);            \
    }

#if SA_P8_TIMING
    int dbg_tile = 0;
#define SP_STAMP(IDX)                                                                                                       \
    {                                                                                                                       \
        if ((wv == 0 || wv == 4) && dbg_tile < 24 && p.dbg) {                                                                \
            const long long t_ = (long long)__builtin_readcyclecounter();                                                    \
            if (fresh_lane() == 0) p.dbg[(((long)blockIdx.x * 2 + (wv >> 2)) * 24 + dbg_tile) * 12 + (IDX)] = t_;             \
        }                                                                                                                   \
    }
#else
#define SP_STAMP(IDX) {}
#endif
    int tile_m = 0, tile_n = 0;
    if (!next_tile(tile_m, tile_n)) return;
    set_req(tile_m, tile_n);
    if constexpr (CONV) { set_conv(tile_m, 0, 0); set_conv(tile_m, 0, 1); set_conv(tile_m, 1, 0); set_conv(tile_m, 1, 1); }
    SP_REQX(0, 0, 0); SP_REQW(0, 1, 0); SP_REQX(1, 2, 0); SP_REQW(1, 3, 0);
    SP_REQX(0, 4, 1); SP_REQW(0, 5, 1);
    SP_VM(0);                                                    // the first pair's early phases run without counted waits (below)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    unsigned char* stage = smem + 8 * HT + wv * 4096;            // this wave's private 4 KiB
    for (;;) {
        const int m0 = tile_m * BM, n0 = tile_n * BN;
        int nt_m = 0, nt_n = 0;
        const bool have_next = next_tile(nt_m, nt_n);
        set_req(tile_m, tile_n);                                 // (again, although the previous tile's tail already set them for its six look-ahead requests:
        set_read();                                              //  recomputed, the offsets are not live across the epilogue -- which needs the registers)
        // (CONV: the gather state of this tile was set by the previous tile's tail -- or the prologue -- and stays live across the epilogue;
        // the residual epilogue has no eight registers for that -- 20 bytes of scratch -- and recomputes it)
        if constexpr (CONV && EPI == EPI_RESIDUAL) { set_conv(tile_m, 0, 0); set_conv(tile_m, 0, 1); set_conv(tile_m, 1, 0); set_conv(tile_m, 1, 1); }
        // The second wave of every SIMD runs one barrier behind the first -- INSIDE a tile's K loop only. The two groups are re-aligned
        // behind the last phase (below) and staggered again here: with the stagger carried through the epilogue, the trailing group waited
        // at its last barrier for the leading group's whole epilogue and the leading group then waited in phase 0 for the trailing group's
        // (tools/microbench/p8_timing.hip: 2 x 9.3k of a 63.7k-cycle tile at K = 1280 -- the two epilogues ran one after the other).
        if (wv >= 4) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        SP_STAMP(0);
        {   // first two K-tiles of the tile: everything phases 0..3 read was requested before the previous tile's stores and retired
            // by the vmcnt(0) in front of them (or by the prologue's) -- no counted wait until phase 4
            SP_PHASE({ SP_RX(xa, 0); SP_RW(1); }, SP_REQX(1, 6, 1), -1, xa, 0, 0, true);       // X0(0) is read HERE, not in the previous tile's last phase: 16 registers less across the epilogue
            SP_PHASE(SP_RX(xb, 2), SP_REQW(1, 7, 1), -1, xb, 0, 1, true);
            SP_PHASE(SP_RW(3),     SP_REQX(0, 0, 2), -1, xb, 1, 1, true);
            SP_PHASE(SP_RX(xb, 4), SP_REQW(0, 1, 2), -1, xa, 1, 0, true);
            SP_STAMP(1);
            SP_PHASE(SP_RW(5),     SP_REQX(1, 2, 2),  8, xb, 0, 0, false);
            SP_STAMP(2);
            SP_PHASE(SP_RX(xa, 6), SP_REQW(1, 3, 2),  8, xa, 0, 1, false);
            SP_PHASE(SP_RW(7),     SP_REQX(0, 4, 3),  8, xa, 1, 1, false);
            SP_PHASE(SP_RX(xa, 0), SP_REQW(0, 5, 3),  8, xb, 1, 0, false);
        }
        SP_STAMP(3);
        for (int pi = 1; pi < pairs - 1; ++pi) {
            const int t = 2 * pi;
            SP_PHASE(SP_RW(1),     SP_REQX(1, 6, t + 1), 8, xa, 0, 0, false);
            SP_PHASE(SP_RX(xb, 2), SP_REQW(1, 7, t + 1), 8, xb, 0, 1, false);
            SP_PHASE(SP_RW(3),     SP_REQX(0, 0, t + 2), 8, xb, 1, 1, false);
            SP_PHASE(SP_RX(xb, 4), SP_REQW(0, 1, t + 2), 8, xa, 1, 0, false);
            SP_PHASE(SP_RW(5),     SP_REQX(1, 2, t + 2), 8, xb, 0, 0, false);
            SP_PHASE(SP_RX(xa, 6), SP_REQW(1, 3, t + 2), 8, xa, 0, 1, false);
            SP_PHASE(SP_RW(7),     SP_REQX(0, 4, t + 3), 8, xa, 1, 1, false);
            SP_PHASE(SP_RX(xa, 0), SP_REQW(0, 5, t + 3), 8, xb, 1, 0, false);
        }
        SP_STAMP(4);
        unsigned bpack_raw = 0, bias_mask = 0;
        {   // last two K-tiles. With another tile behind them its first six half-tiles take the ring's next six requests (same slots, same
            // distances, same counted waits); without, nothing is requested after
            // W1(nk - 1) and the counted waits shrink with what is still in flight. ONE code path (the branches are wave-uniform and
            // enclose requests and waits only): two copies of the phases, each redefining all accumulators, made hipcc spill them.
            const int t = nk - 2;
            // CONV: the next tile's gather state, a quarter per phase (one (half, load) pair is ~350 cycles of address arithmetic -- about one
            // phase's MFMA time -- and the whole of it in one load segment stood exposed). X0's was free since the last middle phase, X1's is
            // after this tile's last X request (phase 0); the next tile's X0 / X1 requests are in phases 2 / 4.
            if constexpr (CONV) { if (have_next) set_conv(nt_m, 0, 0); }
            SP_PHASE(SP_RW(1),     SP_REQX(1, 6, t + 1), 8, xa, 0, 0, false);
            if constexpr (CONV) { if (have_next) set_conv(nt_m, 0, 1); }
            SP_PHASE(SP_RX(xb, 2), SP_REQW(1, 7, t + 1), 8, xb, 0, 1, false);
            SP_STAMP(9);
            if (have_next) set_req(nt_m, nt_n);
            if constexpr (CONV) { if (have_next) set_conv(nt_m, 1, 0); }
            SP_STAMP(10);
            SP_PHASE(SP_RW(3),     { if (have_next) { SP_REQX(0, 0, 0); SP_VM(8); } else { SP_VM(6); } }, -1, xb, 1, 1, false);
            SP_STAMP(11);
            if constexpr (CONV) { if (have_next) set_conv(nt_m, 1, 1); }
            SP_PHASE(SP_RX(xb, 4), { if (have_next) { SP_REQW(0, 1, 0); SP_VM(8); } else { SP_VM(4); } }, -1, xa, 1, 0, false);
            SP_PHASE(SP_RW(5),     { if (have_next) { SP_REQX(1, 2, 0); SP_VM(8); } else { SP_VM(2); } }, -1, xb, 0, 0, false);
            SP_PHASE(SP_RX(xa, 6), { if (have_next) { SP_REQW(1, 3, 0); SP_VM(8); } else { SP_VM(0); } }, -1, xa, 0, 1, false);
            SP_PHASE(SP_RW(7),     { if (have_next) { SP_REQX(0, 4, 1); SP_VM(8); } },                    -1, xa, 1, 1, false);
            // this wave's bias slice: lane l holds columns wn * 128 + 2 l, + 1 (packed bf16 pair). Requested in the tile's LAST phase: one
            // register for one phase (at the top of the tile it lived through the whole K loop and the RoPE instantiation spilled it behind
            // a vmcnt(0) -- at the top of a tile that waits for the previous tile's stores); one more op in the request stream, older than
            // this phase's request, so the counted wait still retires the half-tile it is for.
            // Branch-free on "no bias": a null bias reads a valid dummy address and is masked to zero (a named bool or a branch on it lived as
            // a 0/1 VGPR across the tile loop and was spilled; its reload at the top of a tile waited vmcnt(0) = for the previous tile's
            // stores). Adding +0 is exact here: an accumulator is never -0 (it starts from +0), everything else is unchanged by + 0.
            const TI* bias_p = p.bias ? p.bias : p.W;
            bias_mask = p.bias ? 0xffffffffu : 0u;
            bpack_raw = *reinterpret_cast<const unsigned*>(bias_p + min(n0 + wn * 128 + 2 * fresh_lane(), p.N - 2));
            SP_PHASE({},           { if (have_next) { SP_REQW(0, 5, 1); SP_VM(8); } },                    -1, xb, 1, 0, false);
            if (wv < 4) __builtin_amdgcn_s_barrier();             // the leading half meets the trailing half's last barrier: both enter the epilogue together
        }

        SP_STAMP(5);
        // ---- wave-private epilogue. Accumulator (j, i), register 4 g + r = output row wm * 64 + i * 32 + (lane & 31), column
        // wn * 128 + j * 32 + g * 8 + (lane >> 5) * 4 + r.
        // vmcnt retires loads AND stores in issue order, so a load issued behind a store cannot be waited for without waiting for that
        // store's acknowledgement. The epilogue therefore runs in two passes: (1) registers only -- bias, activation / rotary (its table
        // loads), rounding, packed in place; then the residual rows of the whole 64 x 128 block; ONE vmcnt(0) (which also retires the next
        // tile's first half-tiles); (2) LDS staging + stores only, block by block (32 rows x 64 output columns = 4 KiB), no load behind
        // the first store.
        constexpr bool GLU = (EPI == EPI_SWIGLU || EPI == EPI_GEGLU);
        const int n_out = GLU ? p.N / 2 : p.N, n0_out = GLU ? n0 / 2 : n0;
        const int lane_e = fresh_lane();                         // the epilogue's addresses are functions of the lane id only: not hoistable either
        const int srow = lane_e & 31, lhi = lane_e >> 5;
        // read-back geometry of a block: iteration `it` covers rows it * 8 + (lane >> 3), 16-byte chunk lane & 7
        constexpr int NBLK = GLU ? 2 : 4;                        // GLU: a block spans all four j of an i (128 accumulator columns -> 64 outputs)
        const int rrow = lane_e >> 3, rc = lane_e & 7;
        [[maybe_unused]] u32x4 res[NBLK][4];                     // the residual rows of the wave's whole 64 x 128 block, requested FIRST: pass 1 covers their latency
        // global addresses of the epilogue: scalar tile origin + 32-bit lane offset (tile-local row x pitch + column), like the requests
        [[maybe_unused]] const unsigned char* r_tile = reinterpret_cast<const unsigned char*>(p.R + (long)m0 * p.ldr + n0_out);
        unsigned char* c_tile = reinterpret_cast<unsigned char*>(p.C + (long)m0 * p.ldc + n0_out);
        const int rows_left = p.M - m0, cols_left = n_out - n0_out;           // valid rows / output columns from the tile origin
        if constexpr (EPI == EPI_RESIDUAL) {
#pragma unroll
            for (int blk = 0; blk < NBLK; ++blk)
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int rl = min(wm * 64 + (blk >> 1) * 32 + it * 8 + rrow, rows_left - 1), cl = min(wn * 128 + (blk & 1) * 64 + rc * 8, cols_left - 8);
                    res[blk][it] = *reinterpret_cast<const u32x4*>(r_tile + (unsigned)(rl * (int)(p.ldr * sizeof(TO)) + cl * (int)sizeof(TO)));
                }
        }
        __builtin_amdgcn_sched_barrier(0);
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));       // (native vector type: arrays of HIP's uint2 struct are demoted to scratch)
        using Pk = typename std::conditional<GLU, unsigned, u32x2>::type;     // the lane's 4 (GLU: 2) output values of one (j, i, g), packed
        const unsigned bpack = bpack_raw & bias_mask;
        Pk pk[4][2][4];
#pragma unroll
        for (int jp = 0; jp < 2; ++jp)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                // rotary table entries of this (row block, column-block pair): ONE batch of eight 16-byte loads, then the arithmetic (loaded
                // where they are used, each load is followed by its own vmcnt(0): 32 serial L2 round trips per tile in the first ISA)
                [[maybe_unused]] float4 rope_v[2][4];
                if constexpr (EPI == EPI_ROPE) {
                    const int m = min(m0 + wm * 64 + i * 32 + srow, p.M - 1), half = p.rope_D >> 1;
#pragma unroll
                    for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int n = min(min(n0 + wn * 128 + (jp * 2 + j2) * 32 + g * 8 + lhi * 4, p.N - 4), p.rope_cols - 4);
                            rope_v[j2][g] = *reinterpret_cast<const float4*>(p.rope + (long)m * half + ((n % p.rope_D) >> 1));
                        }
                }
#pragma unroll
                for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int j = jp * 2 + j2;
                        const int wcol = j * 32 + g * 8 + lhi * 4;           // column inside the wave's 128
                        // (ds_bpermute directly: __shfl adds `lane & ~63`, a lane-id term hipcc hoists out of the tile loop and spills)
                        const unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute((wcol >> 1) << 2, (int)bpack), hi = (unsigned)__builtin_amdgcn_ds_bpermute(((wcol >> 1) + 1) << 2, (int)bpack);
                        float v[4] = {acc[j][i][4 * g], acc[j][i][4 * g + 1], acc[j][i][4 * g + 2], acc[j][i][4 * g + 3]};
                        v[0] += __uint_as_float(lo << 16); v[1] += __uint_as_float(lo & 0xffff0000u);
                        v[2] += __uint_as_float(hi << 16); v[3] += __uint_as_float(hi & 0xffff0000u);
                        if constexpr (EPI == EPI_GELU) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = gelu_epi<TI>(v[r]);
                        } else if constexpr (EPI == EPI_HARDSWISH) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = hardswish_f(v[r]);
                        } else if constexpr (EPI == EPI_RELU) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                        }
                        if constexpr (EPI == EPI_ROPE) {
                            const int n = min(n0 + wn * 128 + wcol, p.N - 4);
                            if (n < p.rope_cols) {
                                // reference order (encoder/__init__.py:188-199), products and sums pinned as in gemm_nt_kernel's epilogue
                                const float4 cs = rope_v[j2][g];
                                const float x0 = Ty<TO>::rnd(v[0]), x1 = Ty<TO>::rnd(v[1]), x2 = Ty<TO>::rnd(v[2]), x3 = Ty<TO>::rnd(v[3]);
                                v[0] = __fmaf_rn(x0, cs.x, -__fmul_rn(x1, cs.y)); v[1] = __fmaf_rn(x1, cs.x, __fmul_rn(x0, cs.y));
                                v[2] = __fmaf_rn(x2, cs.z, -__fmul_rn(x3, cs.w)); v[3] = __fmaf_rn(x3, cs.z, __fmul_rn(x2, cs.w));
                            }
                        }
                        if constexpr (GLU) {
                            if constexpr (EPI == EPI_GEGLU) {
                                const float g0 = Ty<TO>::rnd(gelu_tanh_f(Ty<TO>::rnd(v[0]))), g1 = Ty<TO>::rnd(gelu_tanh_f(Ty<TO>::rnd(v[2])));
                                pk[j][i][g] = pack2(g0 * Ty<TO>::rnd(v[1]), g1 * Ty<TO>::rnd(v[3]));
                            } else {
                                pk[j][i][g] = pack2(silu_epi<TI>(v[0]) * v[1], silu_epi<TI>(v[2]) * v[3]);
                            }
                        } else {
                            pk[j][i][g] = Pk{pack2(v[0], v[1]), pack2(v[2], v[3])};
                        }
                    }
            }
        __builtin_amdgcn_sched_barrier(0);                       // pass 1's arithmetic stays in front of the drain (it covers the look-ahead requests' landing)
        SP_STAMP(6);
        SP_VM(0);                                                // every load in flight has landed; from here on this wave issues only LDS traffic and stores
        SP_STAMP(7);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int blk = 0; blk < NBLK; ++blk) {
            const int i = GLU ? blk : blk >> 1, jh = GLU ? 0 : blk & 1;
            const int cl = wn * (GLU ? 64 : 128) + jh * 64 + rc * 8;          // tile-local output column of this lane's chunk
            if constexpr (GLU) {
                // a lane holds 2 outputs per (j, g): columns j * 16 + g * 4 + lhi * 2, + 1 -- 4-byte LDS writes, 16 per block, 4-way bank
                // conflicts (7.3k cycles per tile against 3.7k for a plain epilogue, tools/microbench/p8_timing.hip). v_permlane32_swap trades
                // the halves of a (g, g + 1) pair between lane l and l + 32 (same row): afterwards each lane owns FOUR consecutive outputs
                // (lhi = 0: those of g, lhi = 1: those of g + 1) and writes 8 bytes, as the plain epilogues do.
#pragma unroll
                for (int j2 = 0; j2 < 4; ++j2)
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) {
                        const auto sw = __builtin_amdgcn_permlane32_swap(pk[j2][i][2 * gp], pk[j2][i][2 * gp + 1], false, false);
                        const int boff = (j2 * 16 + gp * 8 + lhi * 4) * (int)sizeof(TO);
                        typedef unsigned int u32x2s __attribute__((ext_vector_type(2)));
                        *reinterpret_cast<u32x2s*>(stage + srow * 128 + ((((boff >> 4) ^ (srow & 7)) << 4) | (boff & 15))) = u32x2s{sw[0], sw[1]};
                    }
            } else {
#pragma unroll
                for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int boff = (j2 * 32 + g * 8 + lhi * 4) * (int)sizeof(TO);       // output column inside the block, in bytes
                        *reinterpret_cast<Pk*>(stage + srow * 128 + ((((boff >> 4) ^ (srow & 7)) << 4) | (boff & 15))) = pk[jh * 2 + j2][i][g];
                    }
            }
            // read the block back as whole 16-byte chunks of contiguous rows (this wave's own LDS writes: the LDS executes a wave's
            // operations in order, no barrier)
            u32x4 raw[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int r = it * 8 + rrow;
                raw[it] = *reinterpret_cast<const u32x4*>(stage + r * 128 + ((rc ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int rl = wm * 64 + i * 32 + it * 8 + rrow;
                if (rl >= rows_left || cl >= cols_left) continue;
                TO* dst = reinterpret_cast<TO*>(c_tile + (unsigned)(rl * (int)(p.ldc * sizeof(TO)) + cl * (int)sizeof(TO)));
                if constexpr (EPI == EPI_RESIDUAL) {
                    float a[8], r8[8];
                    unpack16(make_uint4(raw[it][0], raw[it][1], raw[it][2], raw[it][3]), a, (TO*)nullptr);
                    unpack16(make_uint4(res[blk][it][0], res[blk][it][1], res[blk][it][2], res[blk][it][3]), r8, (TO*)nullptr);
#pragma unroll
                    for (int e = 0; e < 8; e += 4) store4(dst + e, a[e] + r8[e], a[e + 1] + r8[e + 1], a[e + 2] + r8[e + 2], a[e + 3] + r8[e + 3]);
                    __builtin_amdgcn_sched_barrier(0);           // one chunk at a time: interleaved, the unpacked chunks of a block spill (and a reload behind a store waits for that store)
                } else {
                    *reinterpret_cast<u32x4*>(dst) = raw[it];
                }
            }
        }
        SP_STAMP(8);
#if SA_P8_TIMING
        ++dbg_tile;
#endif
        if (!have_next) return;
        tile_m = nt_m; tile_n = nt_n;
    }
#undef SP_STAMP
#undef SP_PHASE
#undef SP_VM
#undef SP_MMA
#undef SP_RW
#undef SP_RX
#undef SP_REQX
#undef SP_REQW
#undef SP_REQ
}
#undef SA_COMPUTE
#undef SA_COMPUTE_FULL
#undef SA_COMPUTE_LEAN
#undef SA_FRAGS
#undef SA_MFMAS

// Optional per-launch timing with HIP events on the launch stream (bench.py's `roofline` object): one record per
// tile configuration, accumulating launches, algorithmic FLOPs / bytes and event-measured milliseconds.
struct GemmProfiler {
    static constexpr int NCFG = 4, POOL = 8192;
    bool enabled = false;
    hipEvent_t ev[2 * POOL];
    bool have_events = false;
    int n = 0;
    int cfg_of[POOL];
    double flops_of[POOL], bytes_of[POOL], slab_of[POOL];   // bytes_of: operands + the result once; slab_of: fp32 split-K partial slabs (a by-product of the decomposition)
    void start() {
        if (!have_events) {
            for (int i = 0; i < 2 * POOL; ++i) (void)hipEventCreate(&ev[i]);
            have_events = true;
        }
        n = 0; enabled = true;
    }
};
inline GemmProfiler& gemm_profiler() { static GemmProfiler p; return p; }
// profiler buckets: 0 = 128x128 (large GEMMs), 1 = tall 256-row tiles (decode regime), 2 = small tiles
inline int gemm_cfg_id(int BM, int BN) { return (BM >= 128 && BN >= 128) ? 0 : (BM == 256 ? 1 : 2); }

template <typename TI, typename TO, int BM, int BN, int WM, int WN, int EPI, bool SPLIT = false, int GLDS = 0, bool CONV = false, int WAUX = 0>
static inline int launch_gemm_cfg(const GemmArgs<TI, TO>& a, hipStream_t s) {
    int tiles = SPLIT ? cdiv(cdiv(a.N, BN) * a.splitk, 8) * 8 * cdiv(a.M, BM) : cdiv(a.M, BM) * cdiv(a.N, BN);
    a.bn_used = BN;
    GemmArgs<TI, TO> aa = a;
    if (!SPLIT && BM >= 128 && BN >= 128) {          // XCD-aware super-tiles for the large-tile configurations
        const int tm = cdiv(a.M, BM), tn = cdiv(a.N, BN);
        constexpr int GRP = ((BM + BN) * 128 * (GLDS > 2 && GLDS != 8 ? GLDS : 2) > 80 * 1024) ? 32 : 64;
        if (tm * tn >= 8 * GRP) {
            aa.swz_n = cdiv(tn, cdiv(tn, 8));                 // equal-width super-columns of <= 8 tile columns
            aa.swz_m = std::max(1, GRP / aa.swz_n);
            tiles = cdiv(tm * tn, 8 * GRP) * 8 * GRP;
        }
    }
    if (!SPLIT && aa.mgroup) tiles = cdiv(cdiv(a.N, BN), 8) * 8 * cdiv(a.M, BM);
    constexpr size_t out_w = ((EPI == EPI_SWIGLU || EPI == EPI_GEGLU) && !SPLIT) ? BN / 2 : BN;
    constexpr size_t out_bytes = (size_t)BM * out_w * (SPLIT ? sizeof(float) : sizeof(TO));
    constexpr size_t stage_bytes = (size_t)(BM + BN) * 128 * (GLDS > 2 && GLDS != 8 ? GLDS : 2);
    constexpr bool argmax_direct = (EPI == EPI_ARGMAX) && ((size_t)BM * BN * 4 > 160 * 1024);     // partials straight from the accumulators
    constexpr size_t lds = (argmax_direct || stage_bytes > out_bytes) ? stage_bytes : out_bytes;   // staging buffers are reused for the output tile
    auto kern = gemm_nt_kernel<TI, TO, BM, BN, WM, WN, EPI, SPLIT, GLDS, CONV, WAUX>;
    static AttrOnce attr;           // >64 KiB dynamic LDS needs the opt-in attribute; harmless below
    attr.ensure(kern, lds);
    GemmProfiler& pf = gemm_profiler();
    const bool prof = pf.enabled && pf.n < GemmProfiler::POOL;
    if (prof) (void)hipEventRecord(pf.ev[2 * pf.n], s);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(64 * WM * WN), lds, s, aa);
    if (prof) {
        (void)hipEventRecord(pf.ev[2 * pf.n + 1], s);
        pf.cfg_of[pf.n] = CONV ? 3 : gemm_cfg_id(BM, BN);       // bucket 3 = implicit-GEMM convolutions
        pf.flops_of[pf.n] = CONV ? 2.0 * a.M * a.N * a.cTaps * a.cCin : 2.0 * a.M * a.N * a.K;
        const double outn = (EPI == EPI_SWIGLU || EPI == EPI_GEGLU) ? a.N / 2 : (EPI == EPI_ARGMAX ? 4.0 * cdiv(a.N, BN) : a.N);
        const double xelems = CONV ? (double)a.M / std::max(1, a.cHo * a.cWo) * a.cH * a.cW * a.cCin : (double)a.M * a.K;   // the input tensor once
        // algorithmic bytes: X + W + the result ONCE in the storage type (what an unsplit GEMM of this shape would move). The fp32
        // partial slabs a split-K launch actually writes are the kernel's own decomposition, not the problem's: counted apart
        // (the bucket's frac 0.148 came from counting them as algorithmic; 0.126 without).
        pf.bytes_of[pf.n] = (xelems + (double)a.N * a.K) * sizeof(TI) + (double)a.M * outn * (SPLIT ? sizeof(TI) : sizeof(TO)) +
                            (EPI == EPI_RESIDUAL ? (double)a.M * a.N * sizeof(TO) : 0.0);
        pf.slab_of[pf.n] = SPLIT ? (double)a.splitk * a.M * a.N * 4.0 : 0.0;
        ++pf.n;
    }
    return (int)hipGetLastError();
}

// Launch of the persistent 8-phase tile loop: one workgroup per CU (160 KB of LDS each), XCD-aware super-tiles as in launch_gemm_cfg.
template <typename TI, typename TO, int EPI, int CONV = 0>
static inline int launch_gemm_persist(const GemmArgs<TI, TO>& a, hipStream_t s) {
    constexpr int BM = 256, BN = 256, GRP = 32;
    const int tm = cdiv(a.M, BM), tn = cdiv(a.N, BN);
    GemmArgs<TI, TO> aa = a;
    a.bn_used = BN;
    aa.swz_n = cdiv(tn, cdiv(tn, 8));
    aa.swz_m = std::max(1, GRP / aa.swz_n);
    const int padded = cdiv(tm * tn, 8 * GRP) * 8 * GRP;
    static int n_cu_of[64] = {};                             // per device (one cached count served every device of the process)
    int dev = 0;
    (void)hipGetDevice(&dev);
    int& n_cu = n_cu_of[dev & 63];
    if (!n_cu) {
        if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
        n_cu = (n_cu / 8) * 8;                              // whole XCD rounds: virtual block b stays on XCD b % 8 in every round
    }
    const int grid = std::min(padded, n_cu);
    constexpr size_t lds = (size_t)2 * (BM + BN) * 128 + 8 * 4096;      // the half-tile ring + 4 KiB of epilogue staging per wave
    auto kern = gemm_nt_p8p_kernel<TI, TO, EPI, CONV>;
    static AttrOnce attr;
    attr.ensure(kern, lds);
    GemmProfiler& pf = gemm_profiler();
    const bool prof = pf.enabled && pf.n < GemmProfiler::POOL;
    if (prof) (void)hipEventRecord(pf.ev[2 * pf.n], s);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, aa);
    if (prof) {
        (void)hipEventRecord(pf.ev[2 * pf.n + 1], s);
        pf.cfg_of[pf.n] = CONV ? 3 : gemm_cfg_id(BM, BN);       // bucket 3 = implicit-GEMM convolutions
        pf.flops_of[pf.n] = CONV ? 2.0 * a.M * a.N * a.cTaps * a.cCin : 2.0 * a.M * a.N * a.K;
        const double outn = (EPI == EPI_SWIGLU || EPI == EPI_GEGLU) ? a.N / 2 : a.N;
        // CONV: the input TENSOR once, not the im2col matrix M x K (~KH * KW times larger), as launch_gemm_cfg prices the same bucket
        const double xelems = CONV ? (double)a.M / std::max(1, a.cHo * a.cWo) * a.cH * a.cW * a.cCin : (double)a.M * a.K;
        pf.bytes_of[pf.n] = (xelems + (double)a.N * a.K) * sizeof(TI) + (double)a.M * outn * sizeof(TO) +
                            (EPI == EPI_RESIDUAL ? (double)a.M * a.N * sizeof(TO) : 0.0);
        pf.slab_of[pf.n] = 0.0;
        ++pf.n;
    }
    return (int)hipGetLastError();
}

// Tile choice. M <= 256 is the decode regime: one workgroup spans all rows (weights stream from HBM once), BN picked so
// the grid has >= ~128 workgroups where N allows. Larger M uses 128x128 tiles, with 64x64 for problems too small to
// fill the chip.
template <typename TI, typename TO, int EPI>
static inline int launch_gemm(const GemmArgs<TI, TO>& a, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0) return SA_OK;
    {   // rows are stored in 16-byte chunks: output width (N, or N/2 after SwiGLU) must be a multiple of 16 bytes of TO
        const int n_out = (EPI == EPI_SWIGLU || EPI == EPI_GEGLU) ? a.N / 2 : a.N;
        if (a.K % Ty<TI>::KE != 0 || a.N % 4 != 0 || n_out % (16 / (int)sizeof(TO)) != 0 || a.ldc % (16 / (int)sizeof(TO)) != 0)
            return SA_ERR_SHAPE;
    }
    if (a.M <= 256) {
        if constexpr (EPI == EPI_ARGMAX) {
            // the fused lm_head hands per-column-block (max, sum-exp) partials to greedy_head: keep the block width a function
            // of N alone so a line's score is summed in the same groups whatever the number of active rows
            // Round 4: lm_head-sized N runs as ONE round of 256-row x 320-column tiles (81920 = 256 x 320: a tile per CU). The 128x128
            // tiling asked L2 for 839 MB to stream 210 MB of weights (each X row block re-read by 640 column tiles, each W tile by 2 row
            // tiles) -- 110 us in situ at the ~15 TB/s L2->LDS ceiling plus cold HBM; this tile asks for 378 MB. Fewer active rows
            // still take the tile (clamped rows): the column grouping, and with it a line's score bits, must not depend on M.
            if (a.N >= 64 * 512 && tuning().lmhead && (a.K / Ty<TI>::KE) % 2 == 0 && a.N % 4 == 0) {
                if constexpr (sizeof(TI) == 2)
                    if (tuning().lmhead == 2) return launch_gemm_cfg<TI, TO, 256, 320, 4, 2, EPI, false, 2, false, 2>(a, s);   // non-temporal W
                return launch_gemm_cfg<TI, TO, 256, 320, 4, 2, EPI, false, 2>(a, s);
            }
            if (a.N >= 64 * 512) return launch_gemm_cfg<TI, TO, 128, 128, 2, 2, EPI, false, 2>(a, s);
            return launch_gemm_cfg<TI, TO, 64, 64, 2, 2, EPI, false, 2>(a, s);
        }
        if (a.M > 128) {
            // measured at M = 256 (tools/microbench/gemm_shapes.py, decode_sweep.py): lm_head-sized N -> 128x128 tiles, everything
            // else 64x64, direct-to-LDS. Larger gate|up tiles (128x64: +26 us/step, 128x128: +69), a 256x128 lm_head tile (+30) and
            // 3- / 4-stage rings for lm_head (+25) all lost in r02 (profiles/r02_decode_sweeps.md).
            if (a.N >= 64 * 512) return launch_gemm_cfg<TI, TO, 128, 128, 2, 2, EPI, false, 2>(a, s);
            if constexpr (EPI == EPI_SWIGLU && sizeof(TI) == 2) {
                // decode gate|up (M = 256, N = 10240, K = 1280): 4 MFMAs per wave and K-tile against a ~700-cycle L2 round trip -- with two
                // stages a workgroup's 20 K-tiles are 20 serial round trips. Round 5 A/B: a 3- / 4-stage ring (the split-K tiles' loop) keeps
                // 2 / 3 K-tiles in flight per workgroup; 48 KiB keeps all 640 workgroups resident (3 per CU), 64 KiB only 512.
                if (tuning().gateup_ring == 3) return launch_gemm_cfg<TI, TO, 64, 64, 2, 2, EPI, false, 3>(a, s);
                if (tuning().gateup_ring == 4) return launch_gemm_cfg<TI, TO, 64, 64, 2, 2, EPI, false, 4>(a, s);
            }
            return launch_gemm_cfg<TI, TO, 64, 64, 2, 2, EPI, false, 2>(a, s);
        }
        if (a.M > 64) {
            if (a.N >= 64 * 128) return launch_gemm_cfg<TI, TO, 128, 64, 4, 1, EPI>(a, s);
            return launch_gemm_cfg<TI, TO, 128, 32, 4, 1, EPI>(a, s);
        }
        if (a.N >= 64 * 128) return launch_gemm_cfg<TI, TO, 64, 64, 2, 2, EPI>(a, s);
        return launch_gemm_cfg<TI, TO, 64, 32, 2, 1, EPI>(a, s);
    }
    if constexpr (EPI == EPI_ARGMAX) {
        // 257 ... 1024 (and more) rows (round 6; the reference's recognition_batch_size is open-ended, its README runs 864): the SAME 256 x 320
        // tile per 256-row block, so the per-column-block (max, sum-exp) partials -- and with them every line's score bits -- are the ones the
        // <= 256-row launch produces, whatever the slot count; the row blocks of a W tile share it through one XCD's L2 (mgroup)
        if (a.N >= 64 * 512 && tuning().lmhead && (a.K / Ty<TI>::KE) % 2 == 0 && a.N % 4 == 0) {
            GemmArgs<TI, TO> g = a;
            g.mgroup = 1;
            int rc;
            if constexpr (sizeof(TI) == 2) {
                if (tuning().lmhead == 2) { rc = launch_gemm_cfg<TI, TO, 256, 320, 4, 2, EPI, false, 2, false, 2>(g, s); a.bn_used = g.bn_used; return rc; }
            }
            rc = launch_gemm_cfg<TI, TO, 256, 320, 4, 2, EPI, false, 2>(g, s);
            a.bn_used = g.bn_used;
            return rc;
        }
    }
    if constexpr (EPI == EPI_SWIGLU && sizeof(TI) == 2 && sizeof(TO) == 2) {
        // decode (and small-prefill) gate|up above 256 rows (round 6): 4-6 row blocks x 40 column tiles of 256 x 256 are 160-240 workgroups -- too few for the cost model
        // below, but the 8-phase tile at 0.62-0.94 of the chip still beats 128 x 128 tiles (1024 rows: 47.7 -> 30 us per layer, 280 us per step;
        // at 768 rows = 120 tiles it loses 90 us per step). Same K order: tokens identical.
        const int mode = tuning().big_m_gateup;             // -1 = by rows, 0 = never, 1 = persistent loop, 2 = one tile per workgroup
        const int rb = cdiv(a.M, 256), t256 = rb * cdiv(a.N, 256);
        const int pick = mode >= 0 ? mode : ((rb >= 4 && t256 >= 160 && t256 < 256) ? 2 : 0);     // (256 tiles and more: the cost model below)
        if (pick && (a.K / Ty<TI>::KE) % 2 == 0 && a.K / Ty<TI>::KE >= 4 && a.N % 8 == 0) {
            if (pick == 1) return launch_gemm_persist<TI, TO, EPI>(a, s);
            return launch_gemm_cfg<TI, TO, 256, 256, 4, 2, EPI, false, 8>(a, s);
        }
    }
    if (a.N <= 64 && a.M >= 128 * 256) return launch_gemm_cfg<TI, TO, 128, 64, 4, 1, EPI>(a, s);   // narrow outputs (1x1 convs to 64 ch)
    const long big = (long)cdiv(a.M, 128) * cdiv(a.N, 128);
    const int glds = tuning().glds;
    if constexpr (sizeof(TO) == 2) {
        // 256x256 tiles (8 waves, 64x128 per wave): the 128x128 tile requests 2/128 bytes per MAC row/column pair from L2 and
        // saturates the L2->CU path at ~15 TB/s (= the same ceiling a load-only kernel reaches, tools/microbench/wstream.hip),
        // which caps it near 64 flop/B x 15 TB/s ~ 0.95 PF/s; doubling both tile edges halves that traffic.
        const int bigtile = tuning().bigtile;
        // pick by whole rounds of resident workgroups (256 x 1 per CU vs 512 x 2 per CU, in units of 128x128 tiles of work);
        // measured throughput ratio of the two kernels on full rounds: ~1.17 for the 2-stage 256x256 loop (r01 microbench), ~1.4 for the
        // persistent 8-phase loop (r05: 1.15-1.2 PF/s against 0.82-0.84 on the shapes where both can be timed). `bigtile_ratio_pct` sets it.
        const long t256 = (long)cdiv(a.M, 256) * cdiv(a.N, 256);
        const double ratio256 = tuning().bigtile_ratio_pct > 0 ? tuning().bigtile_ratio_pct / 100.0
                                                                : ((bigtile == 3 && sizeof(TI) == 2) ? 1.4 : 1.17);
        const double cost256 = (double)cdivl(t256, 256) * 256 * 4 / ratio256, cost128 = (double)cdivl(big, 512) * 512;
        // (r03: 256 x 128 tiles with a 3-stage ring -- 144 KB, deeper prefetch, 1.37x the L2 bytes per flop -- lost 5-15 % on every
        // encoder / prefill shape and on 8k^3 (1212 -> 1026 TF/s, profiles/r03_sweeps.txt): the big-tile loop is bound by L2 -> LDS bytes
        // per flop, not by prefetch depth.)
        if (bigtile && a.K >= tuning().bigtile_min_k && ((t256 >= 256 && cost256 <= cost128) || tuning().bigtile_any)) {
            if constexpr (EPI != EPI_ARGMAX) {
                const int nk = a.K / Ty<TI>::KE;
                if constexpr (sizeof(TI) == 2)
                    if (tuning().persist && nk >= 4 && nk % 2 == 0 && a.N % 8 == 0) return launch_gemm_persist<TI, TO, EPI>(a, s);
            }
            // bigtile = 2: the same tile on FOUR waves (128 x 128 per wave, 256 accumulator registers in AGPRs, one wave per SIMD):
            // two thirds of the LDS fragment bytes per MFMA of the 8-wave layout (32 KB per 64 MFMAs instead of 24 KB per 32). A third
            // stage does not fit (3 x 64 KB > 160 KB of LDS). Same K order and MFMA: bit-identical outputs.
            // A/B: tools/microbench/bigtile_ab.py 1 2.
            if constexpr (sizeof(TI) == 2 && EPI != EPI_ARGMAX) {
                if (bigtile == 2) return launch_gemm_cfg<TI, TO, 256, 256, 2, 2, EPI, false, 2>(a, s);
                // bigtile = 3: the 8-phase schedule (half-tile ring, counted vmcnt, two wave groups one barrier apart); even K-tile counts
                if (bigtile == 3 && (a.K / Ty<TI>::KE) % 2 == 0) return launch_gemm_cfg<TI, TO, 256, 256, 4, 2, EPI, false, 8>(a, s);
            }
            return launch_gemm_cfg<TI, TO, 256, 256, 4, 2, EPI, false, 2>(a, s);
        }
    }
    if (big >= 256) {
        if (glds == 2) return launch_gemm_cfg<TI, TO, 128, 128, 2, 2, EPI, false, 2>(a, s);
        if (glds == 3) return launch_gemm_cfg<TI, TO, 128, 128, 2, 2, EPI, false, 3>(a, s);
        return launch_gemm_cfg<TI, TO, 128, 128, 2, 2, EPI>(a, s);
    }
    return launch_gemm_cfg<TI, TO, 64, 64, 2, 2, EPI>(a, s);
}

// Split-K launch for the decode regime (M <= 256, small N): tiles x splitk workgroups so a skinny GEMM still covers the
// chip; raw fp32 partial sums go to a.part[splitk][M][N] and the NEXT kernel combines them (launch-boundary reduce).
// Returns the slice count actually used through a.splitk (caller passes the same struct to the consumer).
// The slice count is a function of (N, K) ONLY -- it is sized for a full 256-row batch (4 M-tiles) whatever the number of active
// rows: the fp32 partial sums of an output element are then grouped the same way at every batch size, and since every GEMM tile
// shape walks K in the same order with the same MFMA, a line's bf16 results no longer depend on how many other lines are active
// (round 1 picked the count from the actual M: near-tie argmaxes could flip between slot counts).
static inline int pick_splitk(int tiles_n, int nk) {
    const Tuning& t = tuning();
    const int tiles = tiles_n * 4;
    int s = (t.split_target + tiles / 2) / tiles;          // aim at ~target workgroups at M = 256
    s = std::min(s, nk / std::max(1, t.split_min_kt));    // keep enough K-tiles per slice to fill the pipeline
    return std::max(1, std::min(s, std::min(t.split_max, 8)));      // consumers hold <= 8 slabs in flight
}

template <typename TI>
static inline int launch_gemm_splitk(GemmArgs<TI, TI>& a, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0) return SA_OK;
    if (a.K % Ty<TI>::KE != 0 || a.N % 4 != 0 || !a.part) return SA_ERR_SHAPE;
    const int nk = a.K / Ty<TI>::KE;
    // 64x64 tiles with a 4-stage direct-to-LDS ring and ~256 workgroups. 128x64 / 128x128 split-K tiles (fewer L2->CU requests per
    // MAC, more slices) were measured in r02 and lose by 70-180 us per decode step (profiles/r02_decode_sweeps.md).
    a.splitk = pick_splitk(cdiv(a.N, 64), nk);
    // above 256 rows (round 6) the 64 x 64 tile re-reads every W slice once per 64 rows and every X row block once per 64 columns: the slice
    // count stays a function of (N, K) -- same K grouping, same bits -- and only the tile grows
    // (tools/microbench/decode_sweep.py --configs bigm, gpurun r06ad: at 1024 rows 128 x 128 tiles with the 4-stage ring take 290 us off the
    // 2772 us step -- 8 x 14 x 2 / 8 x 10 x 3 workgroups = one round of the chip; at 768 rows 160 us; at 384 / 512 rows they fill half the chip and
    // lose, and 128 x 64 tiles take 60-70 us off. Tokens identical in every arm.)
    if constexpr (sizeof(TI) == 2) {
        const int mode = tuning().big_m_split;              // -1 = by rows, 0 = 64 x 64 always, 2 / 3 = force
        // by workgroup count (one workgroup per CU for both larger tiles): 128 x 128 once it fills more than half the chip (640 rows: 140-150
        // workgroups, 1784 against 1890 us per step; 512 rows: 112-120, loses), else 128 x 64 while it stays within one round (384 / 512 rows:
        // 168-240 workgroups, 60-70 us per step; at 640 rows it needs 280-300 and loses)
        const int w128 = cdiv(a.M, 128) * cdiv(a.N, 128) * a.splitk, w12864 = cdiv(a.M, 128) * cdiv(a.N, 64) * a.splitk;
        const int pick = a.M <= 256 ? 0 : (mode >= 0 ? mode : (w128 >= 136 ? 2 : (w12864 <= 256 ? 3 : 0)));
        if (pick == 2) return launch_gemm_cfg<TI, TI, 128, 128, 2, 2, EPI_BIAS, true, 4>(a, s);
        if (pick == 3) return launch_gemm_cfg<TI, TI, 128, 64, 4, 1, EPI_BIAS, true, 4>(a, s);
    }
    return launch_gemm_cfg<TI, TI, 64, 64, 2, 2, EPI_BIAS, true, 4>(a, s);
}

}  // namespace sa
