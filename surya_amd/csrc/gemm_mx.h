// MXFP8 decode-regime GEMM for gfx950:  C[M,N] = X[M,K] . W[N,K]^T with both operands in OCP e4m3 and one E8M0 scale per
// 32 consecutive K elements of every row (OCP Microscaling), multiplied by v_mfma_scale_f32_32x32x64_f8f6f4, which applies
// both block scales inside the matrix pipe (BASELINE.json configs[4]'s fp8 weight path; there is no counterpart in the
// reference, which runs bf16/fp16 everywhere).
//
// Why it exists: at M <= 256 the decoder projections are bound by the bytes each CU has to pull through its load path
// (~58 GB/s per CU, tools/microbench/wstream), not by HBM or MFMA rate. fp8 operands halve those
// bytes for BOTH operands; the scales add 1/32.
//
// Operand / scale layout of the instruction (measured with tools/microbench/mx_probe.hip, the guides do not state it):
//   * lane l holds 32 bytes of row (l & 31): VGPR 0-3 = K elements 16 h .. 16 h + 15 of the 64-deep step, VGPR 4-7 =
//     elements 32 + 16 h .. 32 + 16 h + 15, h = l >> 5 -- i.e. of the step's four 16-byte chunks lane half h holds chunks h and
//     2 + h;
//   * the scale VGPR of lane l (byte selected by OPSEL) scales MX block h of the step (chunks 2 h and 2 h + 1) of row
//     (l & 31): both lane halves read each block, the half whose number equals the block's supplies its scale;
//   * C/D is the ordinary 32x32 map.
// The LDS image is the bf16 kernel's (gemm.h): 128-byte K-rows (= 128 elements = 4 MX blocks), 16-byte chunks XOR-swizzled
// on the source address of global_load_lds_dwordx4, a ring of STAGES buffers, raw s_barrier + counted vmcnt. The 4 scale
// bytes of a row for one K-tile are one dword, staged by global_load_lds_dword (64 rows per instruction) next to the tile.
// Scale tensors are therefore stored K-TILE-MAJOR: [K / 128][rows][4] bytes, so that instruction reads 256 contiguous bytes.
// (The first version kept them row-major [rows][K / 32]: every lane of the scale load then touched its own cache line, the
// scale gathers issued 4x the L2 requests of the data tiles and each K-tile cost ~0.8 us -- the K = 5120 down-projection
// ran SLOWER than in bf16, profiles/r02_fp8_decode.md.)
// As in gemm.h the WEIGHT fragment is the first MFMA operand (result D[n][m]: a lane owns 4 consecutive columns of one row).
//
// Epilogues: split-K fp32 slabs (consumers: decode attention, reduce + norm), SwiGLU -> MXFP8 (the next GEMM's operand,
// quantised from the fp32 accumulators: a 64-column tile yields exactly one 32-wide block per row), greedy-argmax partials
// (lm_head), plain fp32 (tests).
#pragma once
#include "gemm.h"

namespace sa {

typedef int i32x8 __attribute__((ext_vector_type(8)));

enum MxEpi { MX_EPI_F32 = 0, MX_EPI_SWIGLU = 1, MX_EPI_ARGMAX = 2 };

struct MxArgs {
    const uint8_t* X; long ldx; const uint8_t* SX;   // e4m3 [M][ldx >= K], e8m0 K-tile-major [K / 128][sx_rows][4]
    const uint8_t* W; long ldw; const uint8_t* SW;   // e4m3 [N][ldw >= K], e8m0 K-tile-major [K / 128][sw_rows][4]
    int M, N, K;                                     // K % 128 == 0, N % 4 == 0
    long sx_rows, sw_rows;                           // row capacity (= row stride in dwords) of the two scale tensors
    int splitk = 1;
    float* part = nullptr;                           // SPLIT: [splitk][M][N] raw fp32 partial sums
    float* C = nullptr; long ldc = 0;                // MX_EPI_F32
    uint8_t* Q = nullptr; long ldq = 0;              // MX_EPI_SWIGLU: e4m3 [M][N / 2] ...
    uint8_t* SQ = nullptr; long sq_rows = 0;         // ... and its e8m0 scales, K-tile-major [N / 256][sq_rows][4]
    float4* amax = nullptr;                          // MX_EPI_ARGMAX: {max, argmax bits, sum exp(v - max), 0} per (row, tile column)
    const bf16_t* bias = nullptr;                    // [N], added in the non-split F32 / ARGMAX epilogues (lm_head)
    mutable int bn_used = 0;
};

template <int BM, int BN, int EPI, bool SPLIT, int STAGES>
__global__ __launch_bounds__(256) void gemm_mx_kernel(MxArgs p) {
    constexpr int WM = 2, WN = 2, NT = 256, NW = 4;
    constexpr int WTM = BM / WM, WTN = BN / WN, FM = WTM / 32, FN = WTN / 32;
    constexpr int XBYTES = BM * 128, WBYTES = BN * 128, SXOFF = XBYTES + WBYTES, SWOFF = SXOFF + BM * 4, BUF = SWOFF + BN * 4;
    constexpr int XI = BM / 8 / NW, WI = BN / 8 / NW, SXI = BM / 64, SWI = BN / 64;
    constexpr int LPT = XI + WI + SXI + SWI;            // loads per K-tile per wave (scale dwords are fetched by every wave)
    static_assert(FM >= 1 && FN >= 1 && BM % 64 == 0 && BN % 64 == 0 && STAGES >= 2 && (STAGES - 2) * LPT <= 63, "tile");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    // the M-tiles that stream the same (W tile, K slice) sit on ONE XCD (workgroup b runs on XCD b % 8), see gemm.h
    const int xcd = (int)blockIdx.x & 7, jq = (int)blockIdx.x >> 3;
    const int pair = (jq / tiles_m) * 8 + xcd;
    if (pair >= tiles_n * p.splitk) return;
    const int tile_m = jq % tiles_m, tile_n = pair / p.splitk, ks = pair % p.splitk;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nk_all = p.K / 128;
    const int kt_begin = (int)((long)ks * nk_all / p.splitk), kt_end = (int)((long)(ks + 1) * nk_all / p.splitk);
    const int nk = kt_end - kt_begin, last = nk - 1;

    f32x16 acc[FN][FM];
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

    // ---- staging sources (rows clamped into range; the swizzle is applied to the source chunk)
    const unsigned char* xg[XI];
    const unsigned char* wg[WI];
    const unsigned char* sxg[SXI];
    const unsigned char* swg[SWI];
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        const int row = (wave * XI + i) * 8 + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
        xg[i] = p.X + (long)min(m0 + row, p.M - 1) * p.ldx + c * 16 + (long)kt_begin * 128;
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) {
        const int row = (wave * WI + i) * 8 + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
        wg[i] = p.W + (long)min(n0 + row, p.N - 1) * p.ldw + c * 16 + (long)kt_begin * 128;
    }
#pragma unroll
    for (int i = 0; i < SXI; ++i) sxg[i] = p.SX + ((long)kt_begin * p.sx_rows + min(m0 + i * 64 + lane, p.M - 1)) * 4;
#pragma unroll
    for (int i = 0; i < SWI; ++i) swg[i] = p.SW + ((long)kt_begin * p.sw_rows + min(n0 + i * 64 + lane, p.N - 1)) * 4;
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
#define SA_MX_ISSUE(BUFOFF, KT)                                                                                          \
    {                                                                                                                    \
        const long koff_ = (long)(KT) * 128, sxo_ = (long)(KT) * p.sx_rows * 4, swo_ = (long)(KT) * p.sw_rows * 4;       \
        _Pragma("unroll") for (int i = 0; i < XI; ++i) __builtin_amdgcn_global_load_lds(                                 \
            (gptr_t)(xg[i] + koff_), (lptr_t)(smem + (BUFOFF) + (wave * XI + i) * 1024), 16, 0, 0);                      \
        _Pragma("unroll") for (int i = 0; i < WI; ++i) __builtin_amdgcn_global_load_lds(                                 \
            (gptr_t)(wg[i] + koff_), (lptr_t)(smem + (BUFOFF) + XBYTES + (wave * WI + i) * 1024), 16, 0, 0);             \
        _Pragma("unroll") for (int i = 0; i < SXI; ++i) __builtin_amdgcn_global_load_lds(                                \
            (gptr_t)(sxg[i] + sxo_), (lptr_t)(smem + (BUFOFF) + SXOFF + i * 256), 4, 0, 0);                              \
        _Pragma("unroll") for (int i = 0; i < SWI; ++i) __builtin_amdgcn_global_load_lds(                                \
            (gptr_t)(swg[i] + swo_), (lptr_t)(smem + (BUFOFF) + SWOFF + i * 256), 4, 0, 0);                              \
    }
    const int frow = lane & 31, fh = lane >> 5;
    // One K-tile = two 64-deep MFMA steps. Fragment of step s: chunks 4 s + h and 4 s + 2 + h of the row; scale byte 2 s + h
    // of the row's dword = byte 2 s (OPSEL) of the dword shifted right by 8 h.
#define SA_MX_FRAG(DST, BASE, ROW, STEP)                                                                                 \
    {                                                                                                                    \
        const int sw_ = ((ROW) >> 1) & 7;                                                                                \
        const u32x4 lo_ = *reinterpret_cast<const u32x4*>((BASE) + (ROW) * 128 + (((4 * (STEP) + fh) ^ sw_) << 4));      \
        const u32x4 hi_ = *reinterpret_cast<const u32x4*>((BASE) + (ROW) * 128 + (((4 * (STEP) + 2 + fh) ^ sw_) << 4));  \
        DST = i32x8{(int)lo_[0], (int)lo_[1], (int)lo_[2], (int)lo_[3], (int)hi_[0], (int)hi_[1], (int)hi_[2], (int)hi_[3]}; \
    }
#define SA_MX_COMPUTE(CURP)                                                                                              \
    {                                                                                                                    \
        const unsigned char* cur_ = (CURP);                                                                              \
        int sxd[FM], swd[FN];                                                                                            \
        i32x8 xf[FM], wf[FN];                                                                                            \
        _Pragma("unroll") for (int i = 0; i < FM; ++i)                                                                   \
            sxd[i] = (int)(*reinterpret_cast<const uint32_t*>(cur_ + SXOFF + (wm * WTM + i * 32 + frow) * 4) >> (8 * fh)); \
        _Pragma("unroll") for (int j = 0; j < FN; ++j)                                                                   \
            swd[j] = (int)(*reinterpret_cast<const uint32_t*>(cur_ + SWOFF + (wn * WTN + j * 32 + frow) * 4) >> (8 * fh)); \
        _Pragma("unroll") for (int i = 0; i < FM; ++i) SA_MX_FRAG(xf[i], cur_, wm * WTM + i * 32 + frow, 0);             \
        _Pragma("unroll") for (int j = 0; j < FN; ++j) SA_MX_FRAG(wf[j], cur_ + XBYTES, wn * WTN + j * 32 + frow, 0);    \
        _Pragma("unroll") for (int j = 0; j < FN; ++j)                                                                   \
            _Pragma("unroll") for (int i = 0; i < FM; ++i)                                                               \
                acc[j][i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wf[j], xf[i], acc[j][i], 0, 0, 0, swd[j], 0, sxd[i]); \
        _Pragma("unroll") for (int i = 0; i < FM; ++i) SA_MX_FRAG(xf[i], cur_, wm * WTM + i * 32 + frow, 1);             \
        _Pragma("unroll") for (int j = 0; j < FN; ++j) SA_MX_FRAG(wf[j], cur_ + XBYTES, wn * WTN + j * 32 + frow, 1);    \
        _Pragma("unroll") for (int j = 0; j < FN; ++j)                                                                   \
            _Pragma("unroll") for (int i = 0; i < FM; ++i)                                                               \
                acc[j][i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wf[j], xf[i], acc[j][i], 0, 0, 2, swd[j], 2, sxd[i]); \
    }
    // STAGES-deep ring, exactly as gemm.h's GLDS ring: tiles kt .. kt + STAGES - 2 in flight or resident while tile kt is
    // multiplied; one raw s_barrier per K-tile; only tiles that exist are fetched and the wait count follows the tiles that
    // remain (round 2 re-fetched the last tile STAGES - 1 times at the tail and waited for it: see gemm.h).
    static_assert(STAGES >= 2 && STAGES <= 4, "ring depth");
#pragma unroll
    for (int st = 0; st < STAGES - 1; ++st)
        if (st <= last) SA_MX_ISSUE(st * BUF, st);
    int rd = 0, wr = (STAGES - 1) * BUF;
    for (int kt = 0; kt < nk; ++kt) {
        const int rem = last - kt;
        if (rem >= STAGES - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * LPT) : "memory");
        else if (rem == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (kt + STAGES - 1 <= last) SA_MX_ISSUE(wr, kt + STAGES - 1);
        __builtin_amdgcn_sched_barrier(0);
        SA_MX_COMPUTE(smem + rd);
        __builtin_amdgcn_sched_barrier(0);
        rd = rd + BUF == STAGES * BUF ? 0 : rd + BUF;
        wr = wr + BUF == STAGES * BUF ? 0 : wr + BUF;
    }
#undef SA_MX_COMPUTE
#undef SA_MX_FRAG
#undef SA_MX_ISSUE

    // ---- epilogue through LDS (fp32 tile, rows of 16-byte chunks XOR-swizzled as in gemm.h)
    __syncthreads();
    constexpr bool SWIGLU = (EPI == MX_EPI_SWIGLU && !SPLIT);
    constexpr int OW = SWIGLU ? BN / 2 : BN;
    constexpr int ROWB = OW * 4, CPR = ROWB / 16, XM = CPR >= 8 ? 7 : CPR - 1;
    // bias of the lane's columns in one batch of loads (see gemm.h: loaded at the point of use, each load is its own round trip)
    const bool has_bias = !SPLIT && !SWIGLU && p.bias != nullptr;
    [[maybe_unused]] uint2 bias_raw[FN][4];
    if constexpr (!SPLIT && !SWIGLU) {
        if (has_bias) {
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    bias_raw[j][g] = *reinterpret_cast<const uint2*>(p.bias + min(n0 + wn * WTN + j * 32 + g * 8 + (lane >> 5) * 4, p.N - 4));
        }
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int row = wm * WTM + i * 32 + (lane & 31);
#pragma unroll
        for (int j = 0; j < FN; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ncol = wn * WTN + j * 32 + g * 8 + (lane >> 5) * 4;
                float v0 = acc[j][i][4 * g], v1 = acc[j][i][4 * g + 1], v2 = acc[j][i][4 * g + 2], v3 = acc[j][i][4 * g + 3];
                if constexpr (!SPLIT && !SWIGLU) {
                    if (has_bias) {
                        float b[4];
                        load4(reinterpret_cast<const bf16_t*>(&bias_raw[j][g]), b);
                        v0 += b[0]; v1 += b[1]; v2 += b[2]; v3 += b[3];
                    }
                }
                if constexpr (SWIGLU) {          // weight rows interleaved (gate_j, up_j), as in the bf16 path
                    const int boff = (ncol >> 1) * 4;
                    float* dst = reinterpret_cast<float*>(smem + row * ROWB + ((((boff >> 4) ^ (row & XM)) << 4) | (boff & 15)));
                    store2(dst, silu_f(v0) * v1, silu_f(v2) * v3);
                } else {
                    const int boff = ncol * 4;
                    float* dst = reinterpret_cast<float*>(smem + row * ROWB + ((((boff >> 4) ^ (row & XM)) << 4) | (boff & 15)));
                    store4(dst, v0, v1, v2, v3);
                }
            }
        }
    }
    __syncthreads();
    if constexpr (EPI == MX_EPI_ARGMAX && !SPLIT) {
        // greedy-head partials from the staged tile (same reduction as gemm.h's EPI_ARGMAX)
        constexpr int TPR = NT / BM, SEG = CPR / TPR;
        static_assert(NT % BM == 0 && (TPR & (TPR - 1)) == 0 && TPR <= 8 && CPR % TPR == 0, "argmax epilogue split");
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
                if (n + i < p.N && v[i] > best) { best = v[i]; bi = n + i; }
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
    } else if constexpr (SWIGLU) {
        // MXFP8 output: one 32-wide block per (row, tile) at BN = 64. TPR = 4 adjacent lanes share a row, 8 values each.
        static_assert(BN == 64 && BM == 64, "SwiGLU -> MX epilogue is written for the 64x64 tile");
        const int row = tid >> 2, part = tid & 3;
        const unsigned char* rowp = smem + row * ROWB;
        const f32x4 a = *reinterpret_cast<const f32x4*>(rowp + (((2 * part) ^ (row & XM)) << 4));
        const f32x4 b = *reinterpret_cast<const f32x4*>(rowp + (((2 * part + 1) ^ (row & XM)) << 4));
        float m = fmaxf(fmaxf(fmaxf(fabsf(a[0]), fabsf(a[1])), fmaxf(fabsf(a[2]), fabsf(a[3]))),
                        fmaxf(fmaxf(fabsf(b[0]), fabsf(b[1])), fmaxf(fabsf(b[2]), fabsf(b[3]))));
        m = quad_max(m);
        const int e = mx_block_exp(m);
        const uint32_t q0 = mx_pack4(ldexpf(a[0], -e), ldexpf(a[1], -e), ldexpf(a[2], -e), ldexpf(a[3], -e));
        const uint32_t q1 = mx_pack4(ldexpf(b[0], -e), ldexpf(b[1], -e), ldexpf(b[2], -e), ldexpf(b[3], -e));
        const int mrow = m0 + row, nout = n0 / 2;                  // N / 2 % 32 == 0 is checked by the launcher
        if (mrow < p.M && nout < p.N / 2) {
            *reinterpret_cast<uint2*>(p.Q + (long)mrow * p.ldq + nout + part * 8) = make_uint2(q0, q1);
            if (part == 0) p.SQ[((long)(nout >> 7) * p.sq_rows + mrow) * 4 + ((nout >> 5) & 3)] = (uint8_t)(e + 127);
        }
        return;
    } else {
        for (int id = tid; id < BM * CPR; id += NT) {
            const int row = id / CPR, c = id % CPR;
            const int m = m0 + row, n = n0 + c * 4;
            if (m >= p.M || n >= p.N) continue;
            const u32x4 raw = *reinterpret_cast<const u32x4*>(smem + row * ROWB + ((c ^ (row & XM)) << 4));
            if constexpr (SPLIT) *reinterpret_cast<u32x4*>(p.part + ((long)ks * p.M + m) * p.N + n) = raw;
            else *reinterpret_cast<u32x4*>(p.C + (long)m * p.ldc + n) = raw;
        }
    }
}

template <int BM, int BN, int EPI, bool SPLIT, int STAGES>
static inline int launch_gemm_mx_cfg(const MxArgs& a, hipStream_t s) {
    const int tiles = cdiv(cdiv(a.N, BN) * a.splitk, 8) * 8 * cdiv(a.M, BM);
    a.bn_used = BN;
    constexpr size_t stage_bytes = (size_t)((BM + BN) * 132) * STAGES;
    constexpr size_t out_bytes = (size_t)BM * BN * 4;
    constexpr size_t lds = stage_bytes > out_bytes ? stage_bytes : out_bytes;
    static_assert(lds <= 160 * 1024, "LDS");
    auto kern = gemm_mx_kernel<BM, BN, EPI, SPLIT, STAGES>;
    static AttrOnce attr;
    attr.ensure(kern, lds);
    GemmProfiler& pf = gemm_profiler();
    const bool prof = pf.enabled && pf.n < GemmProfiler::POOL;
    if (prof) (void)hipEventRecord(pf.ev[2 * pf.n], s);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), lds, s, a);
    if (prof) {
        (void)hipEventRecord(pf.ev[2 * pf.n + 1], s);
        pf.cfg_of[pf.n] = gemm_cfg_id(BM, BN);
        pf.flops_of[pf.n] = 2.0 * a.M * a.N * a.K;
        const double in = ((double)a.M + a.N) * a.K * (1.0 + 1.0 / 32);
        pf.bytes_of[pf.n] = in + (SPLIT ? (double)a.M * a.N * 2.0          // the result once (bf16); the fp32 slabs are counted apart
                                        : EPI == MX_EPI_SWIGLU ? a.M * (a.N / 2) * (1.0 + 1.0 / 32)
                                        : EPI == MX_EPI_ARGMAX ? 16.0 * a.M * cdiv(a.N, BN) : 4.0 * a.M * a.N);
        pf.slab_of[pf.n] = SPLIT ? (double)a.splitk * a.M * a.N * 4.0 : 0.0;
        ++pf.n;
    }
    return (int)hipGetLastError();
}

static inline int mx_check(const MxArgs& a) {
    if (a.K % 128 != 0 || a.N % 4 != 0 || a.ldx % 16 != 0 || a.ldw % 16 != 0 || !a.X || !a.W || !a.SX || !a.SW || a.sx_rows < a.M ||
        a.sw_rows < a.N)
        return SA_ERR_SHAPE;
    return SA_OK;
}

// Non-split launches of the decode regime (M <= 256): 64x64 tiles, 128x128 for lm_head-sized N (the bf16 path's choice).
// LDS ring depths (4 split-K / 3 gate|up / 2 lm_head) were swept on the decode step (profiles/r02_fp8_decode.md): deeper rings
// (6, 8 stages) LOSE 20-100 us per step -- a bigger LDS footprint keeps the next kernel's workgroups from moving in beside
// the tail of this one -- and shallower ones are flat (gate|up 2 stages) or lose (split-K 2 stages: +115 us).
template <int EPI>
static inline int launch_gemm_mx(const MxArgs& a, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0) return SA_OK;
    if (int rc = mx_check(a)) return rc;
    if (a.M > 256) return SA_ERR_UNSUPPORTED;
    if constexpr (EPI == MX_EPI_SWIGLU) {
        if (a.N % 256 != 0 || !a.Q || !a.SQ || a.ldq % 8 != 0 || a.sq_rows < a.M) return SA_ERR_SHAPE;   // N / 2 = whole 128-wide K-tiles of the next GEMM
        return launch_gemm_mx_cfg<64, 64, EPI, false, 3>(a, s);
    } else {
        if (a.N >= 64 * 512) return launch_gemm_mx_cfg<128, 128, EPI, false, 2>(a, s);
        return launch_gemm_mx_cfg<64, 64, EPI, false, 3>(a, s);
    }
}

// Split-K launch: slice count from (N, K) only, K-tiles are 128 elements here (pick_splitk counts 128-BYTE tiles: same number).
static inline int launch_gemm_mx_splitk(MxArgs& a, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0) return SA_OK;
    if (int rc = mx_check(a)) return rc;
    if (!a.part || a.M > 256) return SA_ERR_SHAPE;
    a.splitk = pick_splitk(cdiv(a.N, 64), a.K / 128);
    return launch_gemm_mx_cfg<64, 64, MX_EPI_F32, true, 4>(a, s);
}

}  // namespace sa
