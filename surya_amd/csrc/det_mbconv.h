// MBConv as ONE kernel (surya/detection/model/encoderdecoder.py:174-225): expand 1x1 (+ folded BN + Hardswish) -> depthwise 3x3 (+ folded BN +
// Hardswish) -> projection 1x1 (+ folded BN, + residual). The expanded tensor (2048 / 6144 channels at the two stride-2 transitions: 1.07 + 0.81 GB
// per 16 pages, written by the expand GEMM and read back by dwproj_kernel) exists per 64-channel chunk in LDS only.
//   workgroup = one TH x TW tile of output pixels, 8 waves in two roles, one of each on every SIMD, one barrier per chunk:
//   X waves (0..3)  hold the tile's input patch ((TH - 1) S + 3) x ((TW - 1) S + 3) pixels x Cin as MFMA B fragments IN REGISTERS for the whole tile
//     (a patch row = one lane pair; RT row tiles of 32 patch pixels are dealt to the four waves as whole and half tiles); per chunk
//     E^T[64 mid][patch px] = W1_chunk . patch^T over K = Cin, W1 chunks through a two-buffer LDS ring (global_load_lds, XOR swizzle), then
//     + bias, Hardswish, round to bf16 (the expand GEMM's epilogue), ZERO where the patch pixel lies outside the image (the depthwise
//     convolution pads the EXPANDED tensor), into E[chunk & 1] in LDS as [patch px][64 ch].
//   D waves (4..7)  one chunk behind: depthwise 3x3 of the tile from E -- a lane = one output pixel x 8 channels, which is the projection's
//     B-fragment layout -- in fp32 (bias first, (ky, kx) ascending: dwconv_tx_kernel's order), Hardswish, round, into D[chunk & 1] in LDS
//     ([out px][64 ch]); and, two chunks behind, O^T[cout][px] += W2_chunk . D^T with O in accumulators (wave d owns Cout / 4 output channels;
//     its W2 fragments come straight from L2, requested at the top of the iteration).
//   K orders are the GEMMs' (16 per MFMA, ascending, one accumulator per output), rounding points the op list's: bit-identical to the three
//   launches it replaces (tests/test_gpu_det_fused.py).
#pragma once
#include "det_fused.h"

namespace sa {

#ifndef SA_MBC_ABL
#define SA_MBC_ABL 0      // timing ablations (results wrong): 1 X waves skip the MFMAs, 2 X waves skip the epilogue arithmetic, 4 D waves skip the depthwise arithmetic, 8 D waves skip the projection
#endif
#ifndef SA_MBC_DPRIO
#define SA_MBC_DPRIO 0      // 1 measured: the X waves' MFMA stream loses its issue slots to the depthwise (A 480 -> 515 us, B 677 -> 845)
#endif
#ifndef SA_MBC_DREQ
#define SA_MBC_DREQ 1
#endif
#ifndef SA_MBC_W2SPREAD
#define SA_MBC_W2SPREAD 0
#endif
#ifndef SA_MBC_TIMING
#define SA_MBC_TIMING 0   // tools/microbench/mbconv_timing.hip: s_memtime stamps of waves 0 (X) and 4 (D) per chunk iteration into `dbg`
#endif
#if SA_MBC_TIMING
#define MBC_STAMP(ROLE, IT, K) { if (lane == 0 && (IT) < 40) dbg[(((long)blockIdx.x * 2 + (ROLE)) * 40 + (IT)) * 8 + (K)] = (long long)__builtin_amdgcn_s_memtime(); }
#else
#define MBC_STAMP(ROLE, IT, K) {}
#endif

template <int CIN, int S, int TH, int TW, int COUT>
__global__ __launch_bounds__(512) void mbconv_kernel(const bf16_t* __restrict__ in, const bf16_t* __restrict__ w1, const bf16_t* __restrict__ b1,
                                                     const bf16_t* __restrict__ wd, const bf16_t* __restrict__ bd, const bf16_t* __restrict__ w2,
                                                     const bf16_t* __restrict__ b2, const bf16_t* __restrict__ res, bf16_t* __restrict__ out,
                                                     int H, int W, int Ho, int Wo, int Cm, int tiles_x, int tiles_y
#if SA_MBC_TIMING
                                                     , long long* __restrict__ dbg
#endif
                                                     ) {
    constexpr int MCH = 64, KK1 = CIN / 16;
    // the W1 ring's LDS-DMA requests are issued by the D waves at Cin = 128 and by the X waves at Cin = 256, where the D waves already carry 16 W2
    // fragment requests per chunk and the X waves have the slack (one gpurun A/B of all four combinations: 403 / 522 us with D, 406 / 513 with X)
    constexpr bool DREQ = SA_MBC_DREQ && CIN == 128;
    constexpr int PH = (TH - 1) * S + 3, PW = (TW - 1) * S + 3, NROW = PH * PW, RT = (NROW + 31) / 32;
    constexpr int NPX = TH * TW, PT = NPX / 32, NJ = COUT / 128;       // output pixel tiles; cout tiles per D wave
    static_assert(NPX % 32 == 0 && (TW & (TW - 1)) == 0 && COUT % 128 == 0 && CIN % 64 == 0, "tile shape");
    static_assert(RT == 10 || RT == 5, "X-wave deal below is written for 10 (2 + 2 + half) and 5 (1 + half / 1) row tiles");
    constexpr int NF = RT == 10 ? 2 : 1;                                // whole row tiles per X wave
    constexpr int EB = RT * 32 * 128;                                   // one E buffer [RT * 32 patch px][64 ch] bf16
    constexpr int RB = MCH * CIN * 2;                                   // one W1 ring buffer [64 mid][CIN]
    constexpr int DB = NPX * 128;                                       // one D buffer [NPX][64 ch]
    constexpr int OFF_RING = 2 * EB, OFF_D = OFF_RING + 2 * RB, OFF_TAP = OFF_D + 2 * DB, TAPB = 2560, OFF_B1 = OFF_TAP + 2 * TAPB;   // then the expand bias, all Cm of it
    constexpr int CPR1 = CIN / 8, RPI = 64 / CPR1, NI = RB / 1024 / 4;  // W1 ring: 16-byte chunks per row, rows per request, requests per X wave
    static_assert(NPX * COUT * 2 <= 2 * EB, "output tile overlays the E buffers");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 31, lh = lane >> 5;
    // XCD x takes the x-th contiguous eighth of the tile raster (tiles that share patch rows meet in one L2)
    const unsigned nb = gridDim.x, xcd = blockIdx.x & 7, per = nb >> 3, rem = nb & 7;
    const int bid = (int)(xcd * per + min(xcd, rem) + (blockIdx.x >> 3));
    const int b = bid / (tiles_x * tiles_y), tr = bid - b * tiles_x * tiles_y;
    const int oy0 = (tr / tiles_x) * TH, ox0 = (tr % tiles_x) * TW;
    const int nch = Cm / MCH;
    for (int i = tid; i < Cm / 8; i += 512) *reinterpret_cast<uint4*>(smem + OFF_B1 + i * 16) = *reinterpret_cast<const uint4*>(b1 + i * 8);   // visible behind (P0)

    // W1 ring requests (issued by the D waves, SA_MBC_DREQ: an LDS-DMA request costs 60-185 cycles of issue inside the X waves' MFMA stream, and the D
    // waves are the ones with slack): request q = (wv & 3) * NI + i fills rows [q * RPI, + RPI) of the chunk; LDS slot (row, physical chunk pc) <- global chunk pc ^ (row & 15)
    unsigned wq[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int row = ((wv & 3) * NI + i) * RPI + lane / CPR1, pc = lane % CPR1;
        wq[i] = (unsigned)(row * CIN * 2 + ((pc ^ (row & 15)) << 4));
    }
#define MB_ISSUE(BUF, CH)                                                                                               \
    {                                                                                                                   \
    const unsigned char* b_ = reinterpret_cast<const unsigned char*>(w1) + (long)(CH) * MCH * CIN * 2;              \
    _Pragma("unroll") for (int i_ = 0; i_ < NI; ++i_)                                                               \
        __builtin_amdgcn_global_load_lds((gptr_t)(b_ + wq[i_]), (lptr_t)(smem + OFF_RING + (BUF) * RB + ((wv & 3) * NI + i_) * 1024), 16, 0, 0); \
    }
    if (wv < 4) {
        // ------------------------------------------------------------------ X waves: expand
        int ft[NF], hf, hct;
        if constexpr (RT == 10) {
            const int g = wv >> 1, o = wv & 1;
            ft[0] = g * 5 + o * 3; ft[1] = ft[0] + 1; hf = g * 5 + 2; hct = o;
        } else {
            ft[0] = wv == 0 ? 0 : wv + 1; hf = wv < 2 ? 1 : -1; hct = wv & 1;
        }
        const bool has_half = hf >= 0;                       // wave-uniform
        if constexpr (!DREQ) MB_ISSUE(0, 0);
        // the patch: B fragments of this wave's row tiles, straight from global memory (clamped addresses; pixels outside the image are zeroed at the E store)
        u32x4 pf[NF + 1][KK1];
        unsigned emask[NF + 1];
        int erow[NF + 1];
#pragma unroll
        for (int t = 0; t <= NF; ++t) {
            const int rt = t < NF ? ft[t] : max(hf, 0);
            const int idx = rt * 32 + lr, py = idx / PW, px = idx - py * PW;
            const int iy = oy0 * S - 1 + py, ix = ox0 * S - 1 + px;
            const bool valid = idx < NROW && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            emask[t] = valid ? 0xffffffffu : 0u;
            erow[t] = idx * 128 + lh * 8;
            const bf16_t* src = in + (((long)b * H + min(max(iy, 0), H - 1)) * W + min(max(ix, 0), W - 1)) * CIN + lh * 8;
#pragma unroll
            for (int kk = 0; kk < KK1; ++kk) pf[t][kk] = *reinterpret_cast<const u32x4*>(src + kk * 16);
        }
        // read side of the ring: A fragment (ct, kk) of lane (lr, lh) = row ct * 32 + lr, chunk (kk * 2 + lh) ^ (row & 15)
        const int wrow0 = lr * CIN * 2, wrow1 = (32 + lr) * CIN * 2, wrowh = (hct * 32 + lr) * CIN * 2, wkey = lr & 15;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                     // (P0) W1 chunk 0 landed
        for (int it = 0; it < nch + 2; ++it) {
            if (wv == 0) MBC_STAMP(0, it, 0);
            if (it < nch) {
                const int buf = it & 1;
                if constexpr (!DREQ) MB_ISSUE(buf ^ 1, min(it + 1, nch - 1));     // unconditional (clamped): the buffer every X wave left at the last barrier
                f32x16 acc[2 * NF + 1];
#pragma unroll
                for (int u = 0; u < 2 * NF + 1; ++u)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
                const unsigned char* ring = smem + OFF_RING + buf * RB;
                // W1 fragments of K step kk + PD are read while step kk multiplies: PD + 1 named register sets, one LDS read placed behind each of the
                // first MFMAs of a step (left to hipcc: read x 3, s_waitcnt lgkmcnt(0), MFMA x 5 -- every LDS round trip exposed, 39 waits per chunk).
                // PD = 2 where a step is only three MFMAs (RT = 5: 96 cycles of cover against an LDS round trip of 130 and more: the loop ran at 2.4x its
                // MFMA time with PD = 1). A wave without a half tile (RT = 5: waves 2, 3) multiplies its third accumulator anyway -- never stored.
#define MB_R(WF, KK)                                                                                                    \
    {                                                                                                                   \
        const int co_ = (((KK) * 2 + lh) ^ wkey) << 4;                                                                  \
        WF[0] = *reinterpret_cast<const u32x4*>(ring + wrow0 + co_);                                                    \
        WF[1] = *reinterpret_cast<const u32x4*>(ring + wrow1 + co_);                                                    \
        WF[2] = *reinterpret_cast<const u32x4*>(ring + wrowh + co_);                                                    \
    }
#define MB_M(WF, KK)                                                                                                    \
    {                                                                                                                   \
        _Pragma("unroll") for (int i_ = 0; i_ < NF; ++i_) {                                                             \
            acc[2 * i_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, WF[0]), __builtin_bit_cast(bf16x8, pf[i_][KK]), acc[2 * i_], 0, 0, 0); \
            acc[2 * i_ + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, WF[1]), __builtin_bit_cast(bf16x8, pf[i_][KK]), acc[2 * i_ + 1], 0, 0, 0); \
        }                                                                                                               \
        acc[2 * NF] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, WF[2]), __builtin_bit_cast(bf16x8, pf[NF][KK]), acc[2 * NF], 0, 0, 0); \
    }
#define MB_SGB()                                                                                                        \
    {                                                                                                                   \
        _Pragma("unroll") for (int q_ = 0; q_ < 3; ++q_) {                                                              \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                          \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                          \
        }                                                                                                               \
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * NF + 1 - 3, 0);                                                 \
    }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (!(SA_MBC_ABL & 1)) {
                    constexpr int PD = NF == 1 ? 2 : 1, NS = PD + 1;
                    u32x4 wr[NS][3];
#pragma unroll
                    for (int kk = 0; kk < PD; ++kk) MB_R(wr[kk], kk);
#pragma unroll
                    for (int kk = 0; kk < KK1; ++kk) {
                        MB_M(wr[kk % NS], kk);
                        if (kk + PD < KK1) MB_R(wr[(kk + PD) % NS], kk + PD);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x100, 3 * PD, 0);
#pragma unroll
                    for (int kk = 0; kk < KK1 - PD; ++kk) MB_SGB();
                    __builtin_amdgcn_sched_group_barrier(0x008, (2 * NF + 1) * PD, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
#undef MB_R
#undef MB_M
#undef MB_SGB
                if (wv == 0) MBC_STAMP(0, it, 1);
                // the chunk's expand bias (all of b1 sits in LDS): one batch of reads under the draining MFMAs
                uint2 b1r[2][4];
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int g = 0; g < 4; ++g) b1r[ct][g] = *reinterpret_cast<const uint2*>(smem + OFF_B1 + (it * MCH + ct * 32 + g * 8 + lh * 4) * 2);
                __builtin_amdgcn_sched_barrier(0);
                // ---- + bias, Hardswish, bf16, zero outside the image -> E[buf] (accumulator layout: lane = patch pixel lr, quad g = channels 8 g + 4 lh + (0..3))
                unsigned char* eb = smem + buf * EB;
#pragma unroll
                for (int u = 0; u < 2 * NF + 1; ++u) {
                    const int t = u < 2 * NF ? u / 2 : NF;
                    if (u == 2 * NF && !has_half) break;
                    const int ct = u < 2 * NF ? (u & 1) : hct;
                    const int row = erow[t] >> 7;
                    const int key = (row >> 1) & 7;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float bq[4];
                        const uint2 braw = u < 2 * NF ? b1r[u & 1][g] : (hct ? b1r[1][g] : b1r[0][g]);
                        load4(reinterpret_cast<const bf16_t*>(&braw), bq);
                        uint32_t pk[2];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            f32x2 x = f32x2{acc[u][4 * g + 2 * h], acc[u][4 * g + 2 * h + 1]};
                            if constexpr (!(SA_MBC_ABL & 2)) {
                                x = hardswish_pk(x + f32x2{bq[2 * h], bq[2 * h + 1]});      // common.h: the op list's Hardswish on an fp32 pair
                            }
                            pk[h] = __builtin_bit_cast(uint32_t, __builtin_convertvector(x, bf16x2_t)) & emask[t];
                        }
                        *reinterpret_cast<uint2*>(eb + (erow[t] & ~127) + (((ct * 4 + g) ^ key) << 4) + lh * 8) = make_uint2(pk[0], pk[1]);
                    }
                }
                if (wv == 0) MBC_STAMP(0, it, 2);
                if constexpr (!DREQ) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // W1 chunk it + 1 landed
            }
            if (wv == 0) MBC_STAMP(0, it, 3);
            __syncthreads();                                 // (B_it)
        }
    } else {
        // ------------------------------------------------------------------ D waves: depthwise, one chunk behind; projection, two behind
        const int d = wv - 4, pt_ = tid - 256;
#if SA_MBC_DPRIO
        __builtin_amdgcn_s_setprio(SA_MBC_DPRIO);            // the second-dispatched half loses every VALU arbitration against the X waves' epilogue otherwise
#endif
        const int cg = d * 2 + lh;                           // this lane's 16-byte channel group of a chunk: channels [cg * 8, + 8) = K step d of the projection
        int eoff[PT][9], doff[PT];
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            const int q = p * 32 + lr, oyl = q / TW, oxl = q % TW;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int row = (oyl * S + ky) * PW + oxl * S + kx;
                    eoff[p][ky * 3 + kx] = row * 128 + ((cg ^ ((row >> 1) & 7)) << 4);
                }
            doff[p] = q * 128 + ((cg ^ ((q >> 1) & 7)) << 4);
        }
        // tap ring: thread pt_ < 80 carries 16 bytes of (tap pt_ >> 3 | bias = row 9), channels (pt_ & 7) * 8 of the chunk, and stores them as fp32: the
        // depthwise below reads its operands ready-made (broadcast reads), no tap unpacking on the vector ALU. Slot: [10 rows][8 groups][8 floats] = 2560 bytes.
        // (v_dot2c_f32_bf16 against taps stored as (lo, 0) / (0, hi) word pairs needs no unpacking of the inputs either -- 8 instead of 12 instructions per
        // vector -- and was measured: 13 us faster per launch, but it is NOT the fp32 fma: pages 7 and 15 of the 16-page bench input left the op list's bits
        // (9e-3 on the maps) while 2- and 4-page inputs stayed identical. Kept exact.)
        const bool wl = pt_ < 80;
        const bf16_t* wsrc = (pt_ >> 3) < 9 ? wd + (long)(pt_ >> 3) * Cm + (pt_ & 7) * 8 : bd + (pt_ & 7) * 8;
        u32x4 wld = {0u, 0u, 0u, 0u};
        if constexpr (DREQ) MB_ISSUE(0, 0);
        if (wl) wld = *reinterpret_cast<const u32x4*>(wsrc);
        f32x16 acc[NJ][PT];
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int p = 0; p < PT; ++p)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][p][r] = 0.f;
        // W2 comes FRAGMENT-MAJOR (mbconv_w2_fragments_kernel below, once per engine): [chunk][cout tile][K step][lane][8], so a fragment load is 1 KiB
        // contiguous = 8 cache lines. From the row-major [Cout][Cm] weight a load touched 32 lines (one per output channel, 32 bytes used of each): 1024 /
        // 2048 line requests per chunk and CU, and the D waves' iteration was bound by them (4449 / 5544 cycles against ~1700 of arithmetic).
        const bf16_t* w2p = w2 + ((long)(d * NJ) * 4 * 64 + lane) * 8;
        if constexpr (DREQ) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                     // (P0)
        for (int it = 0; it < nch + 2; ++it) {
            if (wv == 4) MBC_STAMP(1, it, 0);
            u32x4 w2f[NJ][4];
            {
                // depthwise of chunk it - 1 (runs on stale buffers at it = 0 and it = nch + 1: results land in the D buffer nobody reads)
                const int c1 = it - 1;
                const unsigned char* eb = smem + (c1 & 1) * EB;
                const unsigned char* ws = smem + OFF_TAP + (c1 & 1) * TAPB + cg * 32;
                // taps (32 bytes, broadcast reads) and input vectors (PT x 16 bytes) travel through a five-set register ring: the first four taps are
                // requested FIRST, then this iteration's global requests are issued (13 / 25 vector-memory instructions: 1500 / 2100 cycles of issue -- the
                // LDS round trips hide under them), then tap t + 4 is requested while tap t is summed. (A three-set ring behind the requests: every step
                // waited on LDS, 2064 cycles for 144 v_dot2c; a row-deep double buffer held 96 registers and spilled, and a scratch reload in this loop
                // waits vmcnt(0), i.e. for the W2 fragments: 5400 of a 6600-cycle iteration.)
                constexpr int NS = 5;
                u32x4 tw[NS][2], ex[NS][PT];
#define MB_LE(SET, T)                                                                                                   \
    {                                                                                                                   \
        tw[SET][0] = *reinterpret_cast<const u32x4*>(ws + (T) * 256);                                                   \
        tw[SET][1] = *reinterpret_cast<const u32x4*>(ws + (T) * 256 + 16);                                              \
        _Pragma("unroll") for (int p_ = 0; p_ < PT; ++p_) ex[SET][p_] = *reinterpret_cast<const u32x4*>(eb + eoff[p_][T]); \
    }
                const f32x4 bl = *reinterpret_cast<const f32x4*>(ws + 9 * 256), bh = *reinterpret_cast<const f32x4*>(ws + 9 * 256 + 16);
#pragma unroll
                for (int t = 0; t < NS - 1; ++t) MB_LE(t, t);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (DREQ) { if (it < nch) MB_ISSUE((it & 1) ^ 1, min(it + 1, nch - 1)); }     // the ring buffer every X wave left at the last barrier
                // taps of chunk `it` into slot it & 1 (read by the depthwise of the NEXT iteration); chunk it + 1's requested
                if (wl) {
                    unsigned char* dst = smem + OFF_TAP + (it & 1) * TAPB + pt_ * 32;
                    *reinterpret_cast<u32x4*>(dst) = u32x4{wld[0] << 16, wld[0] & 0xffff0000u, wld[1] << 16, wld[1] & 0xffff0000u};
                    *reinterpret_cast<u32x4*>(dst + 16) = u32x4{wld[2] << 16, wld[2] & 0xffff0000u, wld[3] << 16, wld[3] & 0xffff0000u};
                    wld = *reinterpret_cast<const u32x4*>(wsrc + min(it + 1, nch - 1) * MCH);
                }
                // W2 fragments of chunk it - 2 (used behind the depthwise arithmetic below)
                const long c2 = (long)min(max(it - 2, 0), nch - 1) * (COUT / 32) * 4 * 64 * 8;
                if constexpr (!SA_MBC_W2SPREAD) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) w2f[j][kk] = *reinterpret_cast<const u32x4*>(w2p + c2 + (j * 4 + kk) * 64 * 8);
                }
                if (wv == 4) MBC_STAMP(1, it, 4);
                __builtin_amdgcn_sched_barrier(0);
                f32x2 a2[PT][4];
#pragma unroll
                for (int p = 0; p < PT; ++p) { a2[p][0] = f32x2{bl[0], bl[1]}; a2[p][1] = f32x2{bl[2], bl[3]}; a2[p][2] = f32x2{bh[0], bh[1]}; a2[p][3] = f32x2{bh[2], bh[3]}; }
                if (wv == 4) MBC_STAMP(1, it, 5);
#pragma unroll
                for (int t = 0; t < 9; ++t) {                // (ky, kx) ascending: dwconv_tx_kernel's order
                    if (t + NS - 1 < 9) MB_LE((t + NS - 1) % NS, t + NS - 1);
                    if constexpr (SA_MBC_W2SPREAD) {             // the W2 requests ride between the taps: NJ * 4 of them over the first eight steps
#pragma unroll
                        for (int q = t * (NJ / 2); q < (t + 1) * (NJ / 2) && t < 8; ++q)
#pragma unroll
                            for (int h = 0; h < 1; ++h) w2f[q / 4][q % 4] = *reinterpret_cast<const u32x4*>(w2p + c2 + q * 64 * 8);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (!(SA_MBC_ABL & 4)) {
                        const f32x4 t0 = __builtin_bit_cast(f32x4, tw[t % NS][0]), t1 = __builtin_bit_cast(f32x4, tw[t % NS][1]);
                        const f32x2 w4[4] = {f32x2{t0[0], t0[1]}, f32x2{t0[2], t0[3]}, f32x2{t1[0], t1[1]}, f32x2{t1[2], t1[3]}};
#pragma unroll
                        for (int p = 0; p < PT; ++p) {
                            f32x2 x[4];
                            const u32x4 xr = ex[t % NS][p];
                            UpsumPk<bf16_t>::unpack(make_uint4(xr[0], xr[1], xr[2], xr[3]), x);
#pragma unroll
                            for (int e = 0; e < 4; ++e) a2[p][e] += x[e] * w4[e];       // dwproj_kernel's expression: the fp32 fma of the op list
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#undef MB_LE
                if (wv == 4) MBC_STAMP(1, it, 6);
#pragma unroll
                for (int p = 0; p < PT; ++p) {
                    float r[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { r[2 * e] = hardswish_f(a2[p][e].x); r[2 * e + 1] = hardswish_f(a2[p][e].y); }
                    *reinterpret_cast<uint4*>(smem + OFF_D + (c1 & 1) * DB + doff[p]) =
                        make_uint4(pack2(r[0], r[1]), pack2(r[2], r[3]), pack2(r[4], r[5]), pack2(r[6], r[7]));
                }
            }
            if (wv == 4) MBC_STAMP(1, it, 1);
            if (it >= 2) {
                const unsigned char* db = smem + OFF_D + (it & 1) * DB;      // chunk it - 2
                u32x4 xf[4][PT];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int p = 0; p < PT; ++p) {
                        const int q = p * 32 + lr;
                        xf[kk][p] = *reinterpret_cast<const u32x4*>(db + q * 128 + (((kk * 2 + lh) ^ ((q >> 1) & 7)) << 4));
                    }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    if constexpr (SA_MBC_ABL & 8) { asm volatile("" ::"v"(xf[kk][0]), "v"(w2f[0][kk])); }
                    else {
#pragma unroll
                        for (int j = 0; j < NJ; ++j)
#pragma unroll
                            for (int p = 0; p < PT; ++p)
                                acc[j][p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w2f[j][kk]), __builtin_bit_cast(bf16x8, xf[kk][p]), acc[j][p], 0, 0, 0);
                    }
                }
            }
            // an unconditional use of the W2 fragments: with their only use inside `if (it >= 2)` LLVM sinks the loads into that block -- issued right in
            // front of the MFMAs, a full L2 round trip exposed per chunk (found in the ISA: global_load x 8, s_waitcnt vmcnt(7), v_mfma)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) asm volatile("" ::"v"(w2f[j][kk]));
            if (wv == 4) MBC_STAMP(1, it, 2);
            if constexpr (DREQ) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // W1 chunk it + 1 landed (the W2 fragments were waited for above)
            __syncthreads();                                 // (B_it)
            if (wv == 4) MBC_STAMP(1, it, 3);
        }
#undef MB_ISSUE
        // ---- epilogue: + bias, round (the projection GEMM's rounding) into the output tile [NPX][COUT] in LDS (over the E buffers: every wave is past B_last)
        constexpr int ROWB = COUT * 2;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            float bq[4][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) load4(b2 + (d * NJ + j) * 32 + q * 8 + lh * 4, bq[q]);
#pragma unroll
            for (int p = 0; p < PT; ++p) {
                const int row = p * 32 + lr;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int chunk = ((d * NJ + j) * 32 + q * 8) >> 3;
                    *reinterpret_cast<uint2*>(smem + row * ROWB + ((chunk ^ (row & 31)) << 4) + lh * 8) =
                        make_uint2(pack2(acc[j][p][4 * q] + bq[q][0], acc[j][p][4 * q + 1] + bq[q][1]),
                                   pack2(acc[j][p][4 * q + 2] + bq[q][2], acc[j][p][4 * q + 3] + bq[q][3]));
                }
            }
        }
    }
    __syncthreads();                                         // (E)
    {
        constexpr int ROWB = COUT * 2, CPR = ROWB / 16;
        for (int id = tid; id < NPX * CPR; id += 512) {
            const int row = id / CPR, cc = id % CPR;
            const int oy = oy0 + row / TW, ox = ox0 + row % TW;
            if (oy >= Ho || ox >= Wo) continue;
            const uint4 rawo = *reinterpret_cast<const uint4*>(smem + row * ROWB + ((cc ^ (row & 31)) << 4));
            const long off = (((long)b * Ho + oy) * Wo + ox) * COUT + cc * 8;
            if (res) {
                float a[8], r8[8];
                unpack16(rawo, a, (bf16_t*)nullptr);
                unpack16(*reinterpret_cast<const uint4*>(res + off), r8, (bf16_t*)nullptr);
                store4(out + off, a[0] + r8[0], a[1] + r8[1], a[2] + r8[2], a[3] + r8[3]);
                store4(out + off + 4, a[4] + r8[4], a[5] + r8[5], a[6] + r8[6], a[7] + r8[7]);
            } else {
                *reinterpret_cast<uint4*>(out + off) = rawo;
            }
        }
    }
}

// W2 [Cout][Cm] row-major -> [Cm / 64][Cout / 32][4][64 lanes][8]: lane (lr, lh) of fragment (chunk c, cout tile jt, K step kk) holds
// W2[jt * 32 + lr][c * 64 + kk * 16 + lh * 8 + (0..7)] -- the MFMA A operand of the projection as one contiguous KiB.
__global__ __launch_bounds__(256) void mbconv_w2_fragments_kernel(const bf16_t* __restrict__ w2, bf16_t* __restrict__ w2f, int Cout, int Cm) {
    const long n = (long)Cout * Cm / 8, i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int lane = (int)(i & 63), kk = (int)((i >> 6) & 3);
    const long r = i >> 8;
    const int jt = (int)(r % (Cout / 32)), c = (int)(r / (Cout / 32));
    *reinterpret_cast<uint4*>(w2f + i * 8) = *reinterpret_cast<const uint4*>(w2 + (long)(jt * 32 + (lane & 31)) * Cm + c * 64 + kk * 16 + (lane >> 5) * 8);
}
static inline int mbconv_w2_fragments(const bf16_t* w2, bf16_t* w2f, int Cout, int Cm, hipStream_t s) {
    if (Cout % 32 || Cm % 64) return SA_ERR_SHAPE;
    const long n = (long)Cout * Cm / 8;
    hipLaunchKernelGGL(mbconv_w2_fragments_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w2, w2f, Cout, Cm);
    return (int)hipGetLastError();
}

static inline bool mbconv_shape_ok(int cin, int mid, int cout, int stride) {
    return stride == 2 && mid % 64 == 0 && mid >= 128 && ((cin == 128 && cout == 256) || (cin == 256 && cout == 512));
}

#if SA_MBC_TIMING
static long long* g_mbc_dbg = nullptr;
#define MBC_DBG_ARG , g_mbc_dbg
#else
#define MBC_DBG_ARG
#endif
// w2f: the projection weight fragment-major (mbconv_w2_fragments)
static inline int launch_mbconv(const bf16_t* in, const bf16_t* w1, const bf16_t* b1, const bf16_t* wd, const bf16_t* bd, const bf16_t* w2,
                                const bf16_t* b2, const bf16_t* res, bf16_t* out, int B, int H, int W, int Cin, int Cm, int Ho, int Wo, int Cout,
                                int stride, hipStream_t s) {
    if (!mbconv_shape_ok(Cin, Cm, Cout, stride) || !b1 || !bd || !b2 || (long)B * H * W * Cin >= (1L << 31) || Cm > 8192) return SA_ERR_SHAPE;
#define SA_MBC(CI, SS, TH_, TW_, CO)                                                                                            \
    {                                                                                                                           \
        constexpr int PH_ = ((TH_) - 1) * (SS) + 3, PW_ = ((TW_) - 1) * (SS) + 3, RT_ = (PH_ * PW_ + 31) / 32;                  \
        const size_t lds = (size_t)2 * RT_ * 32 * 128 + 2 * 64 * (CI) * 2 + 2 * (TH_) * (TW_) * 128 + 2 * 2560 + (size_t)Cm * 2;             \
        const int tx = cdiv(Wo, TW_), ty = cdiv(Ho, TH_);                                                                       \
        auto kern = mbconv_kernel<CI, SS, TH_, TW_, CO>;                                                                        \
        static AttrOnce attr;                                                                                                   \
        attr.ensure(kern, lds);                                                                                                 \
        hipLaunchKernelGGL(kern, dim3((unsigned)(B * tx * ty)), dim3(512), lds, s, in, w1, b1, wd, bd, w2, b2, res, out, H, W, Ho, Wo, Cm, tx, ty MBC_DBG_ARG); \
    }
    if (Cin == 128) SA_MBC(128, 2, 8, 8, 256) else SA_MBC(256, 2, 4, 8, 512)
#undef SA_MBC
    return (int)hipGetLastError();
}
template <typename T>
static inline int launch_mbconv(const T*, const T*, const T*, const T*, const T*, const T*, const T*, const T*, T*, int, int, int, int, int, int, int, int,
                                int, hipStream_t) { return SA_ERR_UNSUPPORTED; }

}  // namespace sa
