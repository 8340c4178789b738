// Fused forms of the text detector's blocks (round 6). The op-by-op kernels of det_kernels.h moved every intermediate through HBM:
// 29 GB per 16 pages of 1024^2 against ~3.5 GB of block-boundary tensors, and the LiteMLA chain spent 2 ms of a 10.5 ms
// forward on 0.3 % of its FLOPs. Each kernel here keeps one intermediate on the chip:
//   litemla_fused_kernel      kv = relu(K)^T [V | 1] and out = relu(Q) kv / (den + eps) per (image, head), both on the fp32 MFMA
//                             (surya/detection/model/encoderdecoder.py:332-359; fp32 like the reference's _attn)
//   dw5_g1x1_kernel           LiteMLA's multi-scale branch: depthwise 5x5 -> (LDS) -> grouped 1x1 on the bf16 MFMA (:313-318)
//   head_z0_kernel            the full-resolution stage's folded 1x1 convolution z0 = A0 x0 + c computed per 128-channel slab on the MFMA inside the
//                             decode head's sum + ReLU + classify + sigmoid pass (:699-722 in the folded form of detection/plan.py)
//   dwproj_kernel             MBConv's depthwise 3x3 (+ bias + Hardswish) as the producer of the projection GEMM's A tile (:174-225)
// All are bf16 product-path kernels (fp32 reference mode keeps the op list); litemla_fused_kernel is templated on the storage type.
// Rounding points are the op list's own: every tensor the reference materialises in the model dtype is rounded to bf16 at the same place.
#pragma once
#include "det_kernels.h"

namespace sa {

// ---------------------------------------------------------------------------------------------------
// LiteMLA, one workgroup per (head, image). v_mfma_f32_32x32x2_f32 is an exact fp32 fma chain over the contraction index (here: tokens
// for kv, the 32 head channels for out), so this is the reference's fp32 arithmetic in a fixed, batch-independent order.
//   phase A  per 128-token chunk: relu(k), v staged as fp32 in LDS; wave w reduces tokens [32 w, 32 w + 32) of the chunk:
//            D[i][j] += relu(k)[t][i] * v[t][j] (A = k column i, B = v column j, contraction over t, two tokens per MFMA); the ones-column
//            of v (row sums of relu(k)) is a running add of the A operand. The four waves' partial 32 x 32 matrices meet in LDS.
//   phase B  per chunk: relu(q) staged [token][33] (padded: the B operand walks tokens across lanes); D[j][t] = sum_i kv[i][j] * q[t][i]
//            (A = kv row i as 16 hoisted registers); the divisor den[t] = sum_i q[t][i] * ksum[i] rides on the same operand loads.
template <typename T, int DIM>
__global__ __launch_bounds__(256) void litemla_fused_kernel(const T* __restrict__ qa, const T* __restrict__ qb, T* __restrict__ out, int HW,
                                                            int heads_a, int heads, float eps) {
    static_assert(DIM == 32, "one 32 x 32 fp32 MFMA tile per head");
    constexpr int CH = 128, V = Ty<T>::V16;
    __shared__ __attribute__((aligned(16))) float sm[CH * 64 + 4 * 1024 + 1024 + 160];
    float* ks = sm;                      // [CH][32] relu(k)        (phase A)
    float* vs = sm + CH * 32;            // [CH][32] v              (phase A)
    float* qs = sm;                      // [CH][33] relu(q)        (phase B, aliases ks / vs)
    float* part = sm + CH * 64;          // [4][32][32] per-wave partial kv
    float* kv = part + 4096;             // [32][32]
    float* ksp = kv + 1024;              // [4][32] per-wave partial column sums of relu(k)
    float* ksf = ksp + 128;              // [32]
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int Ca = heads_a * 3 * DIM, Cb = (heads - heads_a) * 3 * DIM;
    const T* src = h < heads_a ? qa + (long)b * HW * Ca + h * 3 * DIM : qb + (long)b * HW * Cb + (h - heads_a) * 3 * DIM;
    const int C = h < heads_a ? Ca : Cb;
    const int lr = lane & 31, lh = lane >> 5;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float ksum = 0.f;
    // k | v of a token are 2 * DIM contiguous elements behind q: CPT 16-byte pieces per token, NLD of them per thread and chunk. The pieces of
    // chunk n + 1 are requested before the MFMAs of chunk n (first version: load -> barrier -> multiply, one exposed memory round trip per chunk,
    // 52 us per launch of 16 chunk steps)
    constexpr int CPT = 2 * DIM / V, NLD = CH * CPT / 256, QPT = DIM / V, NLQ = CH * QPT / 256;
    static_assert(CH * CPT % 256 == 0 && CH * QPT % 256 == 0, "staging split");
    uint4 pre[NLD];
#define ML_LOADKV(N0)                                                                                                   \
    {                                                                                                                   \
        _Pragma("unroll") for (int i_ = 0; i_ < NLD; ++i_) {                                                            \
            const int idx_ = tid + i_ * 256, n_ = idx_ / CPT, pc_ = idx_ % CPT, tok_ = min((N0) + n_, HW - 1);          \
            pre[i_] = *reinterpret_cast<const uint4*>(src + (long)tok_ * C + DIM + pc_ * V);                            \
        }                                                                                                               \
    }
    ML_LOADKV(0);
    for (int n0 = 0; n0 < HW; n0 += CH) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = tid + i * 256, n = idx / CPT, pc = idx % CPT;
            float x[V];
            unpack16(pre[i], x, (T*)nullptr);
            const bool live = n0 + n < HW;                   // a token past the end contributes relu(k) = 0 (its clamped load is discarded)
            const int e0 = pc * V;
            if (e0 < DIM) {
#pragma unroll
                for (int e = 0; e < V; ++e) ks[n * 32 + e0 + e] = live ? fmaxf(x[e], 0.f) : 0.f;
            } else {
#pragma unroll
                for (int e = 0; e < V; ++e) vs[n * 32 + e0 - DIM + e] = live ? x[e] : 0.f;
            }
        }
        __syncthreads();
        ML_LOADKV(min(n0 + CH, HW - 1));                     // unconditional (clamped): a branch around the loads drains them at the merge
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int t = w * 32 + 2 * s + lh;
            const float a = ks[t * 32 + lr], bv = vs[t * 32 + lr];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc, 0, 0, 0);
            ksum += a;
        }
        __syncthreads();
    }
#undef ML_LOADKV
    ksum += __shfl_xor(ksum, 32, 64);
#pragma unroll
    for (int r = 0; r < 16; ++r) part[w * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + lr] = acc[r];
    if (lane < 32) ksp[w * 32 + lane] = ksum;
    __syncthreads();
    for (int e = tid; e < 1024; e += 256) kv[e] = ((part[e] + part[1024 + e]) + part[2048 + e]) + part[3072 + e];      // fixed order
    if (tid < 32) ksf[tid] = ((ksp[tid] + ksp[32 + tid]) + ksp[64 + tid]) + ksp[96 + tid];
    __syncthreads();

    float ak[16], ksl[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) { ak[s] = kv[(2 * s + lh) * 32 + lr]; ksl[s] = ksf[2 * s + lh]; }
    const int Cout = heads * DIM;
    uint4 preq[NLQ];
#define ML_LOADQ(N0)                                                                                                    \
    {                                                                                                                   \
        _Pragma("unroll") for (int i_ = 0; i_ < NLQ; ++i_) {                                                            \
            const int idx_ = tid + i_ * 256, n_ = idx_ / QPT, pc_ = idx_ % QPT, tok_ = min((N0) + n_, HW - 1);          \
            preq[i_] = *reinterpret_cast<const uint4*>(src + (long)tok_ * C + pc_ * V);                                 \
        }                                                                                                               \
    }
    ML_LOADQ(0);
    for (int n0 = 0; n0 < HW; n0 += CH) {
#pragma unroll
        for (int i = 0; i < NLQ; ++i) {
            const int idx = tid + i * 256, n = idx / QPT, pc = idx % QPT;
            float x[V];
            unpack16(preq[i], x, (T*)nullptr);
#pragma unroll
            for (int e = 0; e < V; ++e) qs[n * 33 + pc * V + e] = fmaxf(x[e], 0.f);
        }
        __syncthreads();
        ML_LOADQ(min(n0 + CH, HW - 1));
        f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = 0.f;
        float den = 0.f;
        const int tl = w * 32 + lr;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float bq = qs[tl * 33 + 2 * s + lh];
            o = __builtin_amdgcn_mfma_f32_32x32x2f32(ak[s], bq, o, 0, 0, 0);
            den += bq * ksl[s];
        }
        den += __shfl_xor(den, 32, 64);
        const float inv = 1.0f / (den + eps);
        const int tok = n0 + tl;
        if (tok < HW) {
            T* dst = out + ((long)b * HW + tok) * Cout + h * DIM;
#pragma unroll
            for (int g = 0; g < 4; ++g) store4(dst + 8 * g + 4 * lh, o[4 * g] * inv, o[4 * g + 1] * inv, o[4 * g + 2] * inv, o[4 * g + 3] * inv);
        }
        __syncthreads();
    }
#undef ML_LOADQ
}

// ---------------------------------------------------------------------------------------------------
// LiteMLA multi-scale branch (encoderdecoder.py:313-318): depthwise 5x5 (no bias) then a grouped 1x1 with 32-channel groups (no bias).
// Workgroup = one 8 x 32 pixel tile of one image x one 32-channel group. The group's input patch (12 x 36 pixels, zero outside the image)
// and its 25 x 32 filter taps sit in LDS; a thread produces 4 vertically adjacent pixels x 8 channels of the depthwise result (fp32, taps in
// (ky, kx) order as dwconv_tx_kernel), rounds them to bf16 into the A tile [256 px][32 ch] and the group's 32 x 32 matrix runs on the bf16 MFMA.
// (Tried and not kept, gpurun r06n: a wave per 8-channel chunk with the tap weights as SCALAR operands -- s_load + SALU unpack, v_fmac with an
// SGPR source, each input vector unpacked once, 86 VGPRs -- runs 79 us against this version's 62: the scalar loads of a rolled row loop sit
// on the critical path of every filter row.)
#ifndef SA_DW5_WG
#define SA_DW5_WG 4
#endif
#ifndef SA_DW5_ORDER
#define SA_DW5_ORDER 0      // 0 = a filter column per iteration (input rows shared by the output rows), 1 = the first version: a filter row per iteration, taps in dwconv_tx_kernel's order
#endif
__global__ __launch_bounds__(256, SA_DW5_WG) void dw5_g1x1_kernel(const bf16_t* __restrict__ in, const bf16_t* __restrict__ wdw, const bf16_t* __restrict__ wg,
                                                      bf16_t* __restrict__ out, const bf16_t* __restrict__ zero, int H, int W, int C, int tiles_x) {
    constexpr int TH = 8, TW = 32, PH = TH + 4, PW = TW + 4;
    __shared__ __attribute__((aligned(16))) unsigned char in_t[PH * PW * 64];
    __shared__ __attribute__((aligned(16))) float wd[25 * 32];
    unsigned char* a_t = in_t;                               // the A tile [256 px][32 ch] takes the patch's place once every thread is done reading it: 31 KB, four workgroups per CU
    static_assert(TH * TW * 64 <= PH * PW * 64, "A tile inside the patch");
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int g = blockIdx.y, b = blockIdx.z;
    const int y0 = ((int)blockIdx.x / tiles_x) * TH, x0 = ((int)blockIdx.x % tiles_x) * TW;
    const bf16_t* img = in + (long)b * H * W * C + g * 32;
    {   // patch: direct-to-LDS requests, all in flight at once (16-byte slot = pixel * 4 + chunk, 27 instructions of 64 slots; outside the image
        // the lane reads the zero page). Staged through registers in a rolled loop it was one global round trip per 16 bytes and thread, 7 in a row.
        typedef const __attribute__((address_space(1))) void* gptr_t;
        typedef __attribute__((address_space(3))) void* lptr_t;
        const int wq = __builtin_amdgcn_readfirstlane(wv);
        static_assert(PH * PW * 4 % 64 == 0, "whole instructions");
#pragma unroll
        for (int k = 0; k < (PH * PW * 4 / 64 + 3) / 4; ++k) {
            const int q = k * 4 + wq;
            if (q < PH * PW * 4 / 64) {
                const int idx = q * 64 + lane, px = idx >> 2, c = idx & 3, ty = px / PW, tx = px - ty * PW;
                const int gy = y0 - 2 + ty, gx = x0 - 2 + tx;
                const bool inb = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
                const bf16_t* gp = inb ? img + ((long)gy * W + gx) * C + c * 8 : zero;
                __builtin_amdgcn_global_load_lds((gptr_t)gp, (lptr_t)(in_t + q * 1024), 16, 0, 0);
            }
        }
    }
    if (tid < 100) {
        const int tap = tid >> 2, c = tid & 3;
        float t[8];
        unpack16(*reinterpret_cast<const uint4*>(wdw + (long)tap * C + g * 32 + c * 8), t, (bf16_t*)nullptr);
#pragma unroll
        for (int e = 0; e < 8; ++e) wd[tap * 32 + c * 8 + e] = t[e];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    {
        const int chunk = tid & 3, x = (tid >> 2) & 31, yg = tid >> 7;
        float acc[4][8];
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[o][e] = 0.f;
#if SA_DW5_ORDER == 0
#pragma unroll 1
        for (int kx = 0; kx < 5; ++kx) {                    // (rolled: unrolled, hipcc hoists all the LDS reads and spills)
            // one filter COLUMN at a time: the thread's eight input rows at column x + kx are read and unpacked once and feed all (output row, ky)
            // pairs -- 40 LDS reads and 320 unpack operations per thread where the row-at-a-time loop below spends 100 and 800 (62 -> 47 us). An
            // accumulator sees its taps in (kx, ky) order instead of dwconv_tx_kernel's (ky, kx): an fp32 re-association, the class this form is in
            float wc[5][8];
#pragma unroll
            for (int ky = 0; ky < 5; ++ky) {
                const float4 w0 = *reinterpret_cast<const float4*>(wd + (ky * 5 + kx) * 32 + chunk * 8);
                const float4 w1 = *reinterpret_cast<const float4*>(wd + (ky * 5 + kx) * 32 + chunk * 8 + 4);
                wc[ky][0] = w0.x; wc[ky][1] = w0.y; wc[ky][2] = w0.z; wc[ky][3] = w0.w;
                wc[ky][4] = w1.x; wc[ky][5] = w1.y; wc[ky][6] = w1.z; wc[ky][7] = w1.w;
            }
#pragma unroll
            for (int hr = 0; hr < 2; ++hr) {                 // four rows at a time: 16 instead of 32 registers of raw rows (four workgroups per CU)
                uint4 raw[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) raw[r] = *reinterpret_cast<const uint4*>(in_t + ((yg * 4 + hr * 4 + r) * PW + x + kx) * 64 + chunk * 16);
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int r = hr * 4 + r4;
                    float xv[8];
                    unpack16(raw[r4], xv, (bf16_t*)nullptr);
#pragma unroll
                    for (int o = 0; o < 4; ++o) {
                        const int ky = r - o;
                        if (ky >= 0 && ky < 5) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) acc[o][e] += xv[e] * wc[ky][e];
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#else
#pragma unroll 1
        for (int ky = 0; ky < 5; ++ky) {                    // (rolled, and one output row at a time below: unrolled, hipcc hoists all 150 LDS reads and spills)
            float wr[5][8];
#pragma unroll
            for (int kx = 0; kx < 5; ++kx) {
                const float4 w0 = *reinterpret_cast<const float4*>(wd + (ky * 5 + kx) * 32 + chunk * 8);
                const float4 w1 = *reinterpret_cast<const float4*>(wd + (ky * 5 + kx) * 32 + chunk * 8 + 4);
                wr[kx][0] = w0.x; wr[kx][1] = w0.y; wr[kx][2] = w0.z; wr[kx][3] = w0.w;
                wr[kx][4] = w1.x; wr[kx][5] = w1.y; wr[kx][6] = w1.z; wr[kx][7] = w1.w;
            }
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                const int row = yg * 4 + o + ky;
#pragma unroll
                for (int kx = 0; kx < 5; ++kx) {
                    float xv[8];
                    unpack16(*reinterpret_cast<const uint4*>(in_t + (row * PW + x + kx) * 64 + chunk * 16), xv, (bf16_t*)nullptr);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[o][e] += xv[e] * wr[kx][e];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#endif
        __syncthreads();                                     // every thread is done with the patch: the A tile may overwrite it
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const int px = (yg * 4 + o) * TW + x;
            *reinterpret_cast<uint4*>(a_t + px * 64 + ((chunk ^ ((px >> 2) & 3)) << 4)) =
                make_uint4(pack2(acc[o][0], acc[o][1]), pack2(acc[o][2], acc[o][3]), pack2(acc[o][4], acc[o][5]), pack2(acc[o][6], acc[o][7]));
        }
    }
    __syncthreads();
    const int lr = lane & 31, lh = lane >> 5;
    u32x4 wf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) wf[ks] = *reinterpret_cast<const u32x4*>(wg + (long)(g * 32 + lr) * 32 + ks * 16 + lh * 8);
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int mt = wv * 2 + m, px = mt * 32 + lr;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const u32x4 xf = *reinterpret_cast<const u32x4*>(a_t + px * 64 + (((ks * 2 + lh) ^ ((px >> 2) & 3)) << 4));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[ks]), __builtin_bit_cast(bf16x8, xf), acc, 0, 0, 0);
        }
        const int gy = y0 + mt, gx = x0 + lr;
        if (gy < H && gx < W) {
            bf16_t* dst = out + (((long)b * H + gy) * W + gx) * C + g * 32;
#pragma unroll
            for (int q = 0; q < 4; ++q) store4(dst + 8 * q + 4 * lh, acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
        }
    }
}

static inline int launch_dw5_g1x1(const bf16_t* in, const bf16_t* wdw, const bf16_t* wg, bf16_t* out, const bf16_t* zero, int B, int H, int W, int C,
                                  hipStream_t s) {
    const int tx = cdiv(W, 32), ty = cdiv(H, 8);
    hipLaunchKernelGGL(dw5_g1x1_kernel, dim3(tx * ty, C / 32, B), dim3(256), 0, s, in, wdw, wg, out, zero, H, W, C, tx);
    return (int)hipGetLastError();
}
template <typename T>
static inline int launch_dw5_g1x1(const T*, const T*, const T*, T*, const T*, int, int, int, int, hipStream_t) { return SA_ERR_UNSUPPORTED; }

// ---------------------------------------------------------------------------------------------------
// Decode head with z0 inside (SA_DET_UPSUM_CLASSIFY + the stage-0 1x1 convolution that feeds it). head_upsum_classify_blk_kernel read
// z0 = A0 x0 + c ([P, 512] bf16, 1.07 GB per 16 pages of 1024^2) that a GEMM launch had just written. Here a workgroup of 256 threads owns 16
// blocks of 4 x 2 pixels (128 pixels); per 128-channel slab its 4 waves compute the slab of z0 for those pixels on the MFMA (x0 fragments straight
// from global memory, loaded once: K = 64; A0 fragments from L2), add the bias in fp32, round to bf16 -- the GEMM epilogue's arithmetic, bit
// for bit -- and hand the slab over in 32 KiB of LDS ([128 px][128 ch], 16-byte chunks XOR-swizzled by the pixel row); from there on the
// pass is head_upsum_classify_blk_kernel's: 16 lanes per block, lane `sub` takes channels [slab + 8 sub, + 8), same tap tiles, same sums.
// A0 comes FRAGMENT-MAJOR ([C / 32][4 K steps][64 lanes][8], det_mbconv.h's mbconv_w2_fragments at engine init): a fragment load is one contiguous KiB
// (from the row-major [C][64] weight it touched 32 cache lines, 16 such loads per slab at one wave per SIMD).
template <int R1, int R2, int R3, int BH>
__global__ __launch_bounds__(256) void head_z0_kernel(const bf16_t* __restrict__ x0, const bf16_t* __restrict__ A0, const bf16_t* __restrict__ zb,
                                                      const bf16_t* __restrict__ z1, const bf16_t* __restrict__ z2, const bf16_t* __restrict__ z3,
                                                      const bf16_t* __restrict__ w, const bf16_t* __restrict__ bias, float* __restrict__ out,
                                                      int B, int H0, int W0, int C, int L) {
    typedef bf16_t T;
    constexpr int K = 64, NP = 4, V = 8;
    static_assert(BH == 2, "4 x 2 pixel blocks: 8 GEMM rows per block");
    __shared__ __attribute__((aligned(16))) unsigned char slab[128 * 256];     // [128 px][256 B]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, lr = lane & 31, lh = lane >> 5;
    const int sub = tid & 15, bl = tid >> 4;
    const int bw = W0 / 4, bh = H0 / BH;
    const long nblk = (long)B * bh * bw;
    // producer side: GEMM row m = wv * 32 + lr = block (m >> 3), pixel (py, px) = ((m >> 2) & 1, m & 3)
    u32x4 xf[4];
    {
        const int m = wv * 32 + lr;
        long gm = (long)blockIdx.x * 16 + (m >> 3);
        gm = gm < nblk ? gm : nblk - 1;
        const int bxm = (int)(gm % bw), bym = (int)((gm / bw) % bh);
        const long im = gm / ((long)bw * bh);
        const bf16_t* xp = x0 + ((im * H0 + BH * bym + ((m >> 2) & 1)) * W0 + 4 * bxm + (m & 3)) * K;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) xf[ks] = *reinterpret_cast<const u32x4*>(xp + (ks * 2 + lh) * 8);
    }
    // consumer side
    const long g = (long)blockIdx.x * 16 + bl;
    const long gc = g < nblk ? g : nblk - 1;
    const int bx = (int)(gc % bw), by = (int)((gc / bw) % bh);
    const long img = gc / ((long)bw * bh);
    float acc[BH][4][2];
#pragma unroll
    for (int py = 0; py < BH; ++py)
#pragma unroll
        for (int px = 0; px < 4; ++px)
#pragma unroll
            for (int l = 0; l < 2; ++l) acc[py][px][l] = 0.f;
    const long w1off = (long)(L > 1 ? 1 : 0) * C;
    // Software pipeline over the 128-channel slabs (first version: load A0 -> MFMA -> barrier -> load taps -> sum, every load's round trip
    // exposed at one wave per SIMD: 804 us, no better than the two launches it replaced):
    //   [A] request the tap tiles + classifier weights of slab s   (global, independent of LDS)
    //   [B] z0 slab s on the MFMA from the A0 fragments requested during slab s - 1's sums, bias, round, into LDS       (covers [A])
    //   [C] barrier, read this lane's rows of the slab, barrier
    //   [D] request A0 fragments + bias of slab s + 1 (clamped at the tail: unconditional)
    //   [E] the sums, ReLU, classifier dot products of slab s                                                           (covers [D])
    u32x4 wf[4][4];
    uint2 braw[4][4];
#define HZ_LOADW(N0)                                                                                                    \
    {                                                                                                                   \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                                \
            _Pragma("unroll") for (int ks_ = 0; ks_ < 4; ++ks_)                                                         \
                wf[j_][ks_] = *reinterpret_cast<const u32x4*>(A0 + ((long)(((N0) >> 5) + j_) * 4 + ks_) * 512 + lane * 8); \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                                \
            _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) braw[j_][q_] = *reinterpret_cast<const uint2*>(zb + (N0) + j_ * 32 + q_ * 8 + lh * 4); \
    }
    HZ_LOADW(0);
    for (int n0 = 0; n0 < C; n0 += 128) {
        const int c = n0 + sub * V;
        UpsumTile<R1, BH> t1; UpsumTile<R2, BH> t2; UpsumTile<R3, BH> t3;
        upsum_load<T, R1, BH>(t1, z1 + c, img, H0 / R1, W0 / R1, C, by, bx);
        upsum_load<T, R2, BH>(t2, z2 + c, img, H0 / R2, W0 / R2, C, by, bx);
        upsum_load<T, R3, BH>(t3, z3 + c, img, H0 / R3, W0 / R3, C, by, bx);
        const uint4 w0r = *reinterpret_cast<const uint4*>(w + c), w1r = *reinterpret_cast<const uint4*>(w + w1off + c);
        __builtin_amdgcn_sched_barrier(0);
        {   // z0 slab [128 px][128 ch]
            f32x16 za[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int r = 0; r < 16; ++r) za[j][r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    za[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[j][ks]), __builtin_bit_cast(bf16x8, xf[ks]), za[j], 0, 0, 0);
            }
            const int row = wv * 32 + lr;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float bq[4];
                    load4(reinterpret_cast<const bf16_t*>(&braw[j][q]), bq);
                    const int chunk = j * 4 + q;
                    *reinterpret_cast<uint2*>(slab + row * 256 + ((chunk ^ (row & 15)) << 4) + lh * 8) =
                        make_uint2(pack2(za[j][4 * q] + bq[0], za[j][4 * q + 1] + bq[1]), pack2(za[j][4 * q + 2] + bq[2], za[j][4 * q + 3] + bq[3]));
                }
        }
        __syncthreads();
        uint4 zr[BH][4];
#pragma unroll
        for (int py = 0; py < BH; ++py)
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                const int m = bl * 8 + py * 4 + px;
                zr[py][px] = *reinterpret_cast<const uint4*>(slab + m * 256 + ((sub ^ (m & 15)) << 4));
            }
        __syncthreads();                                   // every lane holds its rows of the slab: the next slab may overwrite it
        __builtin_amdgcn_sched_barrier(0);
        HZ_LOADW(min(n0 + 128, C - 128));
        __builtin_amdgcn_sched_barrier(0);
        f32x2 v[BH][4][NP];
#pragma unroll
        for (int py = 0; py < BH; ++py)
#pragma unroll
            for (int px = 0; px < 4; ++px) UpsumPk<T>::unpack(zr[py][px], v[py][px]);
        upsum_apply<T, R1, BH, NP>(v, t1, by, bx);
        upsum_apply<T, R2, BH, NP>(v, t2, by, bx);
        upsum_apply<T, R3, BH, NP>(v, t3, by, bx);
        const uint32_t w0u[4] = {w0r.x, w0r.y, w0r.z, w0r.w}, w1u[4] = {w1r.x, w1r.y, w1r.z, w1r.w};
#pragma unroll
        for (int py = 0; py < BH; ++py)
#pragma unroll
            for (int px = 0; px < 4; ++px)
#pragma unroll
                for (int i = 0; i < NP; ++i) {
                    const f32x2 yv = f32x2{fmaxf(v[py][px][i].x, 0.f), fmaxf(v[py][px][i].y, 0.f)};
                    const bf16x2_t yb = __builtin_convertvector(yv, bf16x2_t);
                    acc[py][px][0] = __builtin_amdgcn_fdot2_f32_bf16(yb, __builtin_bit_cast(bf16x2_t, w0u[i]), acc[py][px][0], false);
                    acc[py][px][1] = __builtin_amdgcn_fdot2_f32_bf16(yb, __builtin_bit_cast(bf16x2_t, w1u[i]), acc[py][px][1], false);
                }
    }
#undef HZ_LOADW
    const long HW = (long)H0 * W0;
    const float blv[2] = {Ty<T>::ld(bias), Ty<T>::ld(bias + (L > 1 ? 1 : 0))};
#pragma unroll
    for (int py = 0; py < BH; ++py)
#pragma unroll
        for (int px = 0; px < 4; ++px)
#pragma unroll
            for (int l = 0; l < 2; ++l) {
                const float a = row16_sum(acc[py][px][l]);
                if (sub == 0 && g < nblk && l < L) {
                    const float z = Ty<T>::rnd(a + blv[l]);
                    out[(img * L + l) * HW + (long)(BH * by + py) * W0 + 4 * bx + px] = Ty<T>::rnd(1.0f / (1.0f + expf(-z)));
                }
            }
}

static inline int launch_head_z0(const bf16_t* x0, const bf16_t* A0, const bf16_t* zb, const bf16_t* z1, const bf16_t* z2, const bf16_t* z3,
                                 const bf16_t* w, const bf16_t* bias, float* planes, int B, int H0, int W0, int K, int C, int L, hipStream_t s) {
    if (K != 64 || C % 128 || L > 2 || H0 % 8 || W0 % 8 || !zb) return SA_ERR_SHAPE;
    const long nblk = (long)B * (H0 / 2) * (W0 / 4);
    hipLaunchKernelGGL((head_z0_kernel<2, 4, 8, 2>), dim3((unsigned)cdivl(nblk, 16)), dim3(256), 0, s, x0, A0, zb, z1, z2, z3, w, bias, planes, B, H0, W0, C, L);
    return (int)hipGetLastError();
}
template <typename T>
static inline int launch_head_z0(const T*, const T*, const T*, const T*, const T*, const T*, const T*, const T*, float*, int, int, int, int, int,
                                 int, hipStream_t) { return SA_ERR_UNSUPPORTED; }

// ---------------------------------------------------------------------------------------------------
// MBConv tail (encoderdecoder.py:174-225): depthwise 3x3 (+ bias + Hardswish) -> projection 1x1 (+ folded BN bias, + residual).
// Workgroup = 8 x 16 output pixels x 256 output channels (blockIdx.y picks the 256-column half when Cout = 512: both halves then do the
// depthwise arithmetic of their pixels), 8 waves in two roles -- one of each on every SIMD, so the vector ALU (depthwise) and the matrix
// pipe (projection) of a SIMD work at the same time:
//   producers (waves 4..7): per 64-channel chunk of the expanded tensor the depthwise result of the tile, each thread 4 horizontally
//     adjacent pixels x 8 channels: inputs straight from global memory (16-byte loads, 8 lanes per 128-byte pixel row), fp32 in
//     dwconv_tx_kernel's (ky, kx) order with the bias first, Hardswish, round to bf16, into the A tile [128 px][64 ch] in LDS (gemm.h's XOR
//     swizzle), double buffered. ONE register set of inputs: an input vector is re-requested for the next chunk right after its last
//     use, so every load has about one chunk of arithmetic (~1500 cycles) to arrive. The chunk's 9 filter taps + bias come through a
//     two-slot LDS ring (80 threads fetch the next chunk's, one chunk ahead).
//   consumers (waves 0..3): wave w owns output channels [64 w, 64 w + 64) for all 128 pixels (4 x 2 MFMA tiles, 128 accumulator registers);
//     its W fragments are its own 64 rows of the projection weight -- no other wave reads them, so they skip LDS -- re-requested for the
//     next chunk right after use as well.
//   One s_barrier per chunk: behind it chunk c is complete in buffer c & 1 and every consumer has finished reading chunk c - 1.
// K order and MFMA are the projection GEMM's (one accumulator per output, K-tiles ascending, kk 0..3): with identical A bits the sums are the
// GEMM's, and the epilogue repeats its rounding (bias in fp32, round; + residual, round): bit-identical to the two launches it replaces.
#ifndef SA_DWP_ABL
#define SA_DWP_ABL 0     // ablation builds (tools/microbench/dwproj_ablate.sh): 1 = producers never re-request inputs, 2 = consumers skip the MFMAs, 4 = producers skip the depthwise arithmetic, 8 = consumers never re-request W
#endif
template <int S>
__global__ __launch_bounds__(512) void dwproj_kernel(const bf16_t* __restrict__ in, const bf16_t* __restrict__ wd, const bf16_t* __restrict__ bd,
                                                     const bf16_t* __restrict__ w2, const bf16_t* __restrict__ b2,
                                                     const bf16_t* __restrict__ res, bf16_t* __restrict__ out, int H, int W, int Cm, int Ho, int Wo,
                                                     int Cout, int tiles_x, int tiles_y) {
    constexpr int CW = 256, TX = 4, NIN = (TX - 1) * S + 3;
    constexpr int WRING = 65536, WSLOT = 1280;              // LDS: [0, 32 KiB) A tiles; [0, 64 KiB) the output tile (epilogue); then the tap ring
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 31, lh = lane >> 5;
    const int n0 = (int)blockIdx.y * CW;
    // XCD x takes the x-th contiguous eighth of the tile raster (rows that share input rows meet in one L2, see dwconv_tx_kernel)
    const unsigned nb = gridDim.x, xcd = blockIdx.x & 7, per = nb >> 3, rem = nb & 7;
    const int bid = (int)(xcd * per + min(xcd, rem) + (blockIdx.x >> 3));
    const int b = bid / (tiles_x * tiles_y), tr = bid - b * tiles_x * tiles_y;
    const int oy0 = (tr / tiles_x) * 8, ox0 = (tr % tiles_x) * 16;
    const int nch = Cm / 64;                                 // even (launcher)
    constexpr int ROWB = CW * 2, CPR = ROWB / 16;            // epilogue tile [128 px][256] bf16, 16-byte chunks XOR-swizzled by the pixel row

    if (wv >= 4) {
        // ------------------------------------------------------------------ producers
        const int pt = tid - 256;
        const int dch = pt & 7, dxg = (pt >> 3) & 3, dy = pt >> 5;
        const int iy0 = (oy0 + dy) * S - 1, ix0 = (ox0 + dxg * TX) * S - 1;
        const bf16_t* img = in + (long)b * H * W * Cm + dch * 8;
        int roff[3], coff[NIN];                              // (sums and masks are formed per load: 3 x NIN of each kept live spill)
        unsigned rmask[3], cmask[NIN];
#pragma unroll
        for (int r = 0; r < 3; ++r) { const int iy = iy0 + r; roff[r] = min(max(iy, 0), H - 1) * W * Cm; rmask[r] = (unsigned)iy < (unsigned)H ? 0xffffffffu : 0u; }
#pragma unroll
        for (int j = 0; j < NIN; ++j) { const int ix = ix0 + j; coff[j] = min(max(ix, 0), W - 1) * Cm; cmask[j] = (unsigned)ix < (unsigned)W ? 0xffffffffu : 0u; }
        // tap ring: thread pt < 80 carries 16 bytes of (tap pt >> 3 | bias = row 9), channels (pt & 7) * 8 of the chunk
        const bool wl = pt < 80;
        const bf16_t* wsrc = (pt >> 3) < 9 ? wd + (long)(pt >> 3) * Cm + (pt & 7) * 8 : bd + (pt & 7) * 8;
        u32x4 wld;
        if (wl) {
            *reinterpret_cast<u32x4*>(smem + WRING + pt * 16) = *reinterpret_cast<const u32x4*>(wsrc);
            wld = *reinterpret_cast<const u32x4*>(wsrc + (nch > 1 ? 64 : 0));
        }
        u32x4 raw[3][NIN];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int j = 0; j < NIN; ++j) raw[r][j] = *reinterpret_cast<const u32x4*>(img + (roff[r] + coff[j]));
        __syncthreads();                                     // (P0) tap slot 0 visible
        for (int c = 0; c < nch; ++c) {
            const unsigned char* ws = smem + WRING + (c & 1) * WSLOT + dch * 16;
            // (every re-request below is UNCONDITIONAL, the chunk index clamped at the tail: behind `if (more)` hipcc branches around each load
            // and drains vmcnt(0) per element -- the first ISA of this loop had 27 branches per chunk)
            if (wl) {
                *reinterpret_cast<u32x4*>(smem + WRING + ((c + 1) & 1) * WSLOT + pt * 16) = wld;      // chunk c + 1's taps (visible behind this chunk's barrier)
                wld = *reinterpret_cast<const u32x4*>(wsrc + min(c + 2, nch - 1) * 64);
            }
            f32x2 a2[TX][4];
            {
                f32x2 bv[4];
                UpsumPk<bf16_t>::unpack(*reinterpret_cast<const uint4*>(ws + 9 * 128), bv);
#pragma unroll
                for (int o = 0; o < TX; ++o)
#pragma unroll
                    for (int e = 0; e < 4; ++e) a2[o][e] = bv[e];
            }
            const int cn = min(c + 1, nch - 1) * 64;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                f32x2 w3[3][4];
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) UpsumPk<bf16_t>::unpack(*reinterpret_cast<const uint4*>(ws + (ky * 3 + kx) * 128), w3[kx]);
#pragma unroll
                for (int j = 0; j < NIN; ++j) {
                    const unsigned m = rmask[ky] & cmask[j];
                    f32x2 x[4];
                    UpsumPk<bf16_t>::unpack(make_uint4(raw[ky][j][0] & m, raw[ky][j][1] & m, raw[ky][j][2] & m, raw[ky][j][3] & m), x);
                    if constexpr (!(SA_DWP_ABL & 1)) raw[ky][j] = *reinterpret_cast<const u32x4*>(img + (roff[ky] + coff[j] + cn));       // its last use was the line above
#pragma unroll
                    for (int o = 0; o < TX; ++o) {
                        const int kx = j - o * S;
                        if (kx >= 0 && kx < 3) {
                            if constexpr (SA_DWP_ABL & 4) { if (kx == 0 && ky == 0) a2[o][0] += x[0]; }
                            else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) a2[o][e] += x[e] * w3[kx][e];
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int o = 0; o < TX; ++o) {
                float r[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v0 = a2[o][e].x, v1 = a2[o][e].y;
                    r[2 * e] = hardswish_f(v0);
                    r[2 * e + 1] = hardswish_f(v1);
                }
                const int row = dy * 16 + dxg * TX + o;
                *reinterpret_cast<uint4*>(smem + (c & 1) * 16384 + row * 128 + ((dch ^ ((row >> 1) & 7)) << 4)) =
                    make_uint4(pack2(r[0], r[1]), pack2(r[2], r[3]), pack2(r[4], r[5]), pack2(r[6], r[7]));
            }
            __syncthreads();                                 // (B_c) chunk c is in buffer c & 1
        }
        __syncthreads();                                     // (E1)
    } else {
        // ------------------------------------------------------------------ consumers
        f32x16 acc[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;
        // w2 comes FRAGMENT-MAJOR ([Cm / 64][Cout / 32][4 K steps][64 lanes][8]: det_mbconv.h's mbconv_w2_fragments at engine init): a fragment load is
        // one contiguous KiB instead of 32 cache lines of the row-major [Cout][Cm] weight
        const bf16_t* w2p = w2 + ((long)((n0 + wv * 64) >> 5) * 4 * 64 + lane) * 8;
        const long w2cs = (long)(Cout >> 5) * 4 * 64 * 8;     // elements per 64-channel chunk
        // W fragments: two named sets; all eight 16-byte loads of the NEXT chunk are issued in one burst at the top of a chunk (the four
        // loads of a 128-byte weight row then meet in L1; spread over the chunk, one per kk, every one of them went to L2 again)
        u32x4 wa[2][4], wb[2][4];
#define DP_LW(WF, C0)                                                                                                   \
    {                                                                                                                   \
        _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                                                \
            _Pragma("unroll") for (int kk_ = 0; kk_ < 4; ++kk_)                                                         \
                WF[j_][kk_] = *reinterpret_cast<const u32x4*>(w2p + ((C0) >> 6) * w2cs + (j_ * 4 + kk_) * 512);         \
    }
#define DP_MM(WF, BUF)                                                                                                  \
    {                                                                                                                   \
        const unsigned char* at_ = smem + (BUF) * 16384;                                                                \
        _Pragma("unroll") for (int kk_ = 0; kk_ < 4; ++kk_) {                                                           \
            u32x4 xf_[4];                                                                                               \
            _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                          \
                const int row_ = i_ * 32 + lr;                                                                          \
                xf_[i_] = *reinterpret_cast<const u32x4*>(at_ + row_ * 128 + (((kk_ * 2 + lh) ^ ((row_ >> 1) & 7)) << 4)); \
            }                                                                                                           \
            if constexpr (SA_DWP_ABL & 2) { asm volatile("" :: "v"(xf_[0]), "v"(xf_[1]), "v"(xf_[2]), "v"(xf_[3]), "v"(WF[0][kk_]), "v"(WF[1][kk_])); } else \
            _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                                            \
                _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                        \
                    acc[j_][i_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, WF[j_][kk_]), __builtin_bit_cast(bf16x8, xf_[i_]), acc[j_][i_], 0, 0, 0); \
        }                                                                                                               \
    }
        DP_LW(wa, 0);
        if constexpr (SA_DWP_ABL & 8) DP_LW(wb, 0);
        __syncthreads();                                     // (P0)
        for (int c = 0; c < nch; c += 2) {                   // nch is even
            if constexpr (!(SA_DWP_ABL & 8)) DP_LW(wb, (c + 1) * 64);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();                                 // (B_c)
            DP_MM(wa, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!(SA_DWP_ABL & 8)) DP_LW(wa, min(c + 2, nch - 1) * 64);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();                                 // (B_c+1)
            DP_MM(wb, 1);
            __builtin_amdgcn_sched_barrier(0);
        }
#undef DP_LW
#undef DP_MM
        // ---- epilogue through LDS; then whole rows out, residual added on the way
        __syncthreads();                                     // (E1) every consumer is done with the A tiles
        float bq[2][4][4];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) load4(b2 + n0 + wv * 64 + j * 32 + q * 8 + lh * 4, bq[j][q]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = i * 32 + lr;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int chunk = (wv * 64 + j * 32 + q * 8) >> 3;
                    *reinterpret_cast<uint2*>(smem + row * ROWB + ((chunk ^ (row & 31)) << 4) + lh * 8) =
                        make_uint2(pack2(acc[j][i][4 * q] + bq[j][q][0], acc[j][i][4 * q + 1] + bq[j][q][1]),
                                   pack2(acc[j][i][4 * q + 2] + bq[j][q][2], acc[j][i][4 * q + 3] + bq[j][q][3]));
                }
        }
    }
    __syncthreads();                                         // (E2)
    // whole rows out: this thread's eight 16-byte pieces. All eight residual requests first (rolled, the loop was load -> vmcnt(0) -> add -> store per
    // piece: eight memory round trips in a row at the end of every workgroup), then the adds and stores; rows outside the image load a clamped address and store nothing
    {
        constexpr int NP = 128 * CPR / 512;
        static_assert(128 * CPR % 512 == 0, "whole pieces per thread");
        long offs[NP];
        bool ok[NP];
        uint4 rr[NP];
#pragma unroll
        for (int it = 0; it < NP; ++it) {
            const int id = tid + it * 512, row = id / CPR, cc = id % CPR;
            const int oy = oy0 + (row >> 4), ox = ox0 + (row & 15);
            ok[it] = oy < Ho && ox < Wo;
            offs[it] = (((long)b * Ho + min(oy, Ho - 1)) * Wo + min(ox, Wo - 1)) * Cout + n0 + cc * 8;
            if (res) rr[it] = *reinterpret_cast<const uint4*>(res + offs[it]);
        }
#pragma unroll
        for (int it = 0; it < NP; ++it) {
            const int id = tid + it * 512, row = id / CPR, cc = id % CPR;
            const uint4 rawo = *reinterpret_cast<const uint4*>(smem + row * ROWB + ((cc ^ (row & 31)) << 4));
            if (res) {
                float a[8], r8[8];
                unpack16(rawo, a, (bf16_t*)nullptr);
                unpack16(rr[it], r8, (bf16_t*)nullptr);
                if (ok[it]) {
                    store4(out + offs[it], a[0] + r8[0], a[1] + r8[1], a[2] + r8[2], a[3] + r8[3]);
                    store4(out + offs[it] + 4, a[4] + r8[4], a[5] + r8[5], a[6] + r8[6], a[7] + r8[7]);
                }
            } else if (ok[it]) {
                *reinterpret_cast<uint4*>(out + offs[it]) = rawo;
            }
        }
    }
}

static inline int launch_dwproj(const bf16_t* in, const bf16_t* wd, const bf16_t* bd, int act, const bf16_t* w2, const bf16_t* b2, const bf16_t* res,
                                bf16_t* out, int B, int H, int W, int Cm, int Ho, int Wo, int Cout, int stride, hipStream_t s) {
    if (Cm % 128 || (Cout != 256 && Cout != 512) || (stride != 1 && stride != 2) || (long)B * H * W * Cm >= (1L << 31) || act != ACT_HSWISH) return SA_ERR_SHAPE;
    const int tx = cdiv(Wo, 16), ty = cdiv(Ho, 8);
    const unsigned grid = (unsigned)(B * tx * ty);
#define SA_DWPROJ(SS)                                                                                                           \
    {                                                                                                                           \
        constexpr size_t lds = 65536 + 2 * 1280;                                                                                \
        auto kern = dwproj_kernel<SS>;                                                                                          \
        static AttrOnce attr;                                                                                                   \
        attr.ensure(kern, lds);                                                                                                 \
        hipLaunchKernelGGL(kern, dim3(grid, Cout / 256), dim3(512), lds, s, in, wd, bd, w2, b2, res, out, H, W, Cm, Ho, Wo, Cout, tx, ty); \
    }
    if (stride == 1) SA_DWPROJ(1) else SA_DWPROJ(2)
#undef SA_DWPROJ
    return (int)hipGetLastError();
}
template <typename T>
static inline int launch_dwproj(const T*, const T*, const T*, int, const T*, const T*, const T*, T*, int, int, int, int, int, int, int, int,
                                hipStream_t) { return SA_ERR_UNSUPPORTED; }

// ---------------------------------------------------------------------------------------------------
// Stem convolutions: 3x3, 32 output channels, Cin = 8 (the padded RGB input, stride 2) or 32 (the residual ConvBlock), at 512^2 x 16 pages the
// three largest pixel counts of the network. As implicit GEMMs they are M = 4.2M x N = 32: the 128 x 32 tile of conv_gemm_kernel re-gathered
// every input vector nine times through L1 / L2 and ran at 80 - 250 TF/s (240 - 340 us each against ~100 us of HBM time for one read and one
// write of the tensor). Here a workgroup owns an 8 x 32 pixel tile: the input patch ((8 - 1) S + 3 rows x (32 - 1) S + 3 columns, zero outside
// the image) is loaded ONCE into LDS and every MFMA's pixel fragment is a 16-byte LDS read at (pixel + tap) -- im2col without the copy; the
// whole 32 x K weight matrix sits in registers as MFMA fragments (72 registers at Cin = 32), so the K loop is 18 x (2 LDS reads + 2 MFMAs) per
// wave with nothing else in it. 4 waves, wave w = rows 2 w, 2 w + 1 of the tile. K order = the GEMM's ((ky, kx, ci) ascending in 16-element steps,
// one accumulator per output) and the epilogue is conv_gemm_kernel's (bias, Hardswish, + residual in fp32, ONE rounding): bit-identical to it.
//   Cin = 32: K step s = tap s >> 1, channels (s & 1) * 16 + (lane >> 5) * 8; LDS pixel = 64 bytes, 16-byte chunk XOR-swizzled by (column >> 2) & 3
//   Cin = 8 : K step s = taps 2 s + (lane >> 5) (tap 9 = zero weights, reads tap 8's pixel), LDS pixel = 16 bytes
// The 32 x 32 result D[cout][px] leaves a lane with 4 consecutive channels per group; v_permlane32_swap between lane l and l + 32 (same pixel)
// makes that 8 consecutive channels, so the stores are 16 bytes.
// SRC (Cin = 8 only): where the patch comes from. 0 = the NHWC bf16 activation buffer; 1 = the caller's fp32 NCHW pixel_values (3 planes), 2 = uint8
// NHWC pages with SegformerImageProcessor's rescale + normalise -- the input-layout kernels' conversions (nchw_to_nhwc_kernel / u8_to_nhwc_kernel,
// bit for bit) done in the patch loader, so the 8-channel copy of the page (16 bytes per pixel written and read back: 0.54 GB per 16 pages) never exists.
struct StemSrc { const float* planes; const unsigned char* u8; float m0, m1, m2, s0, s1, s2; int pix; };
template <int SRC>
__device__ __forceinline__ uint4 stem_pixel(const StemSrc& s, int b, long hw, long p) {
    if constexpr (SRC == 1) {
        const float* q = s.planes + (long)b * 3 * hw + p;
        return make_uint4(pack2(q[0], q[hw]), pack2(q[2 * hw], 0.f), 0u, 0u);                 // nchw_to_nhwc_kernel's pixel
    } else {
#pragma clang fp contract(off)
        const unsigned char* px = s.u8 + ((long)b * hw + p) * s.pix;                          // u8_to_nhwc_kernel's pixel (float64 rescale rounded once)
        const double k = 1.0 / 255.0;
        const float v0 = ((float)((double)px[0] * k) - s.m0) / s.s0, v1 = ((float)((double)px[1] * k) - s.m1) / s.s1, v2 = ((float)((double)px[2] * k) - s.m2) / s.s2;
        return make_uint4(pack2(v0, v1), pack2(v2, 0.f), 0u, 0u);
    }
}
template <int CIN, int S, int EPI, int SRC = 0>      // EPI: 0 = bias + Hardswish, 1 = bias + residual
__global__ __launch_bounds__(256, CIN == 32 ? 2 : (SRC ? 3 : 4)) void stem_conv_kernel(const bf16_t* __restrict__ in, const bf16_t* __restrict__ w, const bf16_t* __restrict__ bias,
                                                       const bf16_t* __restrict__ res, bf16_t* __restrict__ out, int H, int W, int Ho, int Wo, int Kpad,
                                                       int tiles_x, int tiles_y, int ntiles, StemSrc src = StemSrc()) {
    static_assert(SRC == 0 || CIN == 8, "pixel sources feed the first convolution only");
    constexpr int TH = 8, TW = 32, PH = (TH - 1) * S + 3, PW = (TW - 1) * S + 3, PB = CIN * 2, CPP = PB / 16;
    constexpr int NSTEP = CIN == 32 ? 18 : 5;
    static_assert((CIN == 32 || CIN == 8), "stem shapes");
    __shared__ __attribute__((aligned(16))) unsigned char patch[PH * PW * PB];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, lr = lane & 31, lh = lane >> 5;
    // weights: all K steps of this lane's output channel (lane & 31), ONCE per workgroup: the grid is persistent (a few workgroups per CU walk the
    // tile raster). One tile per workgroup re-read the 18 KiB of fragments per wave and tile -- 1.2 GB of L2 traffic per launch for 0.27 GB of
    // activations -- and exposed four memory round trips per tile (fragments, patch, residual, stores): 251 us.
    u32x4 wf[NSTEP];
#pragma unroll
    for (int s_ = 0; s_ < NSTEP; ++s_) wf[s_] = *reinterpret_cast<const u32x4*>(w + (long)lr * Kpad + s_ * 16 + lh * 8);
    constexpr int NPRE = (PH * PW * CPP + 255) / 256;
    // XCD x (= blockIdx.x % 8) owns the x-th contiguous eighth of the tile raster; its workgroups take adjacent tiles of it
    const int xcd = blockIdx.x & 7, gx = (int)gridDim.x >> 3, wx = (int)blockIdx.x >> 3;
    const int per = ntiles >> 3, rem = ntiles & 7;
    const int t_begin = xcd * per + min(xcd, rem), t_cnt = per + (xcd < rem ? 1 : 0);
    uint4 pre[NPRE];
#define ST_LOAD(TL)                                                                                                     \
    {                                                                                                                   \
        const int bid_ = t_begin + min((TL), t_cnt - 1);                                                                \
        const int b_ = bid_ / (tiles_x * tiles_y), tr_ = bid_ - b_ * tiles_x * tiles_y;                                 \
        const int iy0_ = (tr_ / tiles_x) * TH * S - 1, ix0_ = (tr_ % tiles_x) * TW * S - 1;                             \
        const bf16_t* img_ = in + (long)b_ * H * W * CIN;                                                               \
        _Pragma("unroll") for (int i_ = 0; i_ < NPRE; ++i_) {                                                           \
            const int idx_ = min(tid + i_ * 256, PH * PW * CPP - 1), px_ = idx_ / CPP, c_ = idx_ % CPP, pr_ = px_ / PW, pc_ = px_ - pr_ * PW; \
            const int iy_ = iy0_ + pr_, ix_ = ix0_ + pc_;                                                               \
            const bool ok_ = (unsigned)iy_ < (unsigned)H && (unsigned)ix_ < (unsigned)W;                                \
            const long pp_ = (long)min(max(iy_, 0), H - 1) * W + min(max(ix_, 0), W - 1);                               \
            uint4 v_;                                                                                                   \
            if constexpr (SRC == 0) v_ = *reinterpret_cast<const uint4*>(img_ + pp_ * CIN + c_ * 8);                    \
            else v_ = stem_pixel<SRC>(src, b_, (long)H * W, pp_);                                                       \
            const unsigned m_ = ok_ ? 0xffffffffu : 0u;      /* unconditional load, masked: no branch between a load and its use */ \
            pre[i_] = make_uint4(v_.x & m_, v_.y & m_, v_.z & m_, v_.w & m_);                                           \
        }                                                                                                               \
    }
    if (wx >= t_cnt) return;
    ST_LOAD(wx);
    for (int tl = wx; tl < t_cnt; tl += gx) {
        const int bid = t_begin + tl;
        const int b = bid / (tiles_x * tiles_y), tr = bid - b * tiles_x * tiles_y;
        const int oy0 = (tr / tiles_x) * TH, ox0 = (tr % tiles_x) * TW;
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            const int idx = tid + i * 256;
            if (idx < PH * PW * CPP) {
                const int px = idx / CPP, c = idx % CPP, pc = px % PW;
                const int pc_sw = CPP == 4 ? (c ^ ((pc >> 2) & 3)) : c;
                *reinterpret_cast<uint4*>(patch + px * PB + pc_sw * 16) = pre[i];
            }
        }
        __syncthreads();
        ST_LOAD(tl + gx);                                    // the next tile's patch travels while this one is multiplied (clamped at the tail)
        // residual rows of this tile, requested before the MFMAs as well
        [[maybe_unused]] uint4 rraw[2][2];
        long obase[2];
        bool live[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int oy = oy0 + wv * 2 + i, ox = ox0 + lr;
            live[i] = oy < Ho && ox < Wo;
            obase[i] = (((long)b * Ho + min(oy, Ho - 1)) * Wo + min(ox, Wo - 1)) * 32;
            if (EPI == 1) {
#pragma unroll
                for (int p_ = 0; p_ < 2; ++p_) rraw[i][p_] = *reinterpret_cast<const uint4*>(res + obase[i] + 8 * (2 * p_ + lh));
            }
        }
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
        for (int s_ = 0; s_ < NSTEP; ++s_) {
            int tap, chunk;
            if (CIN == 32) { tap = s_ >> 1; chunk = (s_ & 1) * 2 + lh; }
            else { tap = min(2 * s_ + lh, 8); chunk = 0; }
            const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int prow = (wv * 2 + i) * S + ky, pcol = lr * S + kx;
                const int csw = CPP == 4 ? (chunk ^ ((pcol >> 2) & 3)) : chunk;
                const u32x4 xf = *reinterpret_cast<const u32x4*>(patch + (prow * PW + pcol) * PB + csw * 16);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[s_]), __builtin_bit_cast(bf16x8, xf), acc[i], 0, 0, 0);
            }
        }
        __syncthreads();                                     // every wave is done with the patch: the next tile may overwrite it
        // epilogue: lane (px = lr, half lh) ends with channels [8 (2 p + lh), + 8) for p = 0, 1
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int p_ = 0; p_ < 2; ++p_) {
                float v[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // quad g = 2 p (register 8 p + r) and quad g = 2 p + 1 (register 8 p + 4 + r): the upper half's copy of the first <-> the lower half's copy of the second
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[i][8 * p_ + r]), __float_as_uint(acc[i][8 * p_ + 4 + r]), false, false);
                    v[r] = __uint_as_float(sw[0]);
                    v[4 + r] = __uint_as_float(sw[1]);
                }
                const int c0 = 8 * (2 * p_ + lh);
                float bv[8];
                unpack16(*reinterpret_cast<const uint4*>(bias + c0), bv, (bf16_t*)nullptr);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += bv[e];
                if (EPI == 0) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = hardswish_f(v[e]);
                } else {
                    float r8[8];
                    unpack16(rraw[i][p_], r8, (bf16_t*)nullptr);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += r8[e];
                }
                if (live[i])
                    *reinterpret_cast<uint4*>(out + obase[i] + c0) = make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
            }
        }
    }
#undef ST_LOAD
}

// Returns SA_ERR_UNSUPPORTED for shapes the kernel does not take (the caller then runs the implicit-GEMM path).
static inline int launch_stem_conv(const bf16_t* in, const bf16_t* w, const bf16_t* bias, const bf16_t* res, bf16_t* out, int B, int H, int W, int Cin,
                                   int Ho, int Wo, int Cout, int k, int stride, int pad, int Kpad, int act, hipStream_t s) {
    if (Cout != 32 || k != 3 || pad != 1 || !bias) return SA_ERR_UNSUPPORTED;
    const int tx = cdiv(Wo, 32), ty = cdiv(Ho, 8), ntiles = B * tx * ty;
    const int wg_per_cu = Cin == 32 ? 2 : 4;                 // by registers (the Cin = 32 kernel holds 72 registers of weight fragments)
    const unsigned grid = (unsigned)std::min(ntiles, 256 * wg_per_cu) / 8 * 8 ? (unsigned)std::min(ntiles, 256 * wg_per_cu) / 8 * 8 : 8u;      // persistent, whole XCD rounds
    if (Cin == 32 && stride == 1 && act == ACT_HSWISH && !res)
        hipLaunchKernelGGL((stem_conv_kernel<32, 1, 0>), dim3(grid), dim3(256), 0, s, in, w, bias, res, out, H, W, Ho, Wo, Kpad, tx, ty, ntiles);
    else if (Cin == 32 && stride == 1 && act == ACT_NONE && res)
        hipLaunchKernelGGL((stem_conv_kernel<32, 1, 1>), dim3(grid), dim3(256), 0, s, in, w, bias, res, out, H, W, Ho, Wo, Kpad, tx, ty, ntiles);
    else if (Cin == 8 && stride == 2 && act == ACT_HSWISH && !res)
        hipLaunchKernelGGL((stem_conv_kernel<8, 2, 0>), dim3(grid), dim3(256), 0, s, in, w, bias, res, out, H, W, Ho, Wo, Kpad, tx, ty, ntiles);
    else
        return SA_ERR_UNSUPPORTED;
    return (int)hipGetLastError();
}
template <typename T>
static inline int launch_stem_conv(const T*, const T*, const T*, const T*, T*, int, int, int, int, int, int, int, int, int, int, int, int, hipStream_t) {
    return SA_ERR_UNSUPPORTED;
}
// ---------------------------------------------------------------------------------------------------
// The stem's residual block (encoderdecoder.py:367-390: x + conv2(hswish(conv1(x))), two 32 -> 32 3x3 convolutions at half resolution) in ONE kernel:
// the tensor between the two convolutions (16.8 MB per page written and read back) lives in LDS. One persistent workgroup of EIGHT waves per CU walks
// 8 x 32 output tiles; its waves have two roles, one of each per SIMD, and the tile loop is a two-stage pipeline with ONE workgroup barrier per tile:
//   waves 0-3 (conv1) in iteration j: request tile j + 1's 12 x 36 input patch (halo 2, zero outside the image) direct-to-LDS into the other patch
//             buffer, then compute conv1 of tile j on the 10 x 34 pixels conv2 needs (11 MFMA row tiles of the pixel-linear grid, three per wave;
//             + bias, Hardswish, round to bf16, ZERO where the pixel lies outside the image: conv2 pads the intermediate TENSOR) into inter[j & 1];
//   waves 4-7 (conv2) in iteration j: conv2 of tile j - 1 from inter[(j - 1) & 1] (two output rows per wave; an intermediate row's fragment is read
//             once and multiplied into both rows' accumulators: 24 LDS reads for 36 MFMAs), + bias + residual (the block's input, re-read from L2).
// A first version (four waves doing both phases in turn, both weight matrices = 144 registers per lane) ran 435 us against the two launches' 283: with
// no registers left the compiler serialised every LDS read in front of its MFMA, and nothing overlapped the patch's memory round trip. Here each wave
// holds ONE convolution's fragments (72 registers). K orders, MFMA and rounding points are the two launches': bit-identical to them.
#ifndef SA_SR_ABL
#define SA_SR_ABL 0      // timing ablations (results wrong): 1 = no conv1 MFMAs, 2 = no conv2 MFMAs, 4 = no patch requests after the first, 8 = no conv1 epilogue, 16 = no conv2 epilogue
#endif
__global__ __launch_bounds__(512, 1) void stem_res_kernel(const bf16_t* __restrict__ in, const bf16_t* __restrict__ w1, const bf16_t* __restrict__ b1,
                                                          const bf16_t* __restrict__ w2, const bf16_t* __restrict__ b2, bf16_t* __restrict__ out,
                                                          int H, int W, int Kpad, int tiles_x, int tiles_y, int ntiles) {
    constexpr int TH = 8, TW = 32, IH = TH + 2, IW = TW + 2, PH = TH + 4, PW = TW + 4, NI = IH * IW, NIT = (NI + 31) / 32, NSTEP = 18;
    constexpr int PATCHB = PH * PW * 64, INTERB = NIT * 32 * 64;
    static_assert(NIT == 11, "three row tiles for conv1 waves 0-2, two for wave 3");
    static_assert(PH * PW / 16 == 27, "7 / 7 / 7 / 6 requests per wave: the counted waits below");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];                    // 3 * PATCHB + 2 * INTERB = 128000 bytes
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int tid = threadIdx.x, lane = tid & 63, wv8 = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 31, lh = lane >> 5;
    const bool conv1 = wv8 < 4;                              // wave-uniform role
    const int wv = wv8 & 3;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    // conv1's bias, this lane's 4 x 4 channels, in registers (a global load in the epilogue would wait on the pending patch; from LDS the four reads
    // of a row tile each exposed their latency: the epilogue was 2 us of a 4.1 us iteration)
    float bq1[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        load4(b1 + g * 8 + lh * 4, bq1[g]);
#pragma unroll
        for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(bq1[g][e]));
    }
    u32x4 wf[NSTEP];
    {
        const bf16_t* w = conv1 ? w1 : w2;
#pragma unroll
        for (int s_ = 0; s_ < NSTEP; ++s_) wf[s_] = *reinterpret_cast<const u32x4*>(w + (long)lr * Kpad + s_ * 16 + lh * 8);
        // pin the fragments' arrival HERE: left alone, hipcc sinks their s_waitcnt vmcnt(0) to the first MFMA inside the tile loop, where it also waits
        // (every iteration) for the patch requests issued just before it -- the asm requests are invisible to its counter bookkeeping
#pragma unroll
        for (int s_ = 0; s_ < NSTEP; ++s_) asm volatile("" : "+v"(wf[s_]));
    }
    const int xcd = blockIdx.x & 7, gx = (int)gridDim.x >> 3, wx = (int)blockIdx.x >> 3;
    const int per = ntiles >> 3, rem = ntiles & 7;
    const int t_begin = xcd * per + min(xcd, rem), t_cnt = per + (xcd < rem ? 1 : 0);
    const int nloc = t_cnt > wx ? (t_cnt - wx + gx - 1) / gx : 0;       // this workgroup's tiles: t_begin + wx + j * gx
    if (nloc == 0) return;
    const int tpp = tiles_x * tiles_y;
    // conv1 geometry of this lane's (up to three) intermediate row tiles: pixel q = t * 32 + lr of the 10 x 34 grid (clamped past its end)
    int qoff[3], qy[3], qx[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int q = min((wv + 4 * k) * 32 + lr, NI - 1);
        qy[k] = q / IW; qx[k] = q - qy[k] * IW;
        qoff[k] = (qy[k] * PW + qx[k]) * 64;
    }
    // patch requests of tile j into patch[buf]: 16-byte slot = patch pixel * 4 + physical chunk; instruction q fills slots [64 q, + 64) = 16 pixels; a
    // lane outside the image writes zeros itself (fmb_kernel's scheme)
    // per-lane byte offset of each of this wave's (up to seven) requests relative to the patch's first pixel: a tile whose patch lies inside the image
    // (85 % of them at 512 x 512) takes one address add per request; the others the per-lane bounds checks below (59 instructions per request)
    int poff[(PH * PW / 16 + 3) / 4];
#pragma unroll
    for (int k = 0; k < (PH * PW / 16 + 3) / 4; ++k) {
        const int q = min(k * 4 + wv, PH * PW / 16 - 1);
        const int p0 = q * 16, r0 = p0 / PW, c0 = p0 - r0 * PW;
        int pc = c0 + (lane >> 2), prw = r0;
        if (pc >= PW) { pc -= PW; ++prw; }
        poff[k] = (prw * W + pc) * 64 + (((lane & 3) ^ ((pc >> 2) & 3)) << 4);
    }
    auto request_patch = [&](int j, int buf) -> int {    // returns this wave's request count on an interior tile, -1 on an edge tile (lanes outside the image skip theirs)
        const int bid = t_begin + wx + j * gx;
        const int b = bid / tpp, tr = bid - b * tpp;
        const int oy0 = (tr / tiles_x) * TH, ox0 = (tr % tiles_x) * TW;
        const unsigned char* img = reinterpret_cast<const unsigned char*>(in + (long)b * H * W * 32);
        // inline asm requests: behind __builtin_amdgcn_global_load_lds hipcc puts s_waitcnt vmcnt(0) in front of EVERY later LDS access of the wave (the
        // zero fills below, conv1's reads of the other buffer): the seven requests ran one after the other and the next tile's patch never travelled
        // under this tile's MFMAs (337 us). The waits that matter are explicit (det_head.h's HM_DMA)
        if (oy0 >= 2 && ox0 >= 2 && oy0 + TH + 2 <= H && ox0 + TW + 2 <= W) {          // (uniform)
            const unsigned char* org = img + ((long)(oy0 - 2) * W + (ox0 - 2)) * 64;
#pragma unroll
            for (int k = 0; k < (PH * PW / 16 + 3) / 4; ++k) {
                const int q = k * 4 + wv;
                if (q < PH * PW / 16) {
                    const void* g_ = org + poff[k];
                    const unsigned l_ = lds0 + (unsigned)(buf * PATCHB + q * 1024);
                    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g_), "s"(l_) : "memory", "m0");
                }
            }
            return wv < (PH * PW / 16) % 4 ? (PH * PW / 16 + 3) / 4 : PH * PW / 16 / 4;
        }
        unsigned char* patch = smem + buf * PATCHB;
        int ln = lane;
        asm volatile("" : "+v"(ln));                         // opaque per tile (see fmb_kernel)
        const int lp = ln >> 2, lc = ln & 3;
#pragma unroll
        for (int k = 0; k < (PH * PW / 16 + 3) / 4; ++k) {
            const int q = k * 4 + wv;
            if (q < PH * PW / 16) {
                const int p0 = q * 16, r0 = p0 / PW, c0 = p0 - r0 * PW;
                int pc = c0 + lp, prw = r0;
                if (pc >= PW) { pc -= PW; ++prw; }
                const int iy = oy0 - 2 + prw, ix = ox0 - 2 + pc;
                const bool inb = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                const unsigned off = (unsigned)(((iy * W + ix) * 32 + ((lc ^ ((pc >> 2) & 3)) << 3)) * 2);
                if (inb) {
                    const void* g_ = img + off;
                    const unsigned l_ = lds0 + (unsigned)(buf * PATCHB + q * 1024);
                    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g_), "s"(l_) : "memory", "m0");
                } else {
                    *reinterpret_cast<uint4*>(patch + q * 1024 + ln * 16) = make_uint4(0u, 0u, 0u, 0u);
                }
            }
        }
        return -1;
    };
    // THREE patch buffers, requests two tiles ahead: with one tile's patch (27 KB) in flight per CU the iteration could not be shorter than the loaded
    // memory round trip (ablation: no requests = -47 of 229 us). The end-of-iteration wait is counted: the newest tile's requests stay in flight
    if (conv1) {
        request_patch(0, 0);
        if (nloc > 1) request_patch(1, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    // one tile loop per role (the same number of barriers in each): in ONE loop hipcc's counter bookkeeping merged the conv2 waves' loads and stores into
    // the conv1 path and put s_waitcnt vmcnt(0) in front of conv1's first MFMA -- behind the patch requests just issued
    if (conv1) {
        int pb = 0;                                          // j % 3
        for (int j = 0; j <= nloc; ++j) {
            int nreq = 0;
            if (j < nloc) {
                if (j + 2 < nloc && !(SA_SR_ABL & 4)) nreq = request_patch(j + 2, pb == 0 ? 2 : pb - 1);
                const int bid = t_begin + wx + j * gx;
                const int b = bid / tpp, tr = bid - b * tpp;
                const int oy0 = (tr / tiles_x) * TH, ox0 = (tr % tiles_x) * TW;
                const unsigned char* patch = smem + pb * PATCHB;
                unsigned char* inter = smem + 3 * PATCHB + (j & 1) * INTERB;
                f32x16 acc[3];
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
                const bool three = wv < 3;                   // wave 3 owns row tiles 3 and 7 only
#pragma unroll
                for (int s_ = 0; s_ < ((SA_SR_ABL & 1) ? 1 : NSTEP); ++s_) {
                    const int tap = s_ >> 1, chunk = (s_ & 1) * 2 + lh, ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int csw = chunk ^ (((qx[k] + kx) >> 2) & 3);
                        const u32x4 xf = *reinterpret_cast<const u32x4*>(patch + qoff[k] + (ky * PW + kx) * 64 + csw * 16);
                        acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[s_]), __builtin_bit_cast(bf16x8, xf), acc[k], 0, 0, 0);
                    }
                }
                if (three) {
#pragma unroll
                    for (int s_ = 0; s_ < ((SA_SR_ABL & 1) ? 1 : NSTEP); ++s_) {
                        const int tap = s_ >> 1, chunk = (s_ & 1) * 2 + lh, ky = tap / 3, kx = tap - ky * 3;
                        const int csw = chunk ^ (((qx[2] + kx) >> 2) & 3);
                        const u32x4 xf = *reinterpret_cast<const u32x4*>(patch + qoff[2] + (ky * PW + kx) * 64 + csw * 16);
                        acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[s_]), __builtin_bit_cast(bf16x8, xf), acc[2], 0, 0, 0);
                    }
                }
                // lane = intermediate pixel (qy, qx), quad g = channels 8 g + 4 lh + (0..3); outside the image the intermediate tensor is zero
#pragma unroll
                for (int k = 0; k < ((SA_SR_ABL & 8) ? 1 : 3); ++k) {
                    if (k == 2 && !three) break;
                    const int iy = oy0 - 1 + qy[k], ix = ox0 - 1 + qx[k];
                    const unsigned m = ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) ? 0xffffffffu : 0u;
                    // (a lane past the grid's end repeats pixel NI - 1 -- same inputs, same value: its store is a harmless duplicate, no predicate)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const uint32_t p01 = pack2(hardswish_f(acc[k][4 * g] + bq1[g][0]), hardswish_f(acc[k][4 * g + 1] + bq1[g][1])) & m;
                        const uint32_t p23 = pack2(hardswish_f(acc[k][4 * g + 2] + bq1[g][2]), hardswish_f(acc[k][4 * g + 3] + bq1[g][3])) & m;
                        *reinterpret_cast<uint2*>(inter + (qy[k] * IW + qx[k]) * 64 + ((g ^ ((qx[k] >> 2) & 3)) << 4) + lh * 8) = make_uint2(p01, p23);
                    }
                }
            }
            // tile j + 1's patch has landed (this wave's share), tile j + 2's requests stay in flight. The builtin, so that hipcc's own bookkeeping sees
            // a clean state too (vmcnt(N) = 0x0F70 | N)
            if (nreq == 7) __builtin_amdgcn_s_waitcnt(0x0F77);
            else if (nreq == 6) __builtin_amdgcn_s_waitcnt(0x0F76);
            else __builtin_amdgcn_s_waitcnt(0x0F70);
            pb = pb == 2 ? 0 : pb + 1;
            __syncthreads();                                 // inter[j & 1] is complete, patch[(j + 1) % 3] has landed, inter[(j - 1) & 1] and patch[j % 3] are free
        }
    } else {
        // conv2's bias, this lane's two 8-channel groups, once (a load in the tile's epilogue waits vmcnt(0) behind the output store in front of it)
        uint4 b2raw[2];
#pragma unroll
        for (int p_ = 0; p_ < 2; ++p_) {
            b2raw[p_] = *reinterpret_cast<const uint4*>(b2 + 8 * (2 * p_ + lh));
            asm volatile("" : "+v"(b2raw[p_].x), "+v"(b2raw[p_].y), "+v"(b2raw[p_].z), "+v"(b2raw[p_].w));
        }
        for (int j = 0; j <= nloc; ++j) {
          if (j >= 1) {
            const int bid = t_begin + wx + (j - 1) * gx;
            const int b = bid / tpp, tr = bid - b * tpp;
            const int oy0 = (tr / tiles_x) * TH, ox0 = (tr % tiles_x) * TW;
            const unsigned char* inter = smem + 3 * PATCHB + ((j - 1) & 1) * INTERB;
            // residual rows of this tile (the block's input, L2-hot: the conv1 waves pulled it one iteration ago), requested before the MFMAs
            uint4 rraw[2][2];
            long obase[2];
            bool live[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int oy = oy0 + wv * 2 + i, ox = ox0 + lr;
                live[i] = oy < H && ox < W;
                obase[i] = (((long)b * H + min(oy, H - 1)) * W + min(ox, W - 1)) * 32;
#pragma unroll
                for (int p_ = 0; p_ < 2; ++p_) rraw[i][p_] = *reinterpret_cast<const uint4*>(in + obase[i] + 8 * (2 * p_ + lh));
            }
            f32x16 acc2[2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[i][r] = 0.f;
            // intermediate row wv * 2 + r feeds output row i at ky = r - i: each accumulator still sees ky = 0, 1, 2 with (kx, half) inside, stem_conv_kernel's order
#pragma unroll
            for (int r = 0; r < ((SA_SR_ABL & 2) ? 1 : 4); ++r) {
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        const int pcol = lr + kx, chunk = hf * 2 + lh;
                        const int csw = chunk ^ ((pcol >> 2) & 3);
                        const u32x4 xf = *reinterpret_cast<const u32x4*>(inter + ((wv * 2 + r) * IW + pcol) * 64 + csw * 16);
                        if (r <= 2)
                            acc2[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[(r * 3 + kx) * 2 + hf]), __builtin_bit_cast(bf16x8, xf), acc2[0], 0, 0, 0);
                        if (r >= 1)
                            acc2[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[((r - 1) * 3 + kx) * 2 + hf]), __builtin_bit_cast(bf16x8, xf), acc2[1], 0, 0, 0);
                    }
                }
            }
            // the residual is consumed HERE on every path: sunk under `if (live)`, a load not waited for on the dead path stays a pending writer of its
            // registers and the next iteration's first LDS read into them waits vmcnt(0) -- for this iteration's output stores too
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int p_ = 0; p_ < 2; ++p_) asm volatile("" : "+v"(rraw[i][p_].x), "+v"(rraw[i][p_].y), "+v"(rraw[i][p_].z), "+v"(rraw[i][p_].w));
#pragma unroll
            for (int i = 0; i < ((SA_SR_ABL & 16) ? 1 : 2); ++i) {
#pragma unroll
                for (int p_ = 0; p_ < ((SA_SR_ABL & 16) ? 1 : 2); ++p_) {
                    float v[8];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc2[i][8 * p_ + r]), __float_as_uint(acc2[i][8 * p_ + 4 + r]), false, false);
                        v[r] = __uint_as_float(sw[0]);
                        v[4 + r] = __uint_as_float(sw[1]);
                    }
                    const int c0 = 8 * (2 * p_ + lh);
                    float bv[8], r8[8];
                    unpack16(b2raw[p_], bv, (bf16_t*)nullptr);
                    unpack16(rraw[i][p_], r8, (bf16_t*)nullptr);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (v[e] + bv[e]) + r8[e];
                    if (live[i])
                        *reinterpret_cast<uint4*>(out + obase[i] + c0) = make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
                }
            }
        }
            __syncthreads();                                 // inter[j & 1] is complete, patch[(j + 1) & 1] has landed, inter[(j - 1) & 1] and patch[j & 1] are free
        }
        // hipcc lays the roles out as `if (!conv1) {...} if (conv1) {...}`: this role's residual loads (consumed under `if (live)`) would reach the conv1
        // loop as pending writers of its accumulator registers -- s_waitcnt vmcnt(0) in front of conv1's first MFMA, behind the patch requests
        __builtin_amdgcn_s_waitcnt(0x0F70);
    }
}

static inline int launch_stem_res(const bf16_t* in, const bf16_t* w1, const bf16_t* b1, const bf16_t* w2, const bf16_t* b2, bf16_t* out, int B, int H,
                                  int W, int Kpad, hipStream_t s) {
    if (!b1 || !b2 || (long)H * W * 64 >= (1L << 31) || (long)B * cdiv(W, 32) * cdiv(H, 8) >= (1L << 31)) return SA_ERR_UNSUPPORTED;   // (per-image byte offsets are 32-bit; find_fusions checks the same)
    const int tx = cdiv(W, 32), ty = cdiv(H, 8), ntiles = B * tx * ty;
    const unsigned g0 = (unsigned)std::min(ntiles, 256) / 8 * 8, grid = g0 ? g0 : 8u;      // persistent: one workgroup per CU, whole XCD rounds
    const size_t lds = 3 * (12 * 36 * 64) + 2 * (11 * 32 * 64);
    auto kern = stem_res_kernel;
    static AttrOnce attr;
    attr.ensure(kern, lds);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, in, w1, b1, w2, b2, out, H, W, Kpad, tx, ty, ntiles);
    return (int)hipGetLastError();
}
template <typename T>
static inline int launch_stem_res(const T*, const T*, const T*, const T*, const T*, T*, int, int, int, int, hipStream_t) { return SA_ERR_UNSUPPORTED; }

// The first convolution straight from the caller's pixels (SRC 1 / 2 above)
static inline int launch_stem_conv_pixels(const StemSrc& src, const bf16_t* w, const bf16_t* bias, bf16_t* out, int B, int H, int W, int Ho, int Wo,
                                          int Cout, int k, int stride, int pad, int Kpad, int act, hipStream_t s) {
    if (Cout != 32 || k != 3 || pad != 1 || !bias || stride != 2 || act != ACT_HSWISH || (!src.planes && !src.u8)) return SA_ERR_UNSUPPORTED;
    const int tx = cdiv(Wo, 32), ty = cdiv(Ho, 8), ntiles = B * tx * ty;
    const unsigned g0 = (unsigned)std::min(ntiles, 256 * 3) / 8 * 8, grid = g0 ? g0 : 8u;
    if (src.planes)
        hipLaunchKernelGGL((stem_conv_kernel<8, 2, 0, 1>), dim3(grid), dim3(256), 0, s, (const bf16_t*)nullptr, w, bias, (const bf16_t*)nullptr, out, H, W, Ho, Wo, Kpad, tx, ty, ntiles, src);
    else
        hipLaunchKernelGGL((stem_conv_kernel<8, 2, 0, 2>), dim3(grid), dim3(256), 0, s, (const bf16_t*)nullptr, w, bias, (const bf16_t*)nullptr, out, H, W, Ho, Wo, Kpad, tx, ty, ntiles, src);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// FusedMBConv (encoderdecoder.py:228-270): 3x3 expand (+ folded BN + Hardswish) -> 1x1 projection (+ folded BN, + residual) in ONE kernel: the
// expanded tensor (256 ... 1024 channels at 256^2 / 128^2, 0.27 - 1.07 GB per 16 pages) never exists. Attention-shaped, as VERDICT r05 drew it:
//   workgroup = an 8 x 32 pixel tile (8 MFMA row tiles), 4 waves, wave w = pixel rows 2 w, 2 w + 1; persistent over the tile raster;
//   the tile's input patch ((8 - 1) S + 3 rows x (32 - 1) S + 3 columns x Cin, zero outside the image) sits in LDS for the whole tile: the
//     3x3's A operand is a 16-byte LDS read at (pixel + tap) -- the stem kernel's im2col without the copy;
//   the mid channels are walked in chunks of 128: S[64 px x 128] per wave (8 accumulators, 128 registers) = patch . W1_chunk over K = 9 Cin, the
//     chunk's weights streamed through a two-buffer LDS ring in 64-wide K-tiles (global_load_lds, gemm.h's swizzle; every wave reads all of a
//     K-tile: 2 + 4 fragment reads per 8 MFMAs);
//   at the end of a chunk: + bias, Hardswish, round to bf16 (the op list's rounding point), v_permlane32_swap turns the accumulator layout
//     (4 consecutive channels per lane) into the A-operand layout (8 consecutive), and O[64 px x Cout] += S . W2_chunk with O in accumulators and
//     the W2 fragments from L2 (requested at the top of the chunk; W2 comes fragment-major, [MID / 64][2 cout tiles][4 K steps][64 lanes][8], made at
//     engine init by det_mbconv.h's mbconv_w2_fragments: one contiguous KiB per load);
//   epilogue: + bias, round (the GEMM's rounding), + residual, round; 16-byte stores.
// K order of the 3x3 = the implicit GEMM's ((ky, kx, ci) ascending, 16 per MFMA, one accumulator per output); the projection sums K = mid in the
// same 16-element steps as the GEMM: with identical S bits the sums are the GEMM's. Built for Cout = 64 (stage 0: the two largest expanded
// tensors); the Cout = 128 blocks of stage 1 keep the op list (their W2 fragments + 128 more accumulator registers do not fit beside S).
// Build-time A/B switches of fmb_kernel (tools/microbench/fmb_variants.sh; gpurun r06ab, op 4 = the Cin 32 stride-2 block, op 6 = the Cin 64 block):
//   SA_FMB_PK    chunk epilogue on fp32 pairs:        op 4 621 -> 579 us, op 6 374 -> 384 (256-register variant: more spills)  => on at MCH = 128 only
//   SA_FMB_NB3   three ring buffers at MCH = 128:     op 4 621 -> 639                                                             => off
//   SA_FMB_OVL   next tile's patch under the tail:    op 4 621 -> 631                                                             => off
// (the timing ablations that pointed at the ring and the patch -- 33 and 65 us of op 4 -- measured the BURST of 256 synchronised workgroups
// asking for 18 MB at once, not a latency a deeper ring or a 3 us earlier request could hide)
#ifndef SA_FMB_NB3
#define SA_FMB_NB3 0
#endif
#ifndef SA_FMB_OVL
#define SA_FMB_OVL 0
#endif
#ifndef SA_FMB_PK
#define SA_FMB_PK 1
#endif
#ifndef SA_FMB_LATEW
#define SA_FMB_LATEW 1
#endif
#ifndef SA_FMB_ABL
#define SA_FMB_ABL 0                     // timing ablations (tools/microbench/fmb_ablate.sh; results are wrong with any bit set): 1 chunk epilogue without
#endif                                   // bias + Hardswish, 2 no W1 ring requests after a tile's first, 4 no patch requests, 8 one K stage per chunk
template <int CIN, int S, int MCH>      // MCH = mid channels per chunk: 128 (one workgroup per CU) or 64 (half the accumulators: two workgroups per CU,
                                        // one multiplying while the other runs its chunk epilogue on the vector ALU)
__global__ __launch_bounds__(256, MCH == 64 ? 2 : 1) void fmb_kernel(const bf16_t* __restrict__ in, const bf16_t* __restrict__ w1, const bf16_t* __restrict__ b1,
                                                 const bf16_t* __restrict__ w2, const bf16_t* __restrict__ b2, const bf16_t* __restrict__ res,
                                                 bf16_t* __restrict__ out, const bf16_t* __restrict__ zero, int H, int W, int Ho, int Wo, int MID,
                                                 int Kpad, int tiles_x, int tiles_y, int ntiles) {
    constexpr int COUT = 64, TH = 8, TW = 32, PH = (TH - 1) * S + 3, PW = (TW - 1) * S + 3, PB = CIN * 2, CPP = CIN / 8;
    constexpr int SH = CPP == 4 ? 2 : (CPP == 8 ? 1 : 0), KS_TAP = CIN / 16, NKT = (9 * CIN + 63) / 64;
    constexpr int KPS = MCH == 64 ? 2 : 1;                                  // K-tiles per ring stage (one barrier per stage: at MCH = 64 a single K-tile is only 16 MFMAs per wave)
    constexpr int NST = (NKT + KPS - 1) / KPS;                               // stages per chunk
    constexpr int PATCH = (PH * PW * PB + 1023) & ~1023, RB = KPS * MCH * 128;     // bytes: the patch; one ring buffer = KPS K-tiles of MCH W1 rows x 64 k
    // One workgroup per CU (MCH = 128): nothing else hides a stall, so the ring has THREE buffers (a stage's requests are issued two stages
    // = ~2000 cycles ahead; one stage ahead left 33 us of op 4 waiting on L2, SA_FMB_ABL = 2) and the NEXT tile's patch and first two stages are
    // requested behind the tile's last K stage, under its last chunk epilogue and output epilogue (patch round trip: 65 us of op 4, SA_FMB_ABL = 4).
    // Two workgroups per CU (MCH = 64): two buffers (LDS), the other workgroup covers.
    constexpr int NB = (MCH == 128 && SA_FMB_NB3) ? 3 : 2, PD = NB - 1;
    constexpr bool OVL = MCH == 128 && SA_FMB_OVL;
    constexpr int NJ = MCH / 32, NI = MCH / 32, NS2 = MCH / 16;             // mid tiles per chunk; request instructions per wave and K-tile; projection K steps per chunk
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* ring = smem + PATCH;
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 31, lh = lane >> 5;
    const int nchunks = MID / MCH;
    // request side of the W1 ring: instruction i of wave wv fills rows (wv * 4 + i) * 8 + (lane >> 3) of a K-tile, physical chunk lane & 7
    unsigned wq[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int row = (wv * NI + i) * 8 + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
        wq[i] = (unsigned)(row * Kpad * 2 + c * 16);
    }
#define FM_ISSUE(BUF, CH, ST)                                                                                           \
    {                                                                                                                   \
        _Pragma("unroll") for (int t_ = 0; t_ < KPS; ++t_) {                                                            \
            const int kt_ = min((ST) * KPS + t_, NKT - 1);          /* a stage past the last K-tile re-reads it (never multiplied) */ \
            const unsigned char* b_ = reinterpret_cast<const unsigned char*>(w1) + ((long)(CH) * MCH * Kpad + kt_ * 64) * 2; \
            _Pragma("unroll") for (int i_ = 0; i_ < NI; ++i_)                                                           \
                __builtin_amdgcn_global_load_lds((gptr_t)(b_ + wq[i_]), (lptr_t)(ring + (BUF) * RB + t_ * MCH * 128 + (wv * NI + i_) * 1024), 16, 0, 0); \
        }                                                                                                               \
    }
    // patch requests of tile (BB, OY0, OX0): direct to LDS, all in flight at once. 16-byte slot = pixel * CPP + physical chunk, instruction q fills
    // slots [64 q, + 64) = 64 / CPP pixels (they wrap over at most one patch row end: no division); a lane outside the image writes zeros itself.
    // The first version staged the patch through registers in a rolled loop -- one global round trip per 16 bytes and thread, 17 in a row at
    // S = 2: ~25 of a tile's 45 us (ISA: load, vmcnt(0), ds_write).
#define FM_PATCH(BB, OY0, OX0)                                                                                          \
    {                                                                                                                   \
        const unsigned char* img_ = reinterpret_cast<const unsigned char*>(in + (long)(BB) * H * W * CIN);              \
        const int iy0_ = (OY0) * S - 1, ix0_ = (OX0) * S - 1;                                                           \
        int ln_ = lane;                                                                                                 \
        asm volatile("" : "+v"(ln_));       /* opaque per tile: visible, hipcc hoists every slot's (row, column) out of the tile loop and spills */ \
        const int lp_ = ln_ / CPP, lc_ = ln_ % CPP;                                                                     \
        _Pragma("unroll") for (int k_ = 0; k_ < (PATCH / 1024 + 3) / 4; ++k_) {                                         \
            const int q_ = k_ * 4 + wv;                              /* wave-uniform */                               \
            if (q_ < PATCH / 1024 && !(SA_FMB_ABL & 4)) {                                                               \
                const int p0_ = q_ * (64 / CPP), r0_ = p0_ / PW, c0_ = p0_ - r0_ * PW;     /* scalar */                 \
                int pc_ = c0_ + lp_;                                                                                    \
                const bool wrap_ = pc_ >= PW;                                                                           \
                pc_ -= wrap_ ? PW : 0;                                                                                  \
                const int iy_ = iy0_ + r0_ + (wrap_ ? 1 : 0), ix_ = ix0_ + pc_;                                         \
                const bool inb_ = p0_ + lp_ < PH * PW && (unsigned)iy_ < (unsigned)H && (unsigned)ix_ < (unsigned)W;    \
                const unsigned off_ = (unsigned)(((iy_ * W + ix_) * CIN + ((lc_ ^ ((pc_ >> SH) & (CPP - 1))) << 3)) * 2); \
                if (inb_) __builtin_amdgcn_global_load_lds((gptr_t)(img_ + off_), (lptr_t)(smem + q_ * 1024), 16, 0, 0); \
                else *reinterpret_cast<uint4*>(smem + q_ * 1024 + ln_ * 16) = make_uint4(0u, 0u, 0u, 0u);               \
            }                                                                                                           \
        }                                                                                                               \
    }
    // a tile's first PD ring stages + its patch
#define FM_HEAD(BB, OY0, OX0)                                                                                           \
    {                                                                                                                   \
        _Pragma("unroll") for (int s_ = 0; s_ < PD; ++s_) FM_ISSUE(s_, s_ / NST, s_ % NST);                             \
        FM_PATCH(BB, OY0, OX0);                                                                                         \
    }
    // read side
    int xb[2], swz[3];
#pragma unroll
    for (int i = 0; i < 2; ++i) xb[i] = (((wv * 2 + i) * S) * PW + lr * S) * PB;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) swz[kx] = ((lr * S + kx) >> SH) & (CPP - 1);
    int wl[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) wl[kk] = lr * 128 + (((kk * 2 + lh) ^ ((lr >> 1) & 7)) << 4);

    const int xcd = blockIdx.x & 7, gx = (int)gridDim.x >> 3, wx = (int)blockIdx.x >> 3;
    const int per = ntiles >> 3, rem = ntiles & 7;
    const int t_begin = xcd * per + min(xcd, rem), t_cnt = per + (xcd < rem ? 1 : 0);
    for (int tl = wx; tl < t_cnt; tl += gx) {
        const int bid = t_begin + tl;
        const int b = bid / (tiles_x * tiles_y), tr = bid - b * tiles_x * tiles_y;
        const int oy0 = (tr / tiles_x) * TH, ox0 = (tr % tiles_x) * TW;
        if (!OVL || tl == wx) {
            __syncthreads();                                 // the previous tile's patch and ring are done with
            FM_HEAD(b, oy0, ox0);
        }
        f32x16 oacc[2][2];                                   // [cout tile][row]
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[j][i][r] = 0.f;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                     // patch + K-tile (chunk 0, 0) landed
        int fb = 0;                                          // buffer of the chunk's stage 0
        for (int ch = 0; ch < nchunks; ++ch) {
            // W2 fragments of this chunk (K = its MCH mid channels, NS2 steps x 2 cout tiles) and its bias: requested at the top of the chunk and used
            // behind the K loop (MCH = 128, one workgroup per CU), or requested behind the K loop (MCH = 64 with SA_FMB_LATEW: 48 registers fewer
            // across the loop -- at 256 registers the early request spilled 38 -- and the second workgroup covers the round trip)
            u32x4 w2f[2][NS2];
            uint2 b1r[NJ][4];
#define FM_LOADW()                                                                                                      \
    {                                                                                                                   \
        _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                                                \
            _Pragma("unroll") for (int s2_ = 0; s2_ < NS2; ++s2_)                                                       \
                w2f[j_][s2_] = *reinterpret_cast<const u32x4*>(w2 + ((long)((ch * MCH >> 6) + (s2_ >> 2)) * 2 + j_) * 2048 + (s2_ & 3) * 512 + lane * 8); \
        _Pragma("unroll") for (int j_ = 0; j_ < NJ; ++j_)                                                               \
            _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) b1r[j_][g_] = *reinterpret_cast<const uint2*>(b1 + ch * MCH + j_ * 32 + g_ * 8 + lh * 4); \
    }
            constexpr bool LATEW = MCH == 64 && SA_FMB_LATEW;
            if (!LATEW) FM_LOADW();
            f32x16 sacc[NJ][2];                              // [mid tile][row]
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sacc[j][i][r] = 0.f;
#pragma unroll
            for (int st = 0; st < NST; ++st) {
                // stage st of this chunk sits in buffer (ch * NST + st) % NB; the one PD stages on in the tile's flat sequence is requested now,
                // into the buffer every wave left at the last barrier
                const int buf = (fb + st) % NB;
                const bool more = st + PD < NST || ch + 1 < nchunks;     // uniform
                if (more && !(SA_FMB_ABL & 2)) FM_ISSUE((fb + st + PD) % NB, st + PD < NST ? ch : ch + 1, (st + PD) % NST);
                if ((SA_FMB_ABL & 8) && st > 0) continue;
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < KPS; ++t) {
                const int kt = st * KPS + t;
                if (kt >= NKT) break;
                // fragments of K step kk + 1 are read while step kk multiplies (two named register sets, one LDS read placed between each
                // pair of MFMAs by sched_group_barrier: left to hipcc the stream was read x 6, s_waitcnt lgkmcnt(0), MFMA x 4, read x 2, wait,
                // MFMA x 4 -- every LDS round trip exposed, the matrix pipe half idle inside the loop)
#define FM_R(XF, WF, KK)                                                                                                \
    {                                                                                                                   \
        const int s_ = kt * 4 + (KK), tap_ = s_ / KS_TAP < 9 ? s_ / KS_TAP : 8, c16_ = s_ % KS_TAP;                     \
        const int ky_ = tap_ / 3, kx_ = tap_ - ky_ * 3;                                                                 \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                                                                \
            XF[i_] = *reinterpret_cast<const u32x4*>(smem + xb[i_] + (ky_ * PW + kx_) * PB + (((c16_ * 2 + lh) ^ swz[kx_]) << 4)); \
        _Pragma("unroll") for (int j_ = 0; j_ < NJ; ++j_) WF[j_] = *reinterpret_cast<const u32x4*>(ring + buf * RB + t * MCH * 128 + j_ * 4096 + wl[KK]); \
    }
#define FM_M(XF, WF)                                                                                                    \
    {                                                                                                                   \
        _Pragma("unroll") for (int j_ = 0; j_ < NJ; ++j_)                                                               \
            _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                                                            \
                sacc[j_][i_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, WF[j_]), __builtin_bit_cast(bf16x8, XF[i_]), sacc[j_][i_], 0, 0, 0); \
    }
#define FM_SGB()                                                                                                        \
    {                                                                                                                   \
        _Pragma("unroll") for (int q_ = 0; q_ < 2 + NJ; ++q_) {                                                         \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                          \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                          \
        }                                                                                                               \
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * NJ - 2 - NJ, 0);                                                \
    }
                {
                    u32x4 xfa[2], wfa[NJ], xfb[2], wfb[NJ];
                    FM_R(xfa, wfa, 0);
                    FM_R(xfb, wfb, 1);
                    FM_M(xfa, wfa);
                    FM_R(xfa, wfa, 2);
                    FM_M(xfb, wfb);
                    FM_R(xfb, wfb, 3);
                    FM_M(xfa, wfa);
                    FM_M(xfb, wfb);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2 + NJ, 0);
                    FM_SGB();
                    FM_SGB();
                    FM_SGB();
                    __builtin_amdgcn_sched_group_barrier(0x008, 2 * NJ, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                }
#undef FM_R
#undef FM_M
#undef FM_SGB
                __builtin_amdgcn_sched_barrier(0);
                // the next stage landed (everything but the requests just issued -- in-order return among loads; no store is in flight here)
                if (NB == 3 && more && !(SA_FMB_ABL & 2)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KPS * NI) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();                             // ... and every wave is done with this one
            }
            fb = (fb + NST) % NB;
            if (LATEW) FM_LOADW();
#undef FM_LOADW
            if (OVL && ch + 1 == nchunks && tl + gx < t_cnt) {   // the next tile's patch and first stages travel under this chunk's epilogue and the stores
                const int bidn = t_begin + tl + gx;
                const int bn = bidn / (tiles_x * tiles_y), trn = bidn - bn * tiles_x * tiles_y;
                FM_HEAD(bn, (trn / tiles_x) * TH, (trn % tiles_x) * TW);
            }
            // ---- S -> bias, Hardswish, bf16, A-operand layout; O += S . W2_chunk
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    uint32_t pk[4][2];                       // quad g: its 4 channels as two packed pairs
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        // on fp32 PAIRS: v_pk_add_f32 (bias), then common.h's hardswish_pk (v_pk_fma_f32 ... clamp, v_pk_mul_f32) -- the scalar literal
                        // form was ~8 vector instructions per value at one wave per SIMD (107 of op 4's 589 us, SA_FMB_ABL = 1). Same operations as
                        // hardswish_f(s + b), every step one fp32 rounding.
                        float bq[4];
                        load4(reinterpret_cast<const bf16_t*>(&b1r[j][g]), bq);
#if SA_FMB_PK
                      if (MCH == 128) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            f32x2 x = f32x2{sacc[j][i][4 * g + 2 * h], sacc[j][i][4 * g + 2 * h + 1]};
                            if (!(SA_FMB_ABL & 1)) {
                                x = hardswish_pk(x + f32x2{bq[2 * h], bq[2 * h + 1]});
                            }
                            pk[g][h] = __builtin_bit_cast(uint32_t, __builtin_convertvector(x, bf16x2_t));
                        }
                      } else
#endif
                      {
                        float v[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = (SA_FMB_ABL & 1) ? sacc[j][i][4 * g + r] : hardswish_f(sacc[j][i][4 * g + r] + bq[r]);
                        pk[g][0] = pack2(v[0], v[1]); pk[g][1] = pack2(v[2], v[3]);
                      }
                    }
#pragma unroll
                    for (int p_ = 0; p_ < 2; ++p_) {
                        // quads 2 p and 2 p + 1: lane l (lh = 0) ends with channels [16 p, + 8), lane l + 32 with [16 p + 8, + 8) of this mid tile
                        const auto s0 = __builtin_amdgcn_permlane32_swap(pk[2 * p_][0], pk[2 * p_ + 1][0], false, false);
                        const auto s1 = __builtin_amdgcn_permlane32_swap(pk[2 * p_][1], pk[2 * p_ + 1][1], false, false);
                        const u32x4 x2 = {s0[0], s1[0], s0[1], s1[1]};
                        const int s2 = j * 2 + p_;
#pragma unroll
                        for (int j2 = 0; j2 < 2; ++j2)
                            oacc[j2][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w2f[j2][s2]), __builtin_bit_cast(bf16x8, x2), oacc[j2][i], 0, 0, 0);
                    }
                }
        }
        // ---- output: + bias, round; + residual, round (the projection GEMM's epilogue); lane ends with channels [8 (2 p + lh), + 8) of a cout tile
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int oy = oy0 + wv * 2 + i, ox = ox0 + lr;
            const bool live = oy < Ho && ox < Wo;
            const long obase = (((long)b * Ho + min(oy, Ho - 1)) * Wo + min(ox, Wo - 1)) * COUT;
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
                for (int p_ = 0; p_ < 2; ++p_) {
                    float v[8];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(oacc[j2][i][8 * p_ + r]), __float_as_uint(oacc[j2][i][8 * p_ + 4 + r]), false, false);
                        v[r] = __uint_as_float(sw[0]);
                        v[4 + r] = __uint_as_float(sw[1]);
                    }
                    const int c0 = j2 * 32 + 8 * (2 * p_ + lh);
                    float bv[8];
                    unpack16(*reinterpret_cast<const uint4*>(b2 + c0), bv, (bf16_t*)nullptr);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = Ty<bf16_t>::rnd(v[e] + bv[e]);
                    if (res) {
                        float r8[8];
                        unpack16(*reinterpret_cast<const uint4*>(res + obase + c0), r8, (bf16_t*)nullptr);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += r8[e];
                    }
                    if (live)
                        *reinterpret_cast<uint4*>(out + obase + c0) = make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
                }
        }
    }
#undef FM_ISSUE
#undef FM_PATCH
#undef FM_HEAD
}

static inline bool fmb_shape_ok(int cin, int mid, int cout, int stride, int ho, int wo) {
    (void)ho; (void)wo;
    return cout == 64 && mid % 128 == 0 && ((cin == 64 && stride == 1) || (cin == 32 && stride == 2));
}

static inline int launch_fmb(const bf16_t* in, const bf16_t* w1, const bf16_t* b1, const bf16_t* w2, const bf16_t* b2, const bf16_t* res, bf16_t* out,
                             const bf16_t* zero, int B, int H, int W, int Cin, int Ho, int Wo, int Mid, int Cout, int stride, int pad, int Kpad,
                             hipStream_t s) {
    if (!fmb_shape_ok(Cin, Mid, Cout, stride, Ho, Wo) || pad != 1 || !b1 || !b2) return SA_ERR_SHAPE;
    const int tx = cdiv(Wo, 32), ty = cdiv(Ho, 8), ntiles = B * tx * ty;
#define SA_FMB(CI, SS, MC, WGS)                                                                                                 \
    {                                                                                                                           \
        constexpr int PH_ = 7 * (SS) + 3, PW_ = 31 * (SS) + 3;                                                                  \
        constexpr size_t lds = (size_t)((PH_ * PW_ * (CI) * 2 + 1023) & ~1023) + ((MC) == 64 ? 2 * 2 : (SA_FMB_NB3 ? 3 : 2)) * (MC) * 128; \
        auto kern = fmb_kernel<CI, SS, MC>;                                                                                     \
        static AttrOnce attr;                                                                                                   \
        attr.ensure(kern, lds);                                                                                                 \
        const unsigned g_ = (unsigned)std::max(8, std::min(ntiles, 256 * (WGS)) / 8 * 8);                                       \
        hipLaunchKernelGGL(kern, dim3(g_), dim3(256), lds, s, in, w1, b1, w2, b2, res, out, zero, H, W, Ho, Wo, Mid, Kpad, tx, ty, ntiles); \
    }
    if (Cin == 64) { if (tuning().fmb_chunk == 64) SA_FMB(64, 1, 64, 2) else SA_FMB(64, 1, 128, 1) }
    else SA_FMB(32, 2, 128, 1)
#undef SA_FMB
    return (int)hipGetLastError();
}
template <typename T>
static inline int launch_fmb(const T*, const T*, const T*, const T*, const T*, const T*, T*, const T*, int, int, int, int, int, int, int, int, int, int,
                             int, hipStream_t) { return SA_ERR_UNSUPPORTED; }

}  // namespace sa
