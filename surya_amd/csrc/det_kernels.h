// Kernels of the text-detection forward pass (EfficientViT-L backbone + SegFormer-style head,
// surya/detection/model/encoderdecoder.py) for gfx950. Activations are NHWC so that a conv's K axis
// (ky, kx, ci) is contiguous per tap: dense convs run as implicit GEMMs on the MFMA tiles of gemm.h with the
// A-tile gathered on the fly (no im2col buffer); BatchNorm is folded into weights/bias at load time.
#pragma once
#include "gemm.h"

namespace sa {

enum DetAct { ACT_NONE = 0, ACT_HSWISH = 1, ACT_RELU = 2 };

// ---------------------------------------------------------------------------------------------------
// pixel_values fp32 NCHW [B,3,H,W] -> NHWC with channels padded to CP (zeros), storage type T.
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ in, T* __restrict__ out, int B, int C, int H, int W, int CP) {
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;      // pixel index over B*H*W
    if (p >= (long)B * H * W) return;
    const long hw = (long)H * W;
    const long b = p / hw, r = p % hw;
    T* o = out + p * CP;
    if constexpr (sizeof(T) == 2) {
        if (CP == 8 && C <= 8) {
            // the shipped configuration (3 -> 8 channels of bf16): ONE 16-byte store per pixel instead of eight 2-byte stores (192 us per
            // 16 pages of 1024^2, 2.4 TB/s on a pure layout change); same conversions, same bits
            float v[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = c < C ? in[(b * C + c) * hw + r] : 0.f;
            *reinterpret_cast<uint4*>(o) = make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
            return;
        }
    }
    for (int c = 0; c < CP; ++c) Ty<T>::st(o + c, c < C ? in[(b * C + c) * hw + r] : 0.f);
}

// uint8 RGB pages (NHWC, as PIL hands them over) -> the first activation buffer: SegformerImageProcessor's rescale and
// normalise (surya/detection/processor.py:126-146: x * (1/255) in fp32, (x - mean) / std) fused into the layout change, so a
// page crosses PCIe as 3 bytes per pixel instead of 12 and the host never touches its pixels (det side).
template <typename T>
__global__ void u8_to_nhwc_kernel(const unsigned char* __restrict__ in, T* __restrict__ out, long P, int CP, float m0, float m1,
                                  float m2, float s0, float s1, float s2, int pix) {
#pragma clang fp contract(off)
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;      // pixel index over B*H*W
    if (p >= P) return;
    const unsigned char* px = in + p * pix;                           // 3 = RGB, 4 = RGBX (PIL's in-memory layout)
    // the reference's rescale: float64 product rounded to float32 once (transformers image_transforms.rescale), then the float32
    // normalisation -- px * float32(1 / 255) is one ulp off for some pixel values
    const double k = 1.0 / 255.0;
    T* o = out + p * CP;
    const float v0 = ((float)((double)px[0] * k) - m0) / s0, v1 = ((float)((double)px[1] * k) - m1) / s1, v2 = ((float)((double)px[2] * k) - m2) / s2;
    if constexpr (sizeof(T) == 2) {
        if (CP == 8) {                                       // one 16-byte store per pixel (see nchw_to_nhwc_kernel)
            *reinterpret_cast<uint4*>(o) = make_uint4(pack2(v0, v1), pack2(v2, 0.f), 0u, 0u);
            return;
        }
    }
    Ty<T>::st(o + 0, v0);
    Ty<T>::st(o + 1, v1);
    Ty<T>::st(o + 2, v2);
    for (int c = 3; c < CP; ++c) Ty<T>::st(o + c, 0.f);
}

// ---------------------------------------------------------------------------------------------------
// Implicit-GEMM convolution: out[m, n] = act(sum_{ky,kx,ci} in[b, oy*s+ky-p, ox*s+kx-p, ci] * w[n, (ky,kx,ci)] + bias[n]) (+ res)
//   m = (b, oy, ox) output pixel, weights [Cout][KH*KW*Cin padded to a multiple of the K-tile] (zero padded).
// Same tile machinery as gemm_nt_kernel (2-deep register prefetch, swizzled LDS, weight fragment first); only the
// A-operand fetch differs: a 16-byte chunk = 8 (bf16) / 4 (fp32) consecutive input channels of one tap, zero outside.
// x / d in conv_gemm_kernel: by the host-made reciprocal, or (A/B builds, -DSA_CONV_TRUE_DIV=1) by the division it replaced
#ifndef SA_CONV_TRUE_DIV
#define SA_CONV_TRUE_DIV 0
#endif
__device__ __forceinline__ int conv_div(int x, int d, FastDiv f) {
#if SA_CONV_TRUE_DIV
    (void)f; return x / d;
#else
    (void)d; return (int)fast_div((unsigned)x, f);
#endif
}

template <typename T>
struct ConvArgs {
    const T* in; const T* w; T* out; const T* bias; const T* res;
    int B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, Kpad, act;
    const T* zero = nullptr;        // >= 16 zero bytes on the device: what the direct-to-LDS gather reads outside the image
    FastDiv fcin, fkw, fhw, fwo;    // reciprocals of Cin, KW, Ho * Wo, Wo (launch_conv sets them): conv_gemm_kernel divides per staged chunk
};

// CIN64: Cin is a multiple of the K-tile (64 bf16 / 32 fp32 elements), so a K-tile never straddles two filter taps: the
// tap and its (ky, kx) are scalars computed once per K-tile instead of two integer divisions per staged chunk.
template <typename T, int BM, int BN, int WM, int WN, bool CIN64 = false>
__global__ __launch_bounds__(64 * WM * WN) void conv_gemm_kernel(ConvArgs<T> p) {
    constexpr int NT = 64 * WM * WN;
    constexpr int KE = Ty<T>::KE, V = Ty<T>::V16;
    constexpr int WTM = BM / WM, WTN = BN / WN, FM = WTM / 32, FN = WTN / 32;
    constexpr int XCH = BM * 8 / NT, WCH = BN * 8 / NT;
    static_assert(XCH >= 1 && WCH >= 1 && (BM * 8) % NT == 0 && (BN * 8) % NT == 0, "tile");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int XBYTES = BM * 128, BUF = XBYTES + BN * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int M = p.B * p.Ho * p.Wo;
    const int tiles_n = (p.Cout + BN - 1) / BN;
    // XCD x (= blockIdx.x % 8) takes the x-th contiguous eighth of the tile raster, so that output rows sharing input rows sit in one
    // XCD's L2 (see dwconv_tx_kernel: FETCH_SIZE of the stem convolutions was 2.2x their input in launch order)
    const unsigned nb_ = gridDim.x, xcd_ = blockIdx.x & 7, per_ = nb_ >> 3, rem_ = nb_ & 7;
    const int bid = (int)(xcd_ * per_ + min(xcd_, rem_) + (blockIdx.x >> 3));
    const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;
    const int ntaps = p.KH * p.KW;

    // per-chunk constants: output pixel of the row, chunk position inside the 128-byte K-row
    int xb[XCH], xiy[XCH], xix[XCH], xc[XCH], xdst[XCH];
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
        const int id = tid + i * NT, row = id >> 3, c = id & 7;
        const int m = min(m0 + row, M - 1);
        const int b = conv_div(m, p.Ho * p.Wo, p.fhw), r = m - b * (p.Ho * p.Wo), oy_ = conv_div(r, p.Wo, p.fwo);
        xb[i] = b; xiy[i] = oy_ * p.stride - p.pad; xix[i] = (r - oy_ * p.Wo) * p.stride - p.pad;
        xc[i] = c * V;
        xdst[i] = row * 128 + ((c ^ ((row >> 1) & 7)) << 4);
    }
    const unsigned char* wsrc[WCH];
    int wdst[WCH];
#pragma unroll
    for (int i = 0; i < WCH; ++i) {
        const int id = tid + i * NT, row = id >> 3, c = id & 7;
        const int gr = min(n0 + row, p.Cout - 1);
        wsrc[i] = reinterpret_cast<const unsigned char*>(p.w + (long)gr * p.Kpad) + c * 16;
        wdst[i] = XBYTES + row * 128 + ((c ^ ((row >> 1) & 7)) << 4);
    }
    f32x16 acc[FN][FM];
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;
    const int nk = p.Kpad / KE;
    u32x4 xr0[XCH], wr0[WCH], xr1[XCH], wr1[WCH];
#define SA_CFETCH(XR, WR, KT)                                                                                   \
    {                                                                                                           \
        const int kbase_ = (KT) * KE;                                                                           \
        int tap_u = 0, ci_u = 0, ky_u = 0, kx_u = 0;                                                            \
        if constexpr (CIN64) {                                                                                  \
            tap_u = conv_div(kbase_, p.Cin, p.fcin); ci_u = kbase_ - tap_u * p.Cin;                             \
            ky_u = conv_div(tap_u, p.KW, p.fkw); kx_u = tap_u - ky_u * p.KW;                                   \
        }                                                                                                       \
        _Pragma("unroll") for (int i = 0; i < XCH; ++i) {                                                       \
            int tap, ci, ky, kx;                                                                                \
            if constexpr (CIN64) {                                                                              \
                tap = tap_u; ci = ci_u + xc[i]; ky = ky_u; kx = kx_u;                                           \
            } else {                                                                                            \
                const int k0 = kbase_ + xc[i];                                                                  \
                tap = conv_div(k0, p.Cin, p.fcin); ci = k0 - tap * p.Cin;              /* (round 5: reciprocals -- two integer */ \
                ky = conv_div(tap, p.KW, p.fkw); kx = tap - ky * p.KW;               /* divisions per chunk and K-tile were ~80 VALU) */ \
            }                                                                                                   \
            const int iy = xiy[i] + ky, ix = xix[i] + kx;                                                       \
            const bool ok = (tap < ntaps) & (iy >= 0) & (iy < p.H) & (ix >= 0) & (ix < p.W);                    \
            const long off = ok ? ((((long)xb[i] * p.H + iy) * p.W + ix) * p.Cin + ci) : 0;                     \
            u32x4 v_ = *reinterpret_cast<const u32x4*>(p.in + off);                                             \
            const unsigned int msk_ = ok ? 0xffffffffu : 0u;                                                    \
            v_[0] &= msk_; v_[1] &= msk_; v_[2] &= msk_; v_[3] &= msk_;                                         \
            XR[i] = v_;                                                                                         \
        }                                                                                                       \
        _Pragma("unroll") for (int i = 0; i < WCH; ++i)                                                         \
            WR[i] = *reinterpret_cast<const u32x4*>(wsrc[i] + (long)(KT) * 128);                                \
    }
#define SA_CSTASH(XR, WR, BUFP)                                                                 \
    {                                                                                          \
        unsigned char* b_ = (BUFP);                                                            \
        _Pragma("unroll") for (int i = 0; i < XCH; ++i) *reinterpret_cast<u32x4*>(b_ + xdst[i]) = XR[i];   \
        _Pragma("unroll") for (int i = 0; i < WCH; ++i) *reinterpret_cast<u32x4*>(b_ + wdst[i]) = WR[i];   \
    }
    const int frow = lane & 31, fch = lane >> 5;
#define SA_CFRAGS(XF, WF, KK)                                                                                  \
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
#define SA_CMFMAS(XF, WF)                                  \
    _Pragma("unroll") for (int j = 0; j < FN; ++j)         \
        _Pragma("unroll") for (int i = 0; i < FM; ++i) Mfma<T>::run(acc[j][i], WF[j], XF[i]);
#define SA_CSGB_PAIRS()                                                                                 \
    _Pragma("unroll") for (int i_ = 0; i_ < (FM + FN < FM * FN ? FM + FN : FM * FN); ++i_) {            \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                              \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                              \
    }                                                                                                   \
    if constexpr (FM * FN > FM + FN) __builtin_amdgcn_sched_group_barrier(0x008, FM * FN - FM - FN, 0); \
    if constexpr (FM + FN > FM * FN) __builtin_amdgcn_sched_group_barrier(0x100, FM + FN - FM * FN, 0);
    // fragment reads of K-step kk+1 interleaved with the MFMAs of step kk, as in gemm.h (bf16; the fp32 Mfma is 4 instructions
    // per tile, so there the pattern only fixes the order of the first of each four)
#define SA_CCOMPUTE(CURP)                                  \
    {                                                      \
        const unsigned char* cur_ = (CURP);                \
        u32x4 xfa[FM], wfa[FN], xfb[FM], wfb[FN];          \
        SA_CFRAGS(xfa, wfa, 0);                            \
        SA_CFRAGS(xfb, wfb, 1);                            \
        SA_CMFMAS(xfa, wfa);                               \
        SA_CFRAGS(xfa, wfa, 2);                            \
        SA_CMFMAS(xfb, wfb);                               \
        SA_CFRAGS(xfb, wfb, 3);                            \
        SA_CMFMAS(xfa, wfa);                               \
        SA_CMFMAS(xfb, wfb);                               \
        if constexpr (sizeof(T) == 2) {                    \
            __builtin_amdgcn_sched_group_barrier(0x100, FM + FN, 0); \
            SA_CSGB_PAIRS();                               \
            SA_CSGB_PAIRS();                               \
            SA_CSGB_PAIRS();                               \
            __builtin_amdgcn_sched_group_barrier(0x008, FM * FN, 0); \
        }                                                  \
    }
    const int last = nk - 1;
    SA_CFETCH(xr0, wr0, 0);
    SA_CFETCH(xr1, wr1, min(1, last));
    SA_CSTASH(xr0, wr0, smem);
    __syncthreads();
    const int pairs = nk >> 1;
    for (int pi = 0; pi < pairs; ++pi) {
        const int kt = 2 * pi;
        SA_CFETCH(xr0, wr0, min(kt + 2, last));
        __builtin_amdgcn_sched_barrier(0);
        SA_CCOMPUTE(smem);
        __builtin_amdgcn_sched_barrier(0);
        SA_CSTASH(xr1, wr1, smem + BUF);
        __syncthreads();
        SA_CFETCH(xr1, wr1, min(kt + 3, last));
        __builtin_amdgcn_sched_barrier(0);
        SA_CCOMPUTE(smem + BUF);
        __builtin_amdgcn_sched_barrier(0);
        SA_CSTASH(xr0, wr0, smem);
        __syncthreads();
    }
    if (nk & 1) SA_CCOMPUTE(smem);
#undef SA_CFETCH
#undef SA_CSTASH
#undef SA_CCOMPUTE
#undef SA_CFRAGS
#undef SA_CMFMAS
#undef SA_CSGB_PAIRS
    // Epilogue. Bias (per (j, g), the same for every i) and the residual rows (per i) are fetched as BATCHES of unconditional
    // loads with clamped addresses: loaded where they are used, under `if (p.bias)` / `if (p.res)` / `continue`, hipcc branches
    // around every load and waits vmcnt(0) behind each -- 2 x FM x FN x 4 dependent L2 round trips per tile (gemm.h, r03 ISA).
    const bool has_bias = p.bias != nullptr, has_res = p.res != nullptr;      // wave-uniform
    using Raw = typename std::conditional<std::is_same<T, float>::value, float4, uint2>::type;   // 4 elements as loaded
    Raw bias_raw[FN][4];
    if (has_bias) {
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                bias_raw[j][g] = *reinterpret_cast<const Raw*>(p.bias + min(n0 + wn * WTN + j * 32 + g * 8 + (lane >> 5) * 4, p.Cout - 4));
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int m = m0 + wm * WTM + i * 32 + (lane & 31);
        Raw res_raw[FN][4];
        if (has_res) {
            const long mrow = (long)min(m, M - 1) * p.Cout;
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    res_raw[j][g] = *reinterpret_cast<const Raw*>(p.res + mrow + min(n0 + wn * WTN + j * 32 + g * 8 + (lane >> 5) * 4, p.Cout - 4));
        }
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * WTN + j * 32 + g * 8 + (lane >> 5) * 4;
                if (n >= p.Cout) continue;
                float v[4] = {acc[j][i][4 * g], acc[j][i][4 * g + 1], acc[j][i][4 * g + 2], acc[j][i][4 * g + 3]};
                if (has_bias) {
                    float b[4];
                    load4(reinterpret_cast<const T*>(&bias_raw[j][g]), b);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += b[r];
                }
                if (p.act == ACT_HSWISH) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = hardswish_f(v[r]);
                } else if (p.act == ACT_RELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                }
                if (has_res) {
                    float r4[4];
                    load4(reinterpret_cast<const T*>(&res_raw[j][g]), r4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += r4[r];
                }
                store4(p.out + (long)m * p.Cout + n, v[0], v[1], v[2], v[3]);
            }
        }
    }
}

template <typename T, int BM, int BN, int WM, int WN, bool CIN64 = false>
static inline int launch_conv_cfg(const ConvArgs<T>& a, hipStream_t s) {
    const int M = a.B * a.Ho * a.Wo;
    const int tiles = cdiv(M, BM) * cdiv(a.Cout, BN);
    constexpr size_t lds = (size_t)(BM + BN) * 128 * 2;
    auto kern = conv_gemm_kernel<T, BM, BN, WM, WN, CIN64>;
    static AttrOnce attr;
    attr.ensure(kern, lds);
    GemmProfiler& pf = gemm_profiler();
    const bool prof = pf.enabled && pf.n < GemmProfiler::POOL;
    if (prof) (void)hipEventRecord(pf.ev[2 * pf.n], s);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(64 * WM * WN), lds, s, a);
    if (prof) {
        (void)hipEventRecord(pf.ev[2 * pf.n + 1], s);
        pf.cfg_of[pf.n] = 3;       // profiler bucket 3 = implicit-GEMM convolutions
        pf.flops_of[pf.n] = 2.0 * M * a.Cout * a.KH * a.KW * a.Cin;
        pf.bytes_of[pf.n] = ((double)a.B * a.H * a.W * a.Cin + (double)a.Cout * a.Kpad + (double)M * a.Cout * (a.res ? 2 : 1)) * sizeof(T);
        pf.slab_of[pf.n] = 0.0;
        ++pf.n;
    }
    return (int)hipGetLastError();
}

// bf16 convolutions run on gemm.h's direct-to-LDS tiles (CONV gather: per-lane source addresses, zero page outside the image) with
// its XCD-aware tile order and LDS-staged, coalesced epilogue; round 2's register-staged conv_gemm_kernel below stays for fp32
// reference mode and for combinations the GEMM epilogues do not cover (r03: 3x3 convs were 0.3-0.5 PF/s on it).
template <typename T, int EPI>
static inline int launch_conv_on_gemm(const ConvArgs<T>& a, hipStream_t s) {
    const long M = (long)a.B * a.Ho * a.Wo;
    GemmArgs<T, T> g{nullptr, 0, a.w, (long)a.Kpad, a.out, (long)a.Cout, a.bias, a.res, (long)a.Cout, (int)M, a.Cout, a.Kpad};
    g.conv_in = a.in; g.conv_zero = a.zero;
    g.cH = a.H; g.cW = a.W; g.cCin = a.Cin; g.cHo = a.Ho; g.cWo = a.Wo; g.cKW = a.KW; g.cStride = a.stride; g.cPad = a.pad;
    g.cTaps = a.KH * a.KW;
    g.conv_lean = tuning().conv_lean;
    g.fd_hw = make_fastdiv((unsigned)(a.Ho * a.Wo)); g.fd_wo = make_fastdiv((unsigned)a.Wo);
    g.cv_m1 = (65536u + (unsigned)(a.Cin / 64) - 1) / (unsigned)std::max(1, a.Cin / 64); g.cv_m2 = (65536u + (unsigned)a.KW - 1) / (unsigned)a.KW;
    g.cv_rowskip = (a.W - a.KW) * a.Cin * (int)sizeof(T);
    if (tuning().bigtile && a.Kpad >= tuning().bigtile_min_k && a.Cout >= 256 && cdivl(M, 256) * cdiv(a.Cout, 256) >= 256) {
        // round 5: the persistent 8-phase loop with the gather in its request stream, for K-tiles aligned with filter taps (an odd K-tile count
        // gets a virtual K-tile of zeros). The one-tile 2-stage kernel spent 15 % of a K = 576 tile in its set-up, 24 % in its epilogue and
        // 4.0k cycles per K-tile in its drain-per-tile loop (tools/microbench/p8_timing.hip).
        if constexpr (sizeof(T) == 2) {
            const int nkc = a.Kpad / 64;
            const bool fits = tuning().persist && a.KH <= 7 && a.KW <= 7 && nkc < 4096 && nkc + (nkc & 1) >= 4 && a.Cout % 8 == 0 &&
                              (long)a.B * a.H * a.W * a.Cin * (long)sizeof(T) < (1L << 31) - (1L << 24);
            if (fits && (tuning().conv_persist & 1) && a.Cin % 64 == 0 && a.Cin <= 1024) return launch_gemm_persist<T, T, EPI, 1>(g, s);
            // Cin = 32 (the detector's largest single launch: 3 x 3 stride 2, 32 -> 512 channels at 256^2): two taps per K-tile, the tap chosen per
            // lane by the half of the 128-byte row its chunk lies in (Hardswish epilogue only: the one such convolution in the network)
            if constexpr (EPI == EPI_HARDSWISH)
                if (fits && (tuning().conv_persist & 2) && a.Cin == 32) return launch_gemm_persist<T, T, EPI, 2>(g, s);
        }
        return launch_gemm_cfg<T, T, 256, 256, 4, 2, EPI, false, 2, true>(g, s);
    }
    if (a.Cout >= 128) return launch_gemm_cfg<T, T, 128, 128, 2, 2, EPI, false, 2, true>(g, s);
    if (a.Cout >= 64) return launch_gemm_cfg<T, T, 128, 64, 4, 1, EPI, false, 2, true>(g, s);
    return launch_gemm_cfg<T, T, 128, 32, 4, 1, EPI, false, 2, true>(g, s);
}

template <typename T>
static inline int launch_conv(const ConvArgs<T>& a_in, hipStream_t s) {
    ConvArgs<T> a = a_in;
    if (a.Cin % Ty<T>::V16 != 0 || a.Kpad % Ty<T>::KE != 0 || a.Cout % 4 != 0) return SA_ERR_SHAPE;
    a.fcin = make_fastdiv((unsigned)a.Cin); a.fkw = make_fastdiv((unsigned)a.KW);
    a.fhw = make_fastdiv((unsigned)(a.Ho * a.Wo)); a.fwo = make_fastdiv((unsigned)a.Wo);
    const long M = (long)a.B * a.Ho * a.Wo;
    if constexpr (std::is_same<T, bf16_t>::value) {
        // (the 32-channel stem convolutions stay on the register-staged kernel: 128x32 tiles measured 5-15 % slower on the gather)
        if (a.zero && a.Cout >= 64 && a.Cout % 8 == 0 && M < (1L << 31)) {
            if (a.res && a.act == ACT_NONE) return launch_conv_on_gemm<T, EPI_RESIDUAL>(a, s);
            if (!a.res && a.act == ACT_HSWISH) return launch_conv_on_gemm<T, EPI_HARDSWISH>(a, s);
            if (!a.res && a.act == ACT_RELU) return launch_conv_on_gemm<T, EPI_RELU>(a, s);
            if (!a.res && a.act == ACT_NONE) return launch_conv_on_gemm<T, EPI_BIAS>(a, s);
        }
    }
    const bool cin64 = a.Cin % Ty<T>::KE == 0;       // K-tiles aligned with filter taps: scalar tap arithmetic
    if (a.Cout >= 128 && cdivl(M, 128) * cdiv(a.Cout, 128) >= 256)
        return cin64 ? launch_conv_cfg<T, 128, 128, 2, 2, true>(a, s) : launch_conv_cfg<T, 128, 128, 2, 2>(a, s);
    if (a.Cout >= 64 && cdivl(M, 128) * cdiv(a.Cout, 64) >= 128)
        return cin64 ? launch_conv_cfg<T, 128, 64, 4, 1, true>(a, s) : launch_conv_cfg<T, 128, 64, 4, 1>(a, s);
    if (a.Cout >= 64) return cin64 ? launch_conv_cfg<T, 64, 64, 2, 2, true>(a, s) : launch_conv_cfg<T, 64, 64, 2, 2>(a, s);
    return launch_conv_cfg<T, 128, 32, 4, 1>(a, s);
}

// ---------------------------------------------------------------------------------------------------
// Depthwise KxK convolution (NHWC), bias + activation. One thread = one output pixel x 8/4 channels (16 bytes).
// HBM-bound: reads K*K input vectors per output vector (neighbours hit L1/L2).
// Register-blocked variant for the shapes the detector uses (K = 3 / 5, stride 1 / 2): one thread = TX consecutive output
// pixels of a row x 16 bytes of channels. The plain kernel above reads every input vector K*K times through L1/L2 (the
// L2->CU path, not HBM, was its limit: 142 us average per launch in the r01 profile); here a loaded input column serves
// up to K outputs from registers. Accumulation order per output is unchanged (ky outer, kx inner).
template <typename T, int K, int S, int TX, bool FD = false>
__global__ __launch_bounds__(256) void dwconv_tx_kernel(const T* __restrict__ in, const T* __restrict__ w, const T* __restrict__ bias,
                                                        T* __restrict__ out, int B, int H, int W, int C, int Ho, int Wo, int pad, int act,
                                                        FastDiv fcv, FastDiv fwx, FastDiv fho) {
    constexpr int V = Ty<T>::V16;
    constexpr int NIN = (TX - 1) * S + K;            // input columns feeding TX outputs
    const int cv = C / V, wx = (Wo + TX - 1) / TX;
    // Workgroup b runs on XCD b % 8, each XCD with its own L2. In launch order the workgroups of vertically adjacent output rows landed
    // on different XCDs, so every input row crossed the fabric once per XCD that needed it: FETCH_SIZE was 2.4x the tensor for 3 x 3
    // and 8x for 5 x 5 (profiles/r03_k_det_hbm_traffic_pmc.md). Give XCD x the x-th contiguous eighth of the raster instead: rows that
    // share input rows are then neighbours in ONE XCD's queue.
    const unsigned nb = gridDim.x, xcd = blockIdx.x & 7, per = nb >> 3, rem = nb & 7;
    const unsigned lb = xcd * per + min(xcd, rem) + (blockIdx.x >> 3);
    int c0, ox0, oy, b;
    if constexpr (FD) {
        // round 5: 32-bit index (the launcher checks the thread count) split by host-made reciprocals of C / V, ceil(Wo / TX) and Ho -- the
        // 64-bit `%` and `/` below are ~700 instructions of software division per thread, in front of 27 loads and 288 FMAs
        const unsigned idx = lb * blockDim.x + threadIdx.x;
        if (idx >= (unsigned)B * Ho * wx * cv) return;
        const unsigned t = fast_div(idx, fcv), t2 = fast_div(t, fwx), bq = fast_div(t2, fho);
        c0 = (int)(idx - t * (unsigned)cv) * V;
        ox0 = (int)(t - t2 * (unsigned)wx) * TX; oy = (int)(t2 - bq * (unsigned)Ho); b = (int)bq;
    } else {
        const long idx = (long)lb * blockDim.x + threadIdx.x;
        if (idx >= (long)B * Ho * wx * cv) return;
        c0 = (int)(idx % cv) * V;
        const long t = idx / cv;
        ox0 = (int)(t % wx) * TX; oy = (int)((t / wx) % Ho); b = (int)(t / ((long)wx * Ho));
    }
    float acc[TX][V];
    if (bias) {
        float bv[V];
        unpack16(*reinterpret_cast<const uint4*>(bias + c0), bv, (T*)nullptr);
#pragma unroll
        for (int o = 0; o < TX; ++o)
#pragma unroll
            for (int i = 0; i < V; ++i) acc[o][i] = bv[i];
    } else {
#pragma unroll
        for (int o = 0; o < TX; ++o)
#pragma unroll
            for (int i = 0; i < V; ++i) acc[o][i] = 0.f;
    }
    const int ix0 = ox0 * S - pad;
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
        const int iy = oy * S + ky - pad;
        if (iy < 0 || iy >= H) continue;
        const T* row = in + ((long)b * H + iy) * W * C + c0;
        float xin[NIN][V];
#pragma unroll
        for (int j = 0; j < NIN; ++j) {
            const int ix = ix0 + j;
            uint4 raw = make_uint4(0u, 0u, 0u, 0u);
            if (ix >= 0 && ix < W) raw = *reinterpret_cast<const uint4*>(row + (long)ix * C);
            unpack16(raw, xin[j], (T*)nullptr);
        }
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            float wv[V];
            unpack16(*reinterpret_cast<const uint4*>(w + (long)(ky * K + kx) * C + c0), wv, (T*)nullptr);
#pragma unroll
            for (int o = 0; o < TX; ++o)
#pragma unroll
                for (int i = 0; i < V; ++i) acc[o][i] += xin[o * S + kx][i] * wv[i];
        }
    }
#pragma unroll
    for (int o = 0; o < TX; ++o) {
        if (ox0 + o >= Wo) break;
        float r[V];
#pragma unroll
        for (int i = 0; i < V; ++i) r[i] = act == ACT_HSWISH ? hardswish_f(acc[o][i]) : (act == ACT_RELU ? fmaxf(acc[o][i], 0.f) : acc[o][i]);
        T* op = out + (((long)b * Ho + oy) * Wo + ox0 + o) * C + c0;
#pragma unroll
        for (int i = 0; i < V; i += 4) store4(op + i, r[i], r[i + 1], r[i + 2], r[i + 3]);
    }
}

// Round 5, measured and NOT the default (sa::Tuning::dwconv_pipe = 1): the same thread mapping and accumulation order with the loads taken
// out of the control flow. In dwconv_tx_kernel every input vector sits behind its own bounds branch and a filter row outside the image is
// skipped by `continue`: hipcc emits a branch per load and an `s_waitcnt vmcnt(0)` after the first two -- six dependent round trips per
// thread at 4 waves per SIMD, 2.5 TB/s of HBM traffic on a kernel that moves two tensors and nothing else
// (profiles/r05_w_det_hbm_traffic_pmc.md). Here every load is unconditional (clamped address, masked to zero: a padded tap contributes
// 0 * w exactly as the reference's zero padding does). What hipcc makes of it -- whether the rows are written double-buffered or one at a
// time between compiler barriers, the loads are readonly / noalias and move over both -- is ALL K * (NIN + K) loads in front of the
// first use: 27 in flight per thread but 187 VGPRs = 2 waves per SIMD (K = 5: 256 VGPRs, 1 wave), and the 16-page forward is 4.5 %
// SLOWER (11.53 -> 12.05 ms, profiles/r05_ab_*): this kernel wants waves, not loads per wave. Bit-identical to dwconv_tx_kernel.
template <typename T, int K, int S, int TX>
__global__ __launch_bounds__(256) void dwconv_pipe_kernel(const T* __restrict__ in, const T* __restrict__ w, const T* __restrict__ bias,
                                                          T* __restrict__ out, int B, int H, int W, int C, int Ho, int Wo, int pad, int act,
                                                          FastDiv fcv, FastDiv fwx, FastDiv fho) {
    constexpr int V = Ty<T>::V16;
    constexpr int NIN = (TX - 1) * S + K;
    const int cv = C / V, wx = (Wo + TX - 1) / TX;
    const unsigned nb = gridDim.x, xcd = blockIdx.x & 7, per = nb >> 3, rem = nb & 7;
    const unsigned lb = xcd * per + min(xcd, rem) + (blockIdx.x >> 3);
    // 32-bit index (the launcher checks the thread count) split by host-made reciprocals of C / V, ceil(Wo / TX) and Ho: the round-3
    // kernel's 64-bit `%` and `/` were ~700 instructions of software division per thread in front of 27 loads and 288 FMAs
    const unsigned idx = lb * blockDim.x + threadIdx.x;
    if (idx >= (unsigned)B * Ho * wx * cv) return;
    const unsigned t = fast_div(idx, fcv), t2 = fast_div(t, fwx), bq = fast_div(t2, fho);
    const int c0 = (int)(idx - t * (unsigned)cv) * V;
    const int ox0 = (int)(t - t2 * (unsigned)wx) * TX, oy = (int)(t2 - bq * (unsigned)Ho), b = (int)bq;
    const int ix0 = ox0 * S - pad;
    int coff[NIN];
    unsigned cmask[NIN];
#pragma unroll
    for (int j = 0; j < NIN; ++j) {
        const int ix = ix0 + j;
        coff[j] = min(max(ix, 0), W - 1) * C;
        cmask[j] = (unsigned)ix < (unsigned)W ? 0xffffffffu : 0u;
    }
    const T* img = in + (long)b * H * W * C + c0;
    float acc[TX][V];
    uint4 braw = *reinterpret_cast<const uint4*>((bias ? bias : w) + c0);     // unconditional as well: a null bias reads the filter and is masked to + 0
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
        // one filter row: its NIN input vectors and K weight vectors
        uint4 raw[NIN], wraw[K];
        const int iy = oy * S + ky - pad;
        const T* row_ = img + (long)min(max(iy, 0), H - 1) * W * C;
#pragma unroll
        for (int j = 0; j < NIN; ++j) raw[j] = *reinterpret_cast<const uint4*>(row_ + coff[j]);
#pragma unroll
        for (int kx = 0; kx < K; ++kx) wraw[kx] = *reinterpret_cast<const uint4*>(w + (long)(ky * K + kx) * C + c0);
        asm volatile("" ::: "memory");       // (no effect on the readonly loads, see above)
        if (ky == 0) {
            const unsigned bm = bias ? 0xffffffffu : 0u;
            braw.x &= bm; braw.y &= bm; braw.z &= bm; braw.w &= bm;
            float bv[V];
            unpack16(braw, bv, (T*)nullptr);
#pragma unroll
            for (int o = 0; o < TX; ++o)
#pragma unroll
                for (int i = 0; i < V; ++i) acc[o][i] = bv[i];
        }
        const unsigned rmask = (unsigned)iy < (unsigned)H ? 0xffffffffu : 0u;
        float wv[K][V];
#pragma unroll
        for (int kx = 0; kx < K; ++kx) unpack16(wraw[kx], wv[kx], (T*)nullptr);
        // input column j feeds output o through tap kx = j - o * S: walking j upwards visits every output's taps in ascending kx, the same
        // order as the (kx outer, o inner) loops of dwconv_tx_kernel -- and only ONE unpacked column is live
#pragma unroll
        for (int j = 0; j < NIN; ++j) {
            uint4 r = raw[j];
            const unsigned m = cmask[j] & rmask;
            r.x &= m; r.y &= m; r.z &= m; r.w &= m;
            float x[V];
            unpack16(r, x, (T*)nullptr);
#pragma unroll
            for (int o = 0; o < TX; ++o) {
                const int kx = j - o * S;
                if (kx >= 0 && kx < K) {
#pragma unroll
                    for (int i = 0; i < V; ++i) acc[o][i] += x[i] * wv[kx][i];
                }
            }
        }
        asm volatile("" ::: "memory");
    }
#pragma unroll
    for (int o = 0; o < TX; ++o) {
        if (ox0 + o >= Wo) break;
        float r[V];
#pragma unroll
        for (int i = 0; i < V; ++i) r[i] = act == ACT_HSWISH ? hardswish_f(acc[o][i]) : (act == ACT_RELU ? fmaxf(acc[o][i], 0.f) : acc[o][i]);
        T* op = out + (((long)b * Ho + oy) * Wo + ox0 + o) * C + c0;
#pragma unroll
        for (int i = 0; i < V; i += 4) store4(op + i, r[i], r[i + 1], r[i + 2], r[i + 3]);
    }
}

// ---------------------------------------------------------------------------------------------------
// Grouped 1x1 convolution with group size GD in == GD out (LiteMLA aggreg[1], encoderdecoder.py:317):
// out[p, g*GD + o] = sum_i in[p, g*GD + i] * w[g*GD + o][i]. Workgroup = 8 pixels x one group (GD <= 32 lanes each).
template <typename T>
__global__ __launch_bounds__(256) void grouped1x1_kernel(const T* __restrict__ in, const T* __restrict__ w, T* __restrict__ out,
                                                         long P, int C, int GD) {
    // thread = (pixel lane pl, output channel o of group g); its weight row stays in registers for 8 pixels. The 32 lanes of a
    // pixel read the same 16-byte input chunks (one request per chunk after coalescing) -- no LDS, no barriers. (The first
    // version staged the 32x32 weights and 8 pixels through LDS per 8-pixel workgroup with scalar 2-byte loads: 138 us.)
    constexpr int V = Ty<T>::V16, PPB = 64;                  // pixels per workgroup
    const int g = blockIdx.y, tid = threadIdx.x, pl = tid >> 5, o = tid & 31;
    const long p0 = (long)blockIdx.x * PPB;
    const int oc = min(o, GD - 1);
    float wr[32];
#pragma unroll
    for (int c = 0; c < 32; c += V) {
        float t[V];
        if (c < GD) unpack16(*reinterpret_cast<const uint4*>(w + (long)(g * GD + oc) * GD + c), t, (T*)nullptr);
#pragma unroll
        for (int i = 0; i < V; ++i) wr[c + i] = (c < GD) ? t[i] : 0.f;
    }
    // the thread's 8 pixels in two batches of 4: all 16-byte loads of a batch are issued before the first use (the pixel loop with
    // its `break` made every pixel a separate dependent round trip: 8 per workgroup, 90 us per launch in the r03 profile)
    constexpr int NCH = 32 / V;                              // 16-byte chunks of a 32-channel group
#pragma unroll
    for (int kb = 0; kb < PPB / 8; kb += 4) {
        uint4 raw[4][NCH];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long pp = min(p0 + (kb + k) * 8 + pl, P - 1);
            const T* xr = in + pp * C + g * GD;
#pragma unroll
            for (int cc = 0; cc < NCH; ++cc) raw[k][cc] = *reinterpret_cast<const uint4*>(xr + min(cc * V, GD - V));
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long pp = p0 + (kb + k) * 8 + pl;
            float acc = 0.f;
#pragma unroll
            for (int cc = 0; cc < NCH; ++cc) {
                if (cc * V < GD) {
                    float x[V];
                    unpack16(raw[k][cc], x, (T*)nullptr);
#pragma unroll
                    for (int i = 0; i < V; ++i) acc += x[i] * wr[cc * V + i];
                }
            }
            if (o < GD && pp < P) Ty<T>::st(out + pp * C + g * GD + o, acc);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// LiteMLA ReLU linear attention (encoderdecoder.py:332-359), fp32 like the reference's _attn:
//   kv = relu(K)^T [V | 1]  (dim x dim+1),  out = relu(Q) kv,  out[:, :dim] / (out[:, dim] + eps)
// One workgroup per (image, head). Heads [0, heads_a) read q|k|v from `qa` (the qkv conv), the rest from `qb` (the
// multi-scale aggregation) -- the reference's channel concat (:349) is never materialised. dim <= 32.
// Two launches so the work spreads over the chip (one workgroup per (image, head) walking all H*W pixels alone took 183 us
// on 256 workgroups): litemla_kv_kernel reduces 256-pixel chunks to partial kv matrices, litemla_out_kernel sums a head's
// partials in chunk order (deterministic) and writes 256 pixels per workgroup.
constexpr int LITEMLA_CHUNK = 256;
constexpr int LITEMLA_KV = 32 * 33;                          // floats per (image, head, chunk) partial

template <typename T, int DIM>
__global__ __launch_bounds__(256) void litemla_kv_kernel(const T* __restrict__ qa, const T* __restrict__ qb, float* __restrict__ kvp,
                                                         int HW, int heads_a, int heads) {
    constexpr int dim = DIM;
    __shared__ float ks[64 * 32];
    __shared__ float vs[64 * 33];
    const int b = blockIdx.x, h = blockIdx.y, ch = blockIdx.z, nch = gridDim.z, tid = threadIdx.x;
    const int Ca = heads_a * 3 * dim, Cb = (heads - heads_a) * 3 * dim;
    const T* src = h < heads_a ? qa + (long)b * HW * Ca + h * 3 * dim : qb + (long)b * HW * Cb + (h - heads_a) * 3 * dim;
    const int C = h < heads_a ? Ca : Cb;
    const int d1 = dim + 1, npair = dim * d1;
    float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};        // pairs tid, tid+256, ... (dim 32: 1056 pairs -> <= 5 per thread)
    const int n_end = min(HW, (ch + 1) * LITEMLA_CHUNK);
    for (int n0 = ch * LITEMLA_CHUNK; n0 < n_end; n0 += 64) {
        const int nn = min(64, n_end - n0);
        // 16-byte loads: the q|k|v tensor is 200 MB per call at 1024^2 x 16 pages, so this kernel is HBM-bound once the
        // staging stops issuing 2-byte loads (first version: 1.3 TB/s)
        constexpr int V = Ty<T>::V16, CPP = dim / V;            // 16-byte chunks of k (and of v) per pixel
        for (int i = tid; i < nn * CPP; i += 256) {
            const int n = i / CPP, c = (i % CPP) * V;
            const T* row = src + (long)(n0 + n) * C;
            float kx[V], vx[V];
            unpack16(*reinterpret_cast<const uint4*>(row + dim + c), kx, (T*)nullptr);
            unpack16(*reinterpret_cast<const uint4*>(row + 2 * dim + c), vx, (T*)nullptr);
#pragma unroll
            for (int e = 0; e < V; ++e) { ks[n * 32 + c + e] = fmaxf(kx[e], 0.f); vs[n * 33 + c + e] = vx[e]; }
            if (c == 0) vs[n * 33 + dim] = 1.f;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            const int pr = tid + r * 256;
            if (pr < npair) {
                const int i = pr / d1, j = pr % d1;
                float a = acc[r];
                for (int n = 0; n < nn; ++n) a += ks[n * 32 + i] * vs[n * 33 + j];
                acc[r] = a;
            }
        }
        __syncthreads();
    }
    float* dst = kvp + (((long)b * heads + h) * nch + ch) * LITEMLA_KV;
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        const int pr = tid + r * 256;
        if (pr < npair) dst[(pr / d1) * 33 + (pr % d1)] = acc[r];
    }
}

template <typename T, int DIM>
__global__ __launch_bounds__(256) void litemla_out_kernel(const T* __restrict__ qa, const T* __restrict__ qb, const float* __restrict__ kvp,
                                                          T* __restrict__ out, int HW, int heads_a, int heads, float eps) {
    constexpr int dim = DIM;
    __shared__ float kv[32 * 33];
    const int b = blockIdx.x, h = blockIdx.y, ch = blockIdx.z, nch = gridDim.z, tid = threadIdx.x;
    const int Ca = heads_a * 3 * dim, Cb = (heads - heads_a) * 3 * dim;
    const T* src = h < heads_a ? qa + (long)b * HW * Ca + h * 3 * dim : qb + (long)b * HW * Cb + (h - heads_a) * 3 * dim;
    const int C = h < heads_a ? Ca : Cb;
    const int d1 = dim + 1, npair = dim * d1;
    const float* part = kvp + ((long)b * heads + h) * nch * LITEMLA_KV;
    for (int pr = tid; pr < npair; pr += 256) {
        const int e = (pr / d1) * 33 + (pr % d1);
        float a = 0.f;
        for (int c = 0; c < nch; ++c) a += part[(long)c * LITEMLA_KV + e];
        kv[e] = a;
    }
    __syncthreads();
    const int Cout = heads * dim;
    const int n = ch * LITEMLA_CHUNK + tid;
    if (n < HW) {
        const T* row = src + (long)n * C;
        float q[DIM];
        constexpr int V = Ty<T>::V16;
#pragma unroll
        for (int i = 0; i < DIM; i += V) {
            float t[V];
            unpack16(*reinterpret_cast<const uint4*>(row + i), t, (T*)nullptr);
#pragma unroll
            for (int e = 0; e < V; ++e) q[i + e] = fmaxf(t[e], 0.f);
        }
        float den = 0.f;                       // the appended ones-column of v: sum_i q_i * sum_n k_ni
#pragma unroll
        for (int i = 0; i < DIM; ++i) den += q[i] * kv[i * 33 + DIM];
        const float inv = 1.0f / (den + eps);
        T* dst = out + ((long)b * HW + n) * Cout + h * dim;
#pragma unroll 2
        for (int j = 0; j < DIM; j += 4) {
            float o4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < DIM; ++i) {
                o4[0] += q[i] * kv[i * 33 + j]; o4[1] += q[i] * kv[i * 33 + j + 1];
                o4[2] += q[i] * kv[i * 33 + j + 2]; o4[3] += q[i] * kv[i * 33 + j + 3];
            }
            store4(dst + j, o4[0] * inv, o4[1] * inv, o4[2] * inv, o4[3] * inv);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Bilinear resize (align_corners=False, torch semantics) of an NHWC map into a channel slice of a wider NHWC map:
// dst[b, y, x, c_off + c] = bilinear(src[b, :, :, c]).  Used for the decode head's upsample + concat (:709-715).
__device__ __forceinline__ void bilin_coeff(int d, int in_size, float scale, int& i0, int& i1, float& l1) {
    float s = ((float)d + 0.5f) * scale - 0.5f;
    s = s < 0.f ? 0.f : s;
    i0 = (int)s;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = s - (float)i0;
}

template <typename T>
__global__ void upsample_concat_kernel(const T* __restrict__ src, T* __restrict__ dst, int B, int Hs, int Ws, int C, int Hd, int Wd,
                                       int Cd, int c_off) {
    constexpr int V = Ty<T>::V16;
    const int cv = C / V;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)B * Hd * Wd * cv) return;
    const int c0 = (int)(idx % cv) * V;
    const long pix = idx / cv;
    const int x = (int)(pix % Wd), y = (int)((pix / Wd) % Hd), b = (int)(pix / ((long)Wd * Hd));
    int y0, y1, x0, x1; float ly, lx;
    bilin_coeff(y, Hs, (float)Hs / (float)Hd, y0, y1, ly);
    bilin_coeff(x, Ws, (float)Ws / (float)Wd, x0, x1, lx);
    float a[V], bq[V], c[V], d[V];
    const T* base = src + (long)b * Hs * Ws * C + c0;
    unpack16(*reinterpret_cast<const uint4*>(base + ((long)y0 * Ws + x0) * C), a, (T*)nullptr);
    unpack16(*reinterpret_cast<const uint4*>(base + ((long)y0 * Ws + x1) * C), bq, (T*)nullptr);
    unpack16(*reinterpret_cast<const uint4*>(base + ((long)y1 * Ws + x0) * C), c, (T*)nullptr);
    unpack16(*reinterpret_cast<const uint4*>(base + ((long)y1 * Ws + x1) * C), d, (T*)nullptr);
    T* o = dst + pix * Cd + c_off + c0;
    float r[V];
#pragma unroll
    for (int i = 0; i < V; ++i)
        r[i] = (1.f - ly) * ((1.f - lx) * a[i] + lx * bq[i]) + ly * ((1.f - lx) * c[i] + lx * d[i]);
#pragma unroll
    for (int i = 0; i < V; i += 4) store4(o + i, r[i], r[i + 1], r[i + 2], r[i + 3]);
}

// ---------------------------------------------------------------------------------------------------
// Classifier 1x1 conv (C -> L labels, L <= 4) + sigmoid, NHWC in -> fp32 NCHW planes [B, L, H, W] (:720, :747).
// 16 lanes share a pixel: each walks the channel row in 16-byte steps 16 chunks apart, so a wave reads four whole pixel rows
// with fully coalesced loads; one DPP row reduction per label. (The first version gave every thread its own pixel: adjacent
// lanes were C * 2 bytes apart, 64 cache lines per load instruction, 1.1 ms per 16 pages = 0.95 TB/s.)
template <typename T>
__global__ __launch_bounds__(256) void classify_sigmoid_kernel(const T* __restrict__ in, const T* __restrict__ w, const T* __restrict__ bias,
                                                               float* __restrict__ out, long P, long HW, int C, int L) {
    constexpr int V = Ty<T>::V16;
    const int sub = threadIdx.x & 15;
    const long p = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const long pc = p < P ? p : P - 1;                      // clamped: all 16 lanes of a row take part in the DPP sums
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const T* row = in + pc * C;
    for (int c = sub * V; c < C; c += 16 * V) {
        float xv[V];
        unpack16(*reinterpret_cast<const uint4*>(row + c), xv, (T*)nullptr);
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            if (l < L) {
                float wv[V];
                unpack16(*reinterpret_cast<const uint4*>(w + (long)l * C + c), wv, (T*)nullptr);
#pragma unroll
                for (int i = 0; i < V; ++i) acc[l] += xv[i] * wv[i];
            }
        }
    }
#pragma unroll
    for (int l = 0; l < 4; ++l) acc[l] = row16_sum(acc[l]);
    if (sub == 0 && p < P) {
        const long b = p / HW, r = p % HW;
        for (int l = 0; l < L; ++l) {
            const float z = Ty<T>::rnd(acc[l] + Ty<T>::ld(bias + l));
            out[(b * L + l) * HW + r] = Ty<T>::rnd(1.0f / (1.0f + expf(-z)));       // expit in the model dtype, then .float()
        }
    }
}

// Decode head, folded (SA_DET_UPSUM_CLASSIFY, include/surya_amd.h): per full-resolution pixel
//   v = z0 + sum_s bilinear(z_s)       (fp32; z_s = the stage's own 1x1 conv to the decoder width, bias and BatchNorm folded into z0's)
//   y = T(relu(v))                     (what the reference stores after linear_fuse + batch_norm + ReLU, :717-719)
//   plane_l = T(sigmoid(T(w_l . y + b_l)))
// replacing upsample_concat x 4 + the K = 512 fuse GEMM + classify_sigmoid: the [P, 512] concat (1.07 GB per 16 pages at 1024^2) is
// neither written nor read, and neither is the fuse GEMM's output. Same lane geometry as classify_sigmoid_kernel: 16 lanes share a
// pixel and walk its channel row in 16-byte steps; the low-resolution taps of neighbouring pixels are the same cache lines.
struct UpsumSrc {
    const void* p[3];
    int h[3], w[3];
    int n;
};

template <typename T>
__global__ __launch_bounds__(256) void head_upsum_classify_kernel(const T* __restrict__ z0, UpsumSrc src, const T* __restrict__ w,
                                                                  const T* __restrict__ bias, float* __restrict__ out, long P, int H0,
                                                                  int W0, int C, int L) {
    constexpr int V = Ty<T>::V16;
    const int sub = threadIdx.x & 15;
    const long p = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const long pc = p < P ? p : P - 1;                      // clamped: all 16 lanes of a row take part in the DPP sums
    const int x = (int)(pc % W0), y = (int)((pc / W0) % H0);
    const long b = pc / ((long)W0 * H0);
    long o00[3], o01[3], o10[3], o11[3];
    float ly[3], lx[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        if (s < src.n) {
            int y0, y1, x0, x1;
            bilin_coeff(y, src.h[s], (float)src.h[s] / (float)H0, y0, y1, ly[s]);
            bilin_coeff(x, src.w[s], (float)src.w[s] / (float)W0, x0, x1, lx[s]);
            const long base = b * src.h[s] * src.w[s];
            o00[s] = (base + (long)y0 * src.w[s] + x0) * C; o01[s] = (base + (long)y0 * src.w[s] + x1) * C;
            o10[s] = (base + (long)y1 * src.w[s] + x0) * C; o11[s] = (base + (long)y1 * src.w[s] + x1) * C;
        }
    }
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const T* row = z0 + pc * C;
    for (int c = sub * V; c < C; c += 16 * V) {
        float v[V];
        unpack16(*reinterpret_cast<const uint4*>(row + c), v, (T*)nullptr);
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            if (s < src.n) {
                const T* zs = reinterpret_cast<const T*>(src.p[s]) + c;
                float a[V], bq[V], cq[V], d[V];
                unpack16(*reinterpret_cast<const uint4*>(zs + o00[s]), a, (T*)nullptr);
                unpack16(*reinterpret_cast<const uint4*>(zs + o01[s]), bq, (T*)nullptr);
                unpack16(*reinterpret_cast<const uint4*>(zs + o10[s]), cq, (T*)nullptr);
                unpack16(*reinterpret_cast<const uint4*>(zs + o11[s]), d, (T*)nullptr);
#pragma unroll
                for (int i = 0; i < V; ++i)     // the expression of upsample_concat_kernel
                    v[i] += (1.f - ly[s]) * ((1.f - lx[s]) * a[i] + lx[s] * bq[i]) + ly[s] * ((1.f - lx[s]) * cq[i] + lx[s] * d[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < V; ++i) v[i] = Ty<T>::rnd(fmaxf(v[i], 0.f));
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            if (l < L) {
                float wv[V];
                unpack16(*reinterpret_cast<const uint4*>(w + (long)l * C + c), wv, (T*)nullptr);
#pragma unroll
                for (int i = 0; i < V; ++i) acc[l] += v[i] * wv[i];
            }
        }
    }
#pragma unroll
    for (int l = 0; l < 4; ++l) acc[l] = row16_sum(acc[l]);
    if (sub == 0 && p < P) {
        const long HW = (long)H0 * W0, r = p % HW;
        for (int l = 0; l < L; ++l) {
            const float z = Ty<T>::rnd(acc[l] + Ty<T>::ld(bias + l));
            out[(b * L + l) * HW + r] = Ty<T>::rnd(1.0f / (1.0f + expf(-z)));       // expit in the model dtype, then .float()
        }
    }
}

// The same pass, register-blocked (gpurun r04k: the per-pixel kernel above issues 13 16-byte loads per pixel and channel chunk --
// 4 taps x 3 addends + z0 -- and is bound by the texture-address path, ~1.1 ms per 16 pages, which ate everything the fold had
// saved). Here 16 lanes own a 4 x 2 block of output pixels; for a stage R = 2 / 4 / 8 times coarser (R a power of two, block aligned)
// the taps of the 8 pixels are a 3 x 4 / 2 x 3 / 2 x 2 tile of the coarse map whose positions RELATIVE to the first pixel's tap are
// compile-time constants, so the tile is loaded once (12 / 6 / 4 loads instead of 32), interpolated horizontally once per tile row and
// vertically per pixel: 30 loads per 8 pixels instead of 104. The bilinear form is PyTorch's with the source index clamped instead of
// the source coordinate (a border pixel reads the edge row twice with weights that sum to 1): equal up to one fp32 rounding.
template <int R, int BH> struct UpsumGeo {              // BH = pixel rows of the block (2, or 1: fewer registers, a third wave per SIMD)
    static constexpr int NR = (R == 2 && BH == 2) ? 3 : 2, NC = R == 2 ? 4 : (R == 4 ? 3 : 2);
    __device__ static constexpr int offy(int py) { return (R == 2 && BH == 2) ? py : 0; }
    __device__ static constexpr int offx(int px) { return R == 2 ? (px + 1) / 2 : (R == 4 ? px / 2 : 0); }
};

// The arithmetic runs on fp32 PAIRS (v_pk_mul_f32 / v_pk_fma_f32: the first blocked version spent ~30 scalar VALU operations per value
// -- 1.14 ms per 16 pages, gpurun r04k, VALU-bound at one wave per SIMD -- unpack, two interpolation passes, a software bf16 rounding
// and two multiply-adds); the bf16 path rounds y with v_cvt_pk_bf16_f32 and feeds the classifier's bf16 weight pairs to
// v_dot2c_f32_bf16 (exact products, fp32 accumulation).
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

template <typename T> struct UpsumPk;
template <> struct UpsumPk<bf16_t> {
    static constexpr int NP = 4;                             // pairs per 16-byte chunk
    __device__ __forceinline__ static void unpack(const uint4& r, f32x2 (&o)[4]) {
        o[0] = f32x2{__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u)};
        o[1] = f32x2{__uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u)};
        o[2] = f32x2{__uint_as_float(r.z << 16), __uint_as_float(r.z & 0xffff0000u)};
        o[3] = f32x2{__uint_as_float(r.w << 16), __uint_as_float(r.w & 0xffff0000u)};
    }
};
template <> struct UpsumPk<float> {
    static constexpr int NP = 2;
    __device__ __forceinline__ static void unpack(const uint4& r, f32x2 (&o)[2]) {
        o[0] = f32x2{__uint_as_float(r.x), __uint_as_float(r.y)};
        o[1] = f32x2{__uint_as_float(r.z), __uint_as_float(r.w)};
    }
};

// An addend's tap tile as it comes from memory. Loading (upsum_load) and using (upsum_apply) are separate calls so that the kernel can
// request z0's 8 rows and all three tiles -- 30 16-byte loads per lane -- BEFORE the first value is needed: with the loads placed next
// to their uses the compiler waited per tile row, eight dependent L2 / HBM round trips per channel chunk (701 us per 16 pages, r04k).
template <int R, int BH> struct UpsumTile { uint4 raw[UpsumGeo<R, BH>::NR][UpsumGeo<R, BH>::NC]; int iy0, ix0; };

template <typename T, int R, int BH>
__device__ __forceinline__ void upsum_load(UpsumTile<R, BH>& t, const T* __restrict__ zs, long img, int Hs, int Ws, int C, int by, int bx) {
    typedef UpsumGeo<R, BH> G;
    // first pixel's source coordinate and tap (may be -1 at the top / left edge: clamped when the tile is addressed)
    const float sy0 = ((float)(BH * by) + 0.5f) / (float)R - 0.5f, sx0 = ((float)(4 * bx) + 0.5f) / (float)R - 0.5f;
    t.iy0 = (int)floorf(sy0); t.ix0 = (int)floorf(sx0);
#pragma unroll
    for (int r = 0; r < G::NR; ++r) {
        const int row = min(max(t.iy0 + r, 0), Hs - 1);
#pragma unroll
        for (int c = 0; c < G::NC; ++c) {
            const int col = min(max(t.ix0 + c, 0), Ws - 1);
            t.raw[r][c] = *reinterpret_cast<const uint4*>(zs + ((img * Hs + row) * Ws + col) * C);
        }
    }
}

template <typename T, int R, int BH, int NP>
__device__ __forceinline__ void upsum_apply(f32x2 (&v)[BH][4][NP], const UpsumTile<R, BH>& tile, int by, int bx) {
    typedef UpsumGeo<R, BH> G;
    float lyv[BH], lxv[4];
#pragma unroll
    for (int py = 0; py < BH; ++py) lyv[py] = (((float)(BH * by + py) + 0.5f) / (float)R - 0.5f) - (float)(tile.iy0 + G::offy(py));
#pragma unroll
    for (int px = 0; px < 4; ++px) lxv[px] = (((float)(4 * bx + px) + 0.5f) / (float)R - 0.5f) - (float)(tile.ix0 + G::offx(px));
#pragma unroll
    for (int r = 0; r < G::NR; ++r) {                       // one tile row at a time: interpolate it for the block's 4 columns, hand it to the rows that use it
        f32x2 t[G::NC][NP];
#pragma unroll
        for (int c = 0; c < G::NC; ++c) UpsumPk<T>::unpack(tile.raw[r][c], t[c]);
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            const f32x2 l1 = f32x2{lxv[px], lxv[px]}, l0 = f32x2{1.f - lxv[px], 1.f - lxv[px]};
            f32x2 hx[NP];
#pragma unroll
            for (int i = 0; i < NP; ++i) hx[i] = l0 * t[G::offx(px)][i] + l1 * t[G::offx(px) + 1][i];
#pragma unroll
            for (int py = 0; py < BH; ++py) {
                if (G::offy(py) == r) {                     // compile-time after unrolling: this row is the pixel row's upper tap ...
                    const f32x2 wy = f32x2{1.f - lyv[py], 1.f - lyv[py]};
#pragma unroll
                    for (int i = 0; i < NP; ++i) v[py][px][i] += wy * hx[i];
                } else if (G::offy(py) + 1 == r) {          // ... or its lower tap
                    const f32x2 wy = f32x2{lyv[py], lyv[py]};
#pragma unroll
                    for (int i = 0; i < NP; ++i) v[py][px][i] += wy * hx[i];
                }
            }
        }
    }
}

template <typename T, int R1, int R2, int R3, int BH>   // ratios of the addends (0 = absent); pixel rows per block
__global__ __launch_bounds__(256) void head_upsum_classify_blk_kernel(const T* __restrict__ z0, const T* __restrict__ z1,
                                                                      const T* __restrict__ z2, const T* __restrict__ z3,
                                                                      const T* __restrict__ w, const T* __restrict__ bias,
                                                                      float* __restrict__ out, int B, int H0, int W0, int C, int L) {
    constexpr int V = Ty<T>::V16, NP = UpsumPk<T>::NP;
    constexpr bool BF = std::is_same<T, bf16_t>::value;
    const int sub = threadIdx.x & 15;
    const int bw = W0 / 4, bh = H0 / BH;
    const long nblk = (long)B * bh * bw;
    const long g = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const long gc = g < nblk ? g : nblk - 1;                // clamped: all 16 lanes take part in the DPP sums
    const int bx = (int)(gc % bw), by = (int)((gc / bw) % bh);
    const long img = gc / ((long)bw * bh);
    float acc[BH][4][2];                                     // bf16: dot2 accumulators; fp32: the two halves are kept apart in accp
    f32x2 accp[BH][4][2];
#pragma unroll
    for (int py = 0; py < BH; ++py)
#pragma unroll
        for (int px = 0; px < 4; ++px)
#pragma unroll
            for (int l = 0; l < 2; ++l) { acc[py][px][l] = 0.f; accp[py][px][l] = f32x2{0.f, 0.f}; }
    const long w1off = (long)(L > 1 ? 1 : 0) * C;
    for (int c = sub * V; c < C; c += 16 * V) {
        // every load of this chunk first ...
        uint4 zr[BH][4];
#pragma unroll
        for (int py = 0; py < BH; ++py)
#pragma unroll
            for (int px = 0; px < 4; ++px)
                zr[py][px] = *reinterpret_cast<const uint4*>(z0 + ((img * H0 + BH * by + py) * W0 + 4 * bx + px) * C + c);
        UpsumTile<R1 ? R1 : 2, BH> t1; UpsumTile<R2 ? R2 : 2, BH> t2; UpsumTile<R3 ? R3 : 2, BH> t3;
        if constexpr (R1 > 0) upsum_load<T, R1, BH>(t1, z1 + c, img, H0 / R1, W0 / R1, C, by, bx);
        if constexpr (R2 > 0) upsum_load<T, R2, BH>(t2, z2 + c, img, H0 / R2, W0 / R2, C, by, bx);
        if constexpr (R3 > 0) upsum_load<T, R3, BH>(t3, z3 + c, img, H0 / R3, W0 / R3, C, by, bx);
        const uint4 w0r = *reinterpret_cast<const uint4*>(w + c), w1r = *reinterpret_cast<const uint4*>(w + w1off + c);
        // ... then the arithmetic, in load order
        f32x2 v[BH][4][NP];
#pragma unroll
        for (int py = 0; py < BH; ++py)
#pragma unroll
            for (int px = 0; px < 4; ++px) UpsumPk<T>::unpack(zr[py][px], v[py][px]);
        if constexpr (R1 > 0) upsum_apply<T, R1, BH, NP>(v, t1, by, bx);
        if constexpr (R2 > 0) upsum_apply<T, R2, BH, NP>(v, t2, by, bx);
        if constexpr (R3 > 0) upsum_apply<T, R3, BH, NP>(v, t3, by, bx);
        const uint32_t w0u[4] = {w0r.x, w0r.y, w0r.z, w0r.w}, w1u[4] = {w1r.x, w1r.y, w1r.z, w1r.w};
        f32x2 w0p[NP], w1p[NP];
        if constexpr (!BF) { UpsumPk<T>::unpack(w0r, w0p); UpsumPk<T>::unpack(w1r, w1p); }
#pragma unroll
        for (int py = 0; py < BH; ++py)
#pragma unroll
            for (int px = 0; px < 4; ++px)
#pragma unroll
                for (int i = 0; i < NP; ++i) {
                    const f32x2 yv = f32x2{fmaxf(v[py][px][i].x, 0.f), fmaxf(v[py][px][i].y, 0.f)};
                    if constexpr (BF) {
                        const bf16x2_t yb = __builtin_convertvector(yv, bf16x2_t);       // y = T(relu(v)), round to nearest even
                        acc[py][px][0] = __builtin_amdgcn_fdot2_f32_bf16(yb, __builtin_bit_cast(bf16x2_t, w0u[i]), acc[py][px][0], false);
                        acc[py][px][1] = __builtin_amdgcn_fdot2_f32_bf16(yb, __builtin_bit_cast(bf16x2_t, w1u[i]), acc[py][px][1], false);
                    } else {
                        accp[py][px][0] += yv * w0p[i];
                        accp[py][px][1] += yv * w1p[i];
                    }
                }
    }
    const long HW = (long)H0 * W0;
    const float bl[2] = {Ty<T>::ld(bias), Ty<T>::ld(bias + (L > 1 ? 1 : 0))};     // once, not per output (16 dependent round trips in r04k's ISA)
#pragma unroll
    for (int py = 0; py < BH; ++py)
#pragma unroll
        for (int px = 0; px < 4; ++px)
#pragma unroll
            for (int l = 0; l < 2; ++l) {
                const float a = row16_sum(BF ? acc[py][px][l] : accp[py][px][l].x + accp[py][px][l].y);
                if (sub == 0 && g < nblk && l < L) {
                    const float z = Ty<T>::rnd(a + bl[l]);
                    out[(img * L + l) * HW + (long)(BH * by + py) * W0 + 4 * bx + px] = Ty<T>::rnd(1.0f / (1.0f + expf(-z)));
                }
            }
}

// fp32 planes [N, Hs, Ws] -> [N, Hd, Wd], bilinear align_corners=False (detection/__init__.py:121-129).
__global__ void upsample_planes_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int Hs, int Ws, int Hd, int Wd) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)N * Hd * Wd) return;
    const int x = (int)(idx % Wd), y = (int)((idx / Wd) % Hd);
    const long n = idx / ((long)Wd * Hd);
    int y0, y1, x0, x1; float ly, lx;
    bilin_coeff(y, Hs, (float)Hs / (float)Hd, y0, y1, ly);
    bilin_coeff(x, Ws, (float)Ws / (float)Wd, x0, x1, lx);
    const float* s = src + n * Hs * Ws;
    dst[idx] = (1.f - ly) * ((1.f - lx) * s[(long)y0 * Ws + x0] + lx * s[(long)y0 * Ws + x1]) +
               ly * ((1.f - lx) * s[(long)y1 * Ws + x0] + lx * s[(long)y1 * Ws + x1]);
}

// The same map, four horizontally adjacent outputs per thread and one 16-byte store (round 6: the one-output kernel wrote 134 MB per 16 pages
// in 4-byte stores, 119 us against ~25 us of HBM time). Per output the expression is the kernel's above, term for term: bit-identical.
__global__ __launch_bounds__(256) void upsample_planes4_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int Hs, int Ws, int Hd, int Wd) {
    const int wq = Wd >> 2;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)N * Hd * wq) return;
    const int xq = (int)(idx % wq), y = (int)((idx / wq) % Hd);
    const long n = idx / ((long)wq * Hd);
    int y0, y1; float ly;
    bilin_coeff(y, Hs, (float)Hs / (float)Hd, y0, y1, ly);
    const float* s0 = src + n * Hs * Ws + (long)y0 * Ws;
    const float* s1 = src + n * Hs * Ws + (long)y1 * Ws;
    float o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int x0, x1; float lx;
        bilin_coeff(xq * 4 + i, Ws, (float)Ws / (float)Wd, x0, x1, lx);
        o[i] = (1.f - ly) * ((1.f - lx) * s0[x0] + lx * s0[x1]) + ly * ((1.f - lx) * s1[x0] + lx * s1[x1]);
    }
    *reinterpret_cast<float4*>(dst + (n * Hd + y) * (long)Wd + xq * 4) = make_float4(o[0], o[1], o[2], o[3]);
}

// The exact x4 case (every shipped configuration: the heat maps leave at 4x the quarter-resolution planes): a thread owns a 4 x 4 block of outputs = one
// source pixel's footprint, reads its 3 x 3 source neighbourhood once (9 loads for 16 outputs; upsample_planes4_kernel: 16 loads and three 64-bit
// divisions for 4) and writes four float4 rows. Coefficients, operands and the expression are upsample_planes4_kernel's (bilin_coeff per output row /
// column; its indices always fall on the three loaded rows / columns: (d + 0.5) / 4 - 0.5 is exact in fp32). 55 -> 33 us per 16 pages.
__global__ __launch_bounds__(256) void upsample_planes_x4_kernel(const float* __restrict__ src, float* __restrict__ dst, int Hs, int Ws) {
    const int xq = blockIdx.x * 64 + threadIdx.x, yq = blockIdx.y * 4 + threadIdx.y;
    if (xq >= Ws || yq >= Hs) return;
    const long n = blockIdx.z;
    const int Hd = Hs * 4, Wd = Ws * 4;
    const float* sp = src + n * Hs * Ws;
    const int xs[3] = {max(xq - 1, 0), xq, min(xq + 1, Ws - 1)}, ys[3] = {max(yq - 1, 0), yq, min(yq + 1, Hs - 1)};
    float v[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) v[r][c] = sp[(long)ys[r] * Ws + xs[c]];
    int cx0[4], cx1[4];
    float lx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int x0, x1;
        bilin_coeff(xq * 4 + i, Ws, (float)Ws / (float)Wd, x0, x1, lx[i]);
        cx0[i] = x0 == xq ? 1 : (x0 < xq ? 0 : 2);
        cx1[i] = x1 == xq ? 1 : (x1 < xq ? 0 : 2);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int y0, y1; float ly;
        bilin_coeff(yq * 4 + j, Hs, (float)Hs / (float)Hd, y0, y1, ly);
        const int r0 = y0 == yq ? 1 : (y0 < yq ? 0 : 2), r1 = y1 == yq ? 1 : (y1 < yq ? 0 : 2);
        float a3[3], b3[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            a3[c] = r0 == 0 ? v[0][c] : (r0 == 1 ? v[1][c] : v[2][c]);
            b3[c] = r1 == 0 ? v[0][c] : (r1 == 1 ? v[1][c] : v[2][c]);
        }
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float s00 = cx0[i] == 0 ? a3[0] : (cx0[i] == 1 ? a3[1] : a3[2]), s01 = cx1[i] == 0 ? a3[0] : (cx1[i] == 1 ? a3[1] : a3[2]);
            const float s10 = cx0[i] == 0 ? b3[0] : (cx0[i] == 1 ? b3[1] : b3[2]), s11 = cx1[i] == 0 ? b3[0] : (cx1[i] == 1 ? b3[1] : b3[2]);
            o[i] = (1.f - ly) * ((1.f - lx[i]) * s00 + lx[i] * s01) + ly * ((1.f - lx[i]) * s10 + lx[i] * s11);
        }
        *reinterpret_cast<float4*>(dst + (n * Hd + yq * 4 + j) * (long)Wd + xq * 4) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

}  // namespace sa
