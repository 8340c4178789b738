// Common device/host helpers for the gfx950 (CDNA4, wave64) kernels of the surya hot path.
// Written for MI355X only: no CUDA / multi-backend paths.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <stdint.h>
#include <stdio.h>

#include "../../include/surya_amd.h"

namespace sa {

typedef unsigned short bf16_t;   // storage type for bf16 (bit pattern)
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
// 16-byte register chunk. NOT HIP's uint4: arrays of that struct type are demoted to scratch memory by hipcc (measured:
// the GEMM's staging registers went through scratch, 272 B/lane), native vector types stay in VGPRs.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Status codes (SA_OK, SA_ERR_*) come from the C-ABI header: negative = caller error, positive = hipError_t.

#define SA_HIP(expr)                                                                                   \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess) {                                                                        \
            fprintf(stderr, "[surya_amd] %s:%d %s -> %s\n", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return (int)_e;                                                                            \
        }                                                                                              \
    } while (0)

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;   // v_cvt_pk_bf16_f32 on gfx950, round-to-nearest-even
    return __builtin_bit_cast(bf16_t, b);
}

template <typename T> struct Ty;
template <> struct Ty<float> {
    static constexpr int KE = 32;            // elements per 128-byte K chunk row
    static constexpr int V16 = 4;            // elements per 16 bytes
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
    __device__ static __forceinline__ float rnd(float v) { return v; }   // round through storage type
};
template <> struct Ty<bf16_t> {
    static constexpr int KE = 64;
    static constexpr int V16 = 8;
    __device__ static __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
    __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
    __device__ static __forceinline__ float rnd(float v) { return bf2f(f2bf(v)); }
};

// Unpack a 16-byte register chunk into floats (4 for f32, 8 for bf16).
__device__ __forceinline__ void unpack16(const uint4& v, float (&o)[4], float*) {
    o[0] = __uint_as_float(v.x); o[1] = __uint_as_float(v.y); o[2] = __uint_as_float(v.z); o[3] = __uint_as_float(v.w);
}
__device__ __forceinline__ void unpack16(const uint4& v, float (&o)[8], bf16_t*) {
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
    o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
    o[4] = __uint_as_float(v.z << 16); o[5] = __uint_as_float(v.z & 0xffff0000u);
    o[6] = __uint_as_float(v.w << 16); o[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ uint32_t pack2(float a, float b) { return (uint32_t)f2bf(a) | ((uint32_t)f2bf(b) << 16); }

// Store 4 consecutive outputs.
__device__ __forceinline__ void store4(float* p, float a, float b, float c, float d) {
    *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
}
__device__ __forceinline__ void store4(bf16_t* p, float a, float b, float c, float d) {
    *reinterpret_cast<uint2*>(p) = make_uint2(pack2(a, b), pack2(c, d));
}
__device__ __forceinline__ void store2(float* p, float a, float b) { *reinterpret_cast<float2*>(p) = make_float2(a, b); }
__device__ __forceinline__ void store2(bf16_t* p, float a, float b) { *reinterpret_cast<uint32_t*>(p) = pack2(a, b); }
__device__ __forceinline__ void load4(const float* p, float (&o)[4]) {
    float4 v = *reinterpret_cast<const float4*>(p); o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
__device__ __forceinline__ void load4(const bf16_t* p, float (&o)[4]) {
    uint2 v = *reinterpret_cast<const uint2*>(p);
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
    o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
}

// Cross-lane sums on the VALU (DPP) instead of __shfl_xor, which hipcc lowers to ds_bpermute_b32 (an LDS round trip of
// ~100 cycles per step: the decode-attention key loop spent 13 of its 24 us in those chains, tools/microbench).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// sum over each aligned group of 4 lanes; every lane of the group gets the total
__device__ __forceinline__ float quad_sum(float v) {
    v += dpp_mov<0xB1>(v);        // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);        // quad_perm [2,3,0,1]
    return v;
}
// sum over each aligned group of 16 lanes (one DPP row); every lane of the row gets the total
__device__ __forceinline__ float row16_sum(float v) {
    v = quad_sum(v);
    v += dpp_mov<0x141>(v);       // row_half_mirror: lane i <-> 7 - i of its 8-lane half (quad sums are uniform per quad)
    v += dpp_mov<0x140>(v);       // row_mirror: lane i <-> 15 - i of its row
    return v;
}

// ---- MX quantisation helpers (producers of an MXFP8 GEMM operand: gemm_mx.h's SwiGLU epilogue, kernels.h, decode_attn.h) -----------
// Scale of a block = the smallest power of two s with absmax / s <= 448 (the e4m3 maximum): nothing saturates. (The OCP
// recipe floor(log2 absmax) - 8 can leave values in (448, 512) s that clip.) E8M0 byte = exponent + 127; an all-zero block
// gets byte 0.
__device__ __forceinline__ int mx_block_exp(float absmax) {
    if (!(absmax > 0.f)) return -127;
    int ex;
    const float f = frexpf(absmax, &ex);             // absmax = f * 2^ex, f in [0.5, 1); 448 = 0.875 * 2^9
    const int e = (f <= 0.875f) ? ex - 9 : ex - 8;
    return max(-127, min(127, e));
}
// 4 floats (already divided by the block scale) -> 4 e4m3 bytes, round to nearest even (v_cvt_pk_fp8_f32, OCP on gfx950)
__device__ __forceinline__ uint32_t mx_pack4(float a, float b, float c, float d) {
    int r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    r = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
    return (uint32_t)r;
}
template <int CTRL>
__device__ __forceinline__ float dpp_max(float v) { return fmaxf(v, dpp_mov<CTRL>(v)); }
__device__ __forceinline__ float quad_max(float v) { return dpp_max<0x4E>(dpp_max<0xB1>(v)); }
__device__ __forceinline__ float oct_max(float v) { return dpp_max<0x141>(quad_max(v)); }      // aligned groups of 8 lanes
// This lane owns 4 consecutive elements; the aligned group of 8 lanes owns one 32-element MX block (all 8 lanes active).
__device__ __forceinline__ uint32_t mx_quant4_oct(const float (&v)[4], int& e8) {
    const float m = oct_max(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    const int e = mx_block_exp(m);
    e8 = e + 127;
    return mx_pack4(ldexpf(v[0], -e), ldexpf(v[1], -e), ldexpf(v[2], -e), ldexpf(v[3], -e));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// SiLU inside the bf16 GEMM epilogues (EPI_SWIGLU): the reciprocal instruction (v_rcp_f32, 1 ulp) instead of an IEEE division (~10
// instructions; 64 of them per lane and 256 x 256 tile: the SiLU arithmetic was ~5k of a 58k-cycle tile, tools/microbench/p8_timing.hip).
// The result is rounded to bf16 right after, and EVERY bf16 tile shape uses this one function, so a row's value still does not depend on the
// tile that computed it; fp32 reference mode keeps the division.
template <typename TI>
__device__ __forceinline__ float silu_epi(float x) {
    if constexpr (sizeof(TI) == 2) return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x));
    else return silu_f(x);
}
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// Exact-erf GELU inside the bf16 GEMM epilogues (EPI_GELU: the Swin MLPs of the layout / table encoders, the recogniser's merger): erff is a
// ~50-instruction piecewise polynomial, 128 of them per lane and 256 x 256 tile -- more vector-ALU time than a K = 512 tile's whole K loop
// has matrix time. Abramowitz-Stegun 7.1.26 instead: erfc(z) = t (a1 + t (a2 + t (a3 + t (a4 + t a5)))) exp(-z^2), t = 1 / (1 + p z), |error| <
// 1.5e-7 absolute on erf -- three orders below a bf16 rounding step of the result; x < 0 takes 1 + erf(x) = erfc(|x|) directly (no
// cancellation). Every bf16 tile shape uses this one function; fp32 reference mode keeps erff.
template <typename TI>
__device__ __forceinline__ float gelu_epi(float x) {
    if constexpr (sizeof(TI) == 2) {
        const float z = fabsf(x) * 0.70710678118654752440f;
        const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
        const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
        const float erfc_abs = poly * __expf(-z * z);                       // erfc(|x| / sqrt 2) in (0, 1]
        return 0.5f * x * (x < 0.f ? erfc_abs : 2.0f - erfc_abs);
    } else {
        return gelu_erf_f(x);
    }
}
// GELU, tanh approximation (torch gelu(approximate="tanh") / transformers gelu_pytorch_tanh)
__device__ __forceinline__ float gelu_tanh_f(float x) { return 0.5f * x * (1.0f + tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x))); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }
// Hardswish x . relu6(x + 3) / 6 as x . clamp(x / 6 + 0.5, 0, 1): ONE v_fma_f32 with the clamp output modifier + one multiply (the literal form is add, max,
// min, multiply, multiply -- and its epilogues are where the detector's kernels are vector-ALU-bound). Two fp32 roundings instead of three; differs from
// the literal form by <= 1 ulp of fp32 before any bf16 rounding (fp32 mode stays <= 1e-4 from the oracle, tests/test_gpu_det.py). EVERY kernel takes its
// Hardswish from here or from hardswish_pk below, so the fused forms keep repeating the op list's bits.
__device__ __forceinline__ float hardswish_f(float x) {
    float t;
    asm("v_fma_f32 %0, %1, %2, 0.5 clamp" : "=v"(t) : "v"(x), "s"(1.0f / 6.0f));
    return x * t;
}
typedef float f32x2_hs __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_hs hardswish_pk(f32x2_hs x) {      // the same two operations on an fp32 pair (v_pk_fma_f32 ... clamp, v_pk_mul_f32)
    f32x2_hs t;
    const f32x2_hs c = {1.0f / 6.0f, 1.0f / 6.0f};
    asm("v_pk_fma_f32 %0, %1, %2, 0.5 op_sel_hi:[1,0,0] clamp" : "=v"(t) : "v"(x), "s"(c));
    return x * t;
}

// Launch-policy knobs, in ONE place. Defaults are the measured best; surya_set_tuning(key, value) changes
// them at run time for A/B sweeps inside one process (tools/microbench/decode_sweep.py). Nothing in a launch path reads the
// environment. Round 2 also swept larger decode tiles (128x64 / 128x128 for gate|up and split-K), a 256x128 / deeper-ring
// lm_head and two-half dual-stream decode through knobs that lived here; all lost (profiles/r02_decode_sweeps.md) and their
// code paths were removed with them.
struct Tuning {
    int graph = 0;           // decode steps as hipGraph replays (1) or plain launches (0; faster with the pipelined host loop)
    int split_target = 256;  // decode split-K projections: aim at this many workgroups
    int split_min_kt = 4;    // at least this many 128-byte K-tiles per slice
    int split_max = 8;       // slice cap (the reduce kernels keep <= 8 slabs in flight)
    int bigtile = 3;         // 256x256 tiles for large bf16 GEMMs: 3 = the 8-phase schedule (round 5; even K-tile counts, else the 2-stage loop), 1 = 2-stage loop, 2 = 2-stage on 4 waves, 0 = 128x128 only
    int bigtile_any = 0;     // 1: every M > 256 bf16 GEMM takes the 256x256 tile whatever its round count (tests / race screens of the tile on small shapes)
    int big_m_split = -1;    // decode projections above 256 rows (round 6): split-K tile -- -1 = by workgroup count (128x128 once it fills half the chip, else 128x64 within one round; 4-stage ring),
                             // 0 = the 64x64 tile of the <= 256-row regime, 2 = 128x128, 3 = 128x64 (slice count unchanged: same bits in every arm)
    int big_m_gateup = -1;   // decode gate|up above 256 rows: -1 = by rows (8-phase 256x256 tile at 4-6 row blocks, else the generic choice), 0 = generic,
                             // 1 = persistent 8-phase loop, 2 = 8-phase tile per workgroup
    int gateup_ring = 2;     // decode gate|up (64x64 tiles, M in (128, 256]): LDS stages of its direct-to-LDS loop (2 = unrolled pair, 3 / 4 = ring with counted vmcnt)
    int conv_persist = 3;    // 3 x 3 / 5 x 5 convolutions on 256x256 tiles on the persistent 8-phase loop with the gather in its request stream: bit 0 = Cin % 64 == 0 (a K-tile is one tap), bit 1 = Cin == 32 (two taps per K-tile); 0 = one-tile 2-stage kernel
    int dwconv_pipe = 2;     // depthwise convolutions: 2 = round-3 kernel with its index split by host-made reciprocals (64-bit % and / were ~700 instructions per thread; +1 % on the forward), 1 = all loads unconditional (hipcc hoists them all: 2 waves per SIMD, -4.5 %), 0 = round-3 kernel
    int conv_lean = 1;       // implicit-GEMM convolutions whose K-tiles align with filter taps: gather state precomputed per workgroup (0 = round-4 per-request arithmetic)
    int bigtile_ratio_pct = 0;  // tile choice: assumed throughput of a full round of 256x256 tiles over one of 128x128 tiles, in percent (0 = built-in: 140 for the 8-phase loop, 117 otherwise)
    int bigtile_min_k = 0;   // ... only when K >= this (short-K GEMMs are prologue / epilogue bound: two 128x128 workgroups per CU overlap those)
    int glds = 2;            // LDS stages of the 128x128 direct-to-LDS GEMM (2 or 3)
    // round 4 (A/B knobs of the decode step's non-GEMM kernels and of the persistent big-tile GEMM; defaults = measured best)
    int dattn = 4;           // bf16 decode attention: 4 = thread-local prologue (decode_attn_flash2_kernel), 3 = third version
    int rnorm = 2;           // split-K reduce + residual + RMSNorm: 2 = slab loads sized by the slice count, 1 = round-3 kernel, 3 = one wave per row
    int ghead = 2;           // greedy head: 2 = registers-only partial reduce + vector bbox head (+ next step's embedding when fused), 1 = round-3 kernel
    int fuse_embed = 1;      // inner decode steps: the greedy head also writes the next step's embedding + first RMSNorm (no embed launch)
    int kvprefetch = 0;      // 1: the two reduce kernels of a decode layer request the next layer's V / K cache rows (extra workgroups, kernels.h KvPrefetch).
                             // Measured slower (gpurun r04d, 256 slots: 1165 us per step on vs 1100 off): the extra workgroups delay the reduce rows more than the warm L2 saves
    int lmhead = 1;          // lm_head (N >= 32768, M <= 256): 1 = one round of 256x320 tiles with the greedy partials taken from the accumulators, 0 = 128x128 tiles
    int dattn_db = 0;        // bf16 decode attention with two K/V tile buffers (decode_attn_flash2_kernel<.., true>: next tile's fetch overlaps this tile's
                             // compute): 0 = when some active slot's context exceeds one 128-key tile (host bound; eager launches only -- under graph replay, `graph` = 1, the single-buffer kernel runs and the bound is not advanced), 1 = always, -1 = never
    int lay_ln = 1;          // layout / table encoder LayerNorm (bf16): 1 = rows held in registers by C / 8 lanes (layernorm_rows_bf16_kernel), 0 = a wave per row
    int det_head_blk = 1;    // detector's folded decode head: 1 = register-blocked sum + classify (4 x 2 pixel blocks), 0 = per-pixel kernel
    int det_fuse = 1023;     // detector's fused forms (det_model.hip find_fusions; bf16): bit 0 = LiteMLA depthwise 5x5 + grouped 1x1, bit 1 = LiteMLA kv + out in one
                             // launch, bit 2 = z0 inside the head's sum + classify pass, bit 3 = MBConv depthwise 3x3 + projection, bit 4 = FusedMBConv
                             // 3x3 + Hardswish + projection, bit 5 = stem convolutions on the patch-in-LDS kernel, bit 6 = whole MBConv blocks (expand + depthwise +
                             // projection) in one launch at the stride-2 transitions (det_mbconv.h), bit 7 = (with bit 2) the folded head entirely on the matrix cores
                             // (det_head.h), bit 8 = (with bit 5) the first convolution reads the caller's pixels itself (no input-layout launch), bit 9 = the stem's residual block (two 3x3
                             // convolutions) in one launch with the tensor between them in LDS; 0 = the op list as written (the checker of
                             // tests/test_gpu_det_fused.py)
    int det_up4 = 1;         // detector's x4 output up-sampling: 1 = a 4 x 4 output block per thread from its 3 x 3 source neighbourhood (upsample_planes_x4_kernel), 0 = upsample_planes4_kernel (the checker: same bits)
    int fmb_chunk = 64;      // FusedMBConv kernel, Cin = 64 stride 1: mid channels per chunk -- 64 = two workgroups per CU (the chunk epilogue of one beside the
                             // MFMAs of the other), 128 = one workgroup per CU with twice the accumulators
    int persist = 1;         // 256x256 bf16 GEMMs (>= 4 even K-tiles) as the PERSISTENT 8-phase loop (gemm_nt_p8p_kernel: the half-tile ring runs on across
                             // tiles, wave-private epilogue): 1 = on (round 5: +3...10 % per encoder / prefill shape once the two wave groups were
                             // re-aligned around the epilogue, +0.9 % on the bench's recognition leg); 0 = one tile per workgroup
};
inline Tuning& tuning() { static Tuning t; return t; }
inline int& tuning_epoch() { static int e = 0; return e; }   // bumped by surya_set_tuning whenever a knob changes value

// Debug aid: SURYA_AMD_POISON=1 fills every engine arena with 0xFF bytes (NaN in bf16 and fp32, -1 as an index) right after its
// allocation, so a read of memory the engine never wrote shows up as NaN / a fault on every box instead of depending on what the
// last tenant left in HBM (gpurun r04f: three layout tests failed on one box and on no other). Off: hipMalloc'ed memory as it comes.
inline void poison_arena(void* p, size_t bytes) {
    static const bool on = [] { const char* e = getenv("SURYA_AMD_POISON"); return e && e[0] == '1'; }();
    if (on) { (void)hipMemset(p, 0xFF, bytes); (void)hipDeviceSynchronize(); }
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per function AND per device: remember, per device, the largest size
// already granted, and raise it when a later launch of the same kernel needs more (kernels whose dynamic LDS depends on the
// problem -- post_boxes_kernel sizes it by the page height -- would otherwise keep the first call's limit).
struct AttrOnce {
    size_t granted[64] = {};
    template <typename F> void ensure(F kern, size_t lds) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        size_t& g = granted[dev & 63];
        if (lds > g || g == 0) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            g = lds > g ? lds : g;
            if (g == 0) g = 1;
        }
    }
};

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline long cdivl(long a, long b) { return (a + b - 1) / b; }

}  // namespace sa
