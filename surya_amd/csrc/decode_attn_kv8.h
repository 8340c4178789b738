// Decode-step attention over an FP8 (OCP e4m3) KV cache: the long-horizon variant of decode_attn_flash_kernel (decode_attn.h).
//
// Why: at the texify horizon (prompt 202 -> 970 cached tokens, 128 crops, BASELINE.json configs[4]) the bf16 flash kernel is the
// largest kernel of the decode step (23 us per layer, 29 % of the step; profiles/r03_f_texify768_kernel_stats.md) and VERDICT r02
// asked for an fp8 cache there. The reference has no fp8 counterpart (its only cache option is HQQ 8-bit,
// surya/recognition/__init__.py:379-395); the format is pinned by oracle/mx_oracle.py::kv8_quantize.
// What it bought, measured at the 768-token horizon (profiles/r03_h_*, r03_m_texify768_*): bf16 flash kernel 23.0 us per layer ->
//   v1 (4 waves, 32 x 32 MFMA, two 128-key tile buffers)                  22.4 us   -- half the bytes, same time: not byte-bound
//   v2 (the same with four buffers, three tiles in flight)                 20.7 us   -- not latency-bound either: ~3.6 us of serial work per tile
//   v3 (this file: 8 waves = two per SIMD, 16 x 16 x 32 MFMA, 256-key tiles) 16.8 us -- compute per tile down to ~1.2 us; what remains is the
//      HBM side: 256 workgroups x 64 KB per tile are served at ~2.9 TB/s, and the first tile's burst is ~45 % of the kernel at 586 tokens
//   v4 (tried after v3, not kept: 128-key tiles in four buffers, 16 keys per wave, P V on 16 x 16 x 16 MFMAs) 17.4-18.1 us -- the smaller
//      first burst shortens the prologue (19.7k -> 16.4k cycles) but the per-tile fixed work of a wave (scale reads, two ds_bpermute for
//      the shared maximum, exp, 32 accumulator rescales, 8 P V MFMAs) is paid twice as often: 3.2k cycles per 128 keys vs 2.9k per 256
// texify run (128 crops x 768 tokens): 1.17x over bf16 with MXFP8 weights alone -> 1.29-1.30x with this cache on top.
//
// Format ("KV8"): per (slot, kv head, token) ONE power-of-two scale 2^e, the smallest with absmax / 2^e <= 448 (mx_block_exp:
// an MX block that spans the head dim), elements round-to-nearest-even to e4m3. Dequantised values are exact in bf16, so the
// kernel converts bytes -> bf16 in registers (v_cvt_pk_f32_fp8 + a pack) and keeps the bf16 MFMA and the fp32 softmax of the flash
// kernel; the scales never touch an element:
//     score_j = sk_j * (q . k8_j)            -- one multiply per score
//     O      += sum_j (p_j * sv_j) v8_j      -- folded into the P fragment before it is rounded to bf16
// Layout per layer:   k8  [slot][kv head][Tmax][D]            bytes + ksc [slot][kv head][Tmax8] fp32
//                     v8t [slot][kv head][Tmax8 / 128][D][128] bytes + vsc [slot][kv head][Tmax8] fp32    (Tmax8 = Tmax rounded up to 256)
// V is stored TRANSPOSED inside each 128-token tile so that the V^T fragment of O^T += V^T P^T (8 keys of one output dim per lane) is
// two 4-byte LDS reads, and tile-blocked so that a tile is ONE contiguous 128 D-byte piece of memory (the first version kept whole
// [D][Tmax8] rows: a tile was D separate 128-byte pieces 1 KB apart and the kernel ran no faster than the bf16 one). The caches are
// zero-filled at allocation, so key columns past the context hold finite bytes (P = 0 there).
// A 256-key K + V^T tile pair is 64 KB at D = 128; two buffers: tile t + 1 streams in through global_load_lds while tile t is
// multiplied, and the wait at the top of a tile is a counted s_waitcnt (every wave issues exactly D / 16 + 1 LDS-DMA instructions per
// tile -- whole tiles; rows / chunks past the context re-fetch a needed one, a cache hit -- so the count is a compile-time constant).
// 143 KB of LDS at D = 128: one workgroup of eight waves per CU.
#pragma once
#include "decode_attn.h"

namespace sa {

typedef float f32x2_t __attribute__((ext_vector_type(2)));

// two e4m3 bytes of `src` (low or high half) -> two bf16 in one dword; exact (e4m3 has 3 mantissa bits)
template <bool HI>
__device__ __forceinline__ uint32_t fp8x2_to_bf16x2(uint32_t src) {
    const f32x2_t f = __builtin_amdgcn_cvt_pk_f32_fp8((int)src, HI);
    return (__float_as_uint(f[0]) >> 16) | (__float_as_uint(f[1]) & 0xffff0000u);
}
__device__ __forceinline__ u32x4 fp8x8_to_bf16x8(uint32_t lo, uint32_t hi) {
    u32x4 r;
    r[0] = fp8x2_to_bf16x2<false>(lo); r[1] = fp8x2_to_bf16x2<true>(lo);
    r[2] = fp8x2_to_bf16x2<false>(hi); r[3] = fp8x2_to_bf16x2<true>(hi);
    return r;
}

// Prefill: quantise the prompt's freshly appended bf16 cache rows (rope_kv_append_kernel) into the KV8 arrays. One wave per
// (prompt token, kv head); lanes < D / 4 own 4 elements each.
template <int D>
__global__ __launch_bounds__(256) void kv8_quant_rows_kernel(const bf16_t* __restrict__ kc, const bf16_t* __restrict__ vc,
                                                             const int* __restrict__ tok_slot, const int* __restrict__ tok_pos, int M,
                                                             uint8_t* __restrict__ k8, uint8_t* __restrict__ v8t, float* __restrict__ ksc,
                                                             float* __restrict__ vsc, int nkv, int Tmax, int Tmax8) {
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (item >= M * nkv) return;
    const int t = item / nkv, kvh = item % nkv;
    const int slot = tok_slot[t], pos = tok_pos[t];
    const long rowi = (long)slot * nkv + kvh;
    const bool on = lane < D / 4;
    const int c = on ? lane * 4 : 0;
    float kv[4], vv[4];
    load4(kc + (rowi * Tmax + pos) * D + c, kv);
    load4(vc + (rowi * Tmax + pos) * D + c, vv);
    const float km = wave_max(on ? fmaxf(fmaxf(fabsf(kv[0]), fabsf(kv[1])), fmaxf(fabsf(kv[2]), fabsf(kv[3]))) : 0.f);
    const float vm = wave_max(on ? fmaxf(fmaxf(fabsf(vv[0]), fabsf(vv[1])), fmaxf(fabsf(vv[2]), fabsf(vv[3]))) : 0.f);
    const int ek = mx_block_exp(km), ev = mx_block_exp(vm);
    if (lane == 0) { ksc[rowi * Tmax8 + pos] = ldexpf(1.0f, ek); vsc[rowi * Tmax8 + pos] = ldexpf(1.0f, ev); }
    if (!on) return;
    *reinterpret_cast<uint32_t*>(k8 + (rowi * Tmax + pos) * D + c) = mx_pack4(ldexpf(kv[0], -ek), ldexpf(kv[1], -ek), ldexpf(kv[2], -ek), ldexpf(kv[3], -ek));
    const uint32_t pv = mx_pack4(ldexpf(vv[0], -ev), ldexpf(vv[1], -ev), ldexpf(vv[2], -ev), ldexpf(vv[3], -ev));
    uint8_t* vd = v8t + rowi * D * Tmax8 + ((long)(pos >> 7) * D + c) * 128 + (pos & 127);
    vd[0] = (uint8_t)pv; vd[128] = (uint8_t)(pv >> 8); vd[256] = (uint8_t)(pv >> 16); vd[384] = (uint8_t)(pv >> 24);
}

// Geometry of the shipped version (the third; see the header comment for what the first two measured): 512 threads = 8 waves, so every
// SIMD holds TWO waves whose dependent chains (LDS read -> convert -> MFMA -> softmax) hide each other; 256-key tiles, wave w owns keys
// [32w, 32w + 32) of a tile as two 16-key blocks; v_mfma_f32_16x16x32_bf16: a lane owns head (lane & 15) -- the G <= 8 real heads fill half
// of the 16 columns instead of a sixth of 32 -- and, per 16-key block, keys 4g .. 4g + 3 (g = lane >> 4). Fragments:
//   S^T block = K (16 keys x 32 dims per step) . q^T : A = 8 key bytes [dims 32 step + 8g ..], B = 16 q bytes of the lane's head;
//   O^T      += V^T (16 dims x 32 keys) . P^T        : B = the lane's 8 probabilities (block 0 keys 4g.., block 1 keys 4g..) x v-scale,
//                                                     A = 4 + 4 bytes of dim row (lane & 15) at those keys -- two ds_read_b32 of the V^T tile.
// The per-head maximum is shared by the four g-lanes of a head (two ds_bpermute per tile); the sums stay per lane until the end.
constexpr int KV8_THREADS = 512;

// Phase cycle counters for tools/microbench/decode_attn_kv8_bench.hip (-DSA_DA_TIMING): workgroup (64, 0), thread 0 accumulates shader
// cycles per phase of the tile loop. Compiled out of the product library.
#ifdef SA_DA_TIMING
__device__ unsigned long long sa_kv8_cycles[8];
#define SA_KV8_T(var) const unsigned long long var = __builtin_readcyclecounter();
#define SA_KV8_ACC(i, a, b) if (blockIdx.x == 64 && blockIdx.y == 0 && threadIdx.x == 0) sa_kv8_cycles[i] += (b) - (a);
#else
#define SA_KV8_T(var)
#define SA_KV8_ACC(i, a, b)
#endif

template <int D, int MAXG>
__global__ __launch_bounds__(KV8_THREADS) void decode_attn_kv8_kernel(const float* __restrict__ qkv_part, int S, const bf16_t* __restrict__ qkv_bias,
                                                                      bf16_t* __restrict__ out, uint8_t* __restrict__ k8, uint8_t* __restrict__ v8t,
                                                                      float* __restrict__ ksc, float* __restrict__ vsc,
                                                                      const int* __restrict__ active_slots, const int* __restrict__ row_len,
                                                                      const float2* __restrict__ rope_cs, int nq, int nkv, int Tmax, int Tmax8,
                                                                      float scale, uint8_t* __restrict__ out8 = nullptr,
                                                                      uint8_t* __restrict__ sout = nullptr, int srows = 0) {
    typedef bf16_t T;
    constexpr int NT = KV8_THREADS, NW = NT / 64;
    constexpr int KT = 256;                                  // keys per tile: 32 per wave
    constexpr int KCPR = D / 16;                             // 16-byte chunks per K row
    constexpr int KRPI = 1024 / D;                           // K rows moved by one global_load_lds (64 lanes x 16 B)
    constexpr int KXM = KCPR >= 8 ? 7 : KCPR - 1;            // XOR mask of the K-tile chunk swizzle
    constexpr int QROWB = D * 2, QCPR = D / 8;               // q operand rows are bf16: [16][D], chunk c of head row r at c ^ (r & QXM)
    constexpr int QXM = QCPR >= 16 ? 15 : QCPR - 1;
    constexpr int NKS = D / 32;                              // 32-dim MFMA steps of S^T
    constexpr int NDB = D / 16;                              // 16-dim output blocks of O^T
    constexpr int CW = D + 4;
    constexpr int KTILE = KT * D, VTILE = D * KT, VSUB = D * 128;     // bytes; a V^T tile = two 128-token sub-tiles [D][128]
    constexpr int NBUF = 2;
    constexpr int PER = D / 16 + 1;                          // LDS-DMA instructions per wave per tile: K D/32, V^T D/32, scales 1
    constexpr size_t QX = (size_t)16 * D * 2 + (size_t)(MAXG + 2) * D * 4;
    static_assert(D % 32 == 0 && D <= 128 && MAXG <= 16 && 1024 % D == 0 && (KT / KRPI) % NW == 0 && (2 * D / 8) % NW == 0, "geometry");
    static_assert((size_t)NW * MAXG * CW * 4 <= (size_t)NBUF * KTILE, "the combine records alias the K buffers");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Ks = smem;                                // [NBUF][KT][D] bytes, 16-byte chunk c of row r at c ^ (r & KXM)
    unsigned char* Vs = Ks + NBUF * KTILE;                   // [NBUF][2][D][128] bytes (V^T), chunk c of dim row d at c ^ (d & 7)
    float* Sc = reinterpret_cast<float*>(Vs + NBUF * VTILE); // [NBUF][2][KT] k | v scales of the tile's keys
    T* qT = reinterpret_cast<T*>(Sc + NBUF * 2 * KT);        // [16][D] q heads (rows >= G are zero)
    float* xrow = reinterpret_cast<float*>(qT + 16 * D);     // [(MAXG + 2) * D]
    unsigned char* new8 = reinterpret_cast<unsigned char*>(qT) + QX;      // [2][D] quantised new k | v row
    float* nscale = reinterpret_cast<float*>(new8 + 2 * D);  // [2] their scales
    float* comb = reinterpret_cast<float*>(Ks);              // [NW][MAXG][CW], aliases the K buffers after the key loop

    SA_KV8_T(tk0)
    const int G = nq / nkv;
    const int a = blockIdx.x, kvh = blockIdx.y;
    const int slot = active_slots[a];
    const int len = row_len[a];                              // cached tokens; the new token sits at index len
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int total = len + 1;
    const long rowi = (long)slot * nkv + kvh;
    const unsigned char* kb = k8 + rowi * Tmax * D;
    const unsigned char* vb = v8t + rowi * D * Tmax8;
    const float* kscb = ksc + rowi * Tmax8;
    const float* vscb = vsc + rowi * Tmax8;
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

    // Every wave issues exactly PER instructions per tile (whole tiles; K rows past the context are clamped duplicates, V^T / scale
    // columns past it are whatever the zero-filled arrays hold -- finite, and multiplied by P = 0; Tmax8 is a multiple of KT).
    auto issue_tile = [&](int base, int buf) {
        const int rows16 = min(KT, (total - base + 15) & ~15);            // keys of this tile a 16-key MFMA block can touch
#pragma unroll
        for (int i = 0; i < D / 32; ++i) {                                // K: KT / KRPI groups of KRPI rows, dealt to the waves
            const int g = wave + NW * i;
            const int r = g * KRPI + lane / KCPR, pc = lane % KCPR;
            const long j = min(base + r, max(len - 1, 0));
            __builtin_amdgcn_global_load_lds((gptr_t)(kb + j * D + ((pc ^ (r & KXM)) << 4)), (lptr_t)(Ks + buf * KTILE + g * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < D / 32; ++i) {                                // V^T: 2 sub-tiles x D / 8 groups of 8 dim rows x 128 key bytes
            const int gg = wave + NW * i, sub = gg / (D / 8), g = gg % (D / 8);
            const int r = g * 8 + (lane >> 3), lc = (lane & 7) ^ (r & 7);  // this lane's LDS position holds logical chunk lc
            // 16-key chunks past the context are never multiplied by a non-zero P: fetch a needed chunk again instead (a cache hit, no
            // HBM bytes) -- the instruction count per wave stays PER, which the counted waits rely on
            const int need = min(128, max(rows16 - sub * 128, 0)) >> 4;      // chunks of this sub-tile that hold visible keys
            const int sub_s = need > 0 ? sub : 0, lc_s = min(lc, max((need > 0 ? need : min(128, rows16) >> 4) - 1, 0));
            __builtin_amdgcn_global_load_lds((gptr_t)(vb + ((long)(base >> 7) + sub_s) * VSUB + r * 128 + (lc_s << 4)),
                                             (lptr_t)(Vs + buf * VTILE + sub * VSUB + g * 1024), 16, 0, 0);
        }
        // scales: waves 0..3 -> k scales of keys 64 w .. + 63, waves 4..7 -> v scales
        __builtin_amdgcn_global_load_lds((gptr_t)((wave < 4 ? kscb : vscb) + base + (wave & 3) * 64 + lane),
                                         (lptr_t)(Sc + buf * 2 * KT + wave * 64), 4, 0, 0);
    };
    const int hn = lane & 15, g4 = lane >> 4;                // this lane's head (column of S^T) and key / dim group
    const int n_tiles = (total + KT - 1) / KT;

    // ---- prologue: q/k/v of this row (split-K slabs + bias); its loads go out before the first tile's LDS-DMA (vector memory returns
    // in order). Phase counters (tools/microbench/decode_attn_kv8_bench.hip): with >= 2 tiles the prologue is ~19k of the kernel's 42k
    // cycles at 586 cached tokens IN EITHER ORDER -- the issue of tile 1 blocks until tile 0 has drained, and 256 workgroups asking for
    // their first 64 KB at once are a 16.7 MB burst that HBM serves at ~3 TB/s (~13k cycles). The tile loop itself: ~2.9k cycles of
    // compute, ~1k of barrier skew and ~2k of blocked issue per 256-key tile.
    const int qkv_dim = (nq + 2 * nkv) * D;
    const int Mrows = gridDim.x;
    const int half = D / 2;
    constexpr int NI = ((MAXG + 2) * D + NT - 1) / NT;
    const int n_items = (G + 2) * D;
    float p8[NI][8], bias_v[NI];
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const int it = min(tid + k * NT, n_items - 1);
        const int hh = it / D, i = it % D;
        const int col = (hh < G ? (kvh * G + hh) * D : (hh == G ? (nq + kvh) * D : (nq + nkv + kvh) * D)) + i;
#pragma unroll
        for (int sidx = 0; sidx < 8; ++sidx) p8[k][sidx] = qkv_part[((long)min(sidx, S - 1) * Mrows + a) * qkv_dim + col];
        bias_v[k] = Ty<T>::ld(qkv_bias + col);
    }
    const float4 cs2 = *reinterpret_cast<const float4*>(rope_cs + (long)len * half + (tid % (half / 2)) * 2);   // (cos, sin) of dims i, i + 1
    issue_tile(0, 0);
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        float val = bias_v[k];
#pragma unroll
        for (int sidx = 0; sidx < 8; ++sidx) val += (sidx < S) ? p8[k][sidx] : 0.f;
        if (tid + k * NT < n_items) xrow[tid + k * NT] = Ty<T>::rnd(val);
    }
    for (int i = tid; i < (16 - G) * QCPR; i += NT)
        *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned char*>(qT) + G * QROWB + i * 16) = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
    // RoPE (decoder/__init__.py:60-84, cos/sin rounded to the storage dtype): q -> qT (scaled, chunk-swizzled), k -> back into xrow
    auto qput2 = [&](int hh, int e, float v0, float v1) {
        const int c = e / 8, w = e % 8;
        store2(reinterpret_cast<T*>(reinterpret_cast<unsigned char*>(qT) + hh * QROWB + ((c ^ (hh & QXM)) << 4)) + w, v0, v1);
    };
    for (int it = tid; it < (G + 1) * (half / 2); it += NT) {
        const int i = (it % (half / 2)) * 2, hh = it / (half / 2);
        const float xa1 = xrow[hh * D + i], xa2 = xrow[hh * D + i + half], xb1 = xrow[hh * D + i + 1], xb2 = xrow[hh * D + i + 1 + half];
        const float ya1 = Ty<T>::rnd(xa1 * cs2.x - xa2 * cs2.y), ya2 = Ty<T>::rnd(xa2 * cs2.x + xa1 * cs2.y);
        const float yb1 = Ty<T>::rnd(xb1 * cs2.z - xb2 * cs2.w), yb2 = Ty<T>::rnd(xb2 * cs2.z + xb1 * cs2.w);
        if (hh < G) {
            qput2(hh, i, ya1 * scale, yb1 * scale); qput2(hh, i + half, ya2 * scale, yb2 * scale);
        } else {                                             // the thread owns these four dims of the k row: in place
            xrow[G * D + i] = ya1; xrow[G * D + i + 1] = yb1; xrow[G * D + i + half] = ya2; xrow[G * D + i + 1 + half] = yb2;
        }
    }
    __syncthreads();
    // quantise the new token's k (wave 0) and v (wave 1) rows: scale, bytes -> LDS (for the tile that holds index len) and the caches
    if (wave < 2) {
        const float* src = xrow + (G + wave) * D;
        const bool on = lane < D / 4;
        const int c = on ? lane * 4 : 0;
        const float x0 = src[c], x1 = src[c + 1], x2 = src[c + 2], x3 = src[c + 3];
        const float m = wave_max(on ? fmaxf(fmaxf(fabsf(x0), fabsf(x1)), fmaxf(fabsf(x2), fabsf(x3))) : 0.f);
        const int e = mx_block_exp(m);
        const float sc = ldexpf(1.0f, e);
        const uint32_t pk = mx_pack4(ldexpf(x0, -e), ldexpf(x1, -e), ldexpf(x2, -e), ldexpf(x3, -e));
        if (lane == 0) { nscale[wave] = sc; (wave == 0 ? ksc : vsc)[rowi * Tmax8 + len] = sc; }
        if (on) {
            *reinterpret_cast<uint32_t*>(new8 + wave * D + c) = pk;
            if (wave == 0) *reinterpret_cast<uint32_t*>(k8 + (rowi * Tmax + len) * D + c) = pk;
            else {
                uint8_t* vd = v8t + rowi * D * Tmax8 + ((long)(len >> 7) * D + c) * 128 + (len & 127);
                vd[0] = (uint8_t)pk; vd[128] = (uint8_t)(pk >> 8); vd[256] = (uint8_t)(pk >> 16); vd[384] = (uint8_t)(pk >> 24);
            }
        }
    }
    if (1 < n_tiles) issue_tile(KT, 1);                      // block-uniform
    __syncthreads();                                         // qT / new8 / nscale are visible

    // ---- per-wave flash attention over keys [32 * wave, 32 * wave + 32) of every tile
    u32x4 qf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
        qf[ks] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(qT) + hn * QROWB + (((ks * 4 + g4) ^ (hn & QXM)) << 4));
    f32x4 oacc[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db) oacc[db] = f32x4{0.f, 0.f, 0.f, 0.f};
    float mrun = -INFINITY, lrun = 0.f;                      // mrun: shared by the four g-lanes of a head; lrun: this lane's keys only
    const int new_tile = len / KT, new_row = len % KT;
    const float nsk = nscale[0], nsv = nscale[1];
    const int k0 = wave * 32;
    SA_KV8_T(tl0)
    SA_KV8_ACC(0, tk0, tl0)
    for (int t = 0; t < n_tiles; ++t) {
        const int base = t * KT, nk = min(KT, total - base), cur = t & 1;
        SA_KV8_T(ta)
        // tile t has landed once at most (the one tile in flight behind it) x PER of this wave's LDS-DMA instructions are outstanding
        if (t + 1 < n_tiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SA_KV8_T(tb)
        __syncthreads();                                     // ... everyone's share of it
        SA_KV8_T(tc)
        SA_KV8_ACC(1, ta, tb)
        SA_KV8_ACC(2, tb, tc)
        if (new_tile == t) {                                 // the new token's row falls into this tile
            unsigned char* Kc = Ks + cur * KTILE;
            unsigned char* Vc = Vs + cur * VTILE + (new_row >> 7) * VSUB;
            const int col = new_row & 127;
            if (tid < KCPR)
                *reinterpret_cast<u32x4*>(Kc + new_row * D + ((tid ^ (new_row & KXM)) << 4)) = *reinterpret_cast<const u32x4*>(new8 + tid * 16);
            if (tid >= 64 && tid < 64 + D) {
                const int dd = tid - 64;
                Vc[dd * 128 + (((col >> 4) ^ (dd & 7)) << 4) + (col & 15)] = new8[D + dd];
            }
            if (tid == 32) { Sc[cur * 2 * KT + new_row] = nsk; Sc[cur * 2 * KT + KT + new_row] = nsv; }
            __syncthreads();
        }
        if (k0 < nk) {                                       // wave-uniform
            const unsigned char* Kc = Ks + cur * KTILE;
            const unsigned char* Vc = Vs + cur * VTILE + (k0 >> 7) * VSUB;
            const int vcol = k0 & 127;                       // this wave's first key inside its 128-token sub-tile
            f32x4 sk[2], sv[2];                              // scales of this lane's keys: block kb, keys k0 + 16 kb + 4 g4 + i
#pragma unroll
            for (int kbk = 0; kbk < 2; ++kbk) {
                sk[kbk] = *reinterpret_cast<const f32x4*>(Sc + cur * 2 * KT + k0 + 16 * kbk + 4 * g4);
                sv[kbk] = *reinterpret_cast<const f32x4*>(Sc + cur * 2 * KT + KT + k0 + 16 * kbk + 4 * g4);
            }
            f32x4 sacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const int q8i = ks * 4 + g4;                 // 8-byte piece of a key row: 16-byte chunk q8i / 2, half q8i % 2
#pragma unroll
                for (int kbk = 0; kbk < 2; ++kbk) {
                    const int krow = k0 + 16 * kbk + hn;
                    const uint2 kb8 = *reinterpret_cast<const uint2*>(Kc + krow * D + (((q8i >> 1) ^ (krow & KXM)) << 4) + (q8i & 1) * 8);
                    const u32x4 kf = fp8x8_to_bf16x8(kb8.x, kb8.y);
                    sacc[kbk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, qf[ks]), sacc[kbk], 0, 0, 0);
                }
            }
            float bm = -INFINITY;
#pragma unroll
            for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float sv_ = (k0 + 16 * kbk + 4 * g4 + i < nk) ? sacc[kbk][i] * sk[kbk][i] : -INFINITY;
                    sacc[kbk][i] = sv_;
                    bm = fmaxf(bm, sv_);
                }
            bm = fmaxf(bm, __shfl_xor(bm, 16, 64));
            bm = fmaxf(bm, __shfl_xor(bm, 32, 64));          // finite: key k0 is visible and belongs to g4 = 0
            const float mnew = fmaxf(mrun, bm);
            const float alpha = __expf(mrun - mnew);         // exp(-inf) = 0 on the first tile
            mrun = mnew;
            float psum = 0.f;
#pragma unroll
            for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float pv = __expf(sacc[kbk][i] - mnew);
                    sacc[kbk][i] = pv;
                    psum += pv;
                }
            lrun = lrun * alpha + psum;
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int i = 0; i < 4; ++i) oacc[db][i] *= alpha;
            u32x4 pf;                                        // P * v-scale rounded to bf16: element i <-> key 4 g4 + i of block 0, i + 4 <-> of block 1
            pf[0] = pack2(sacc[0][0] * sv[0][0], sacc[0][1] * sv[0][1]);
            pf[1] = pack2(sacc[0][2] * sv[0][2], sacc[0][3] * sv[0][3]);
            pf[2] = pack2(sacc[1][0] * sv[1][0], sacc[1][1] * sv[1][1]);
            pf[3] = pack2(sacc[1][2] * sv[1][2], sacc[1][3] * sv[1][3]);
            const int c0 = vcol >> 4;                        // 16-byte chunk of block 0 in the dim rows; block 1 is the next chunk
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
                const int dd = db * 16 + hn;
                const unsigned char* vrow = Vc + dd * 128 + g4 * 4;
                const uint32_t lo = *reinterpret_cast<const uint32_t*>(vrow + ((c0 ^ (dd & 7)) << 4));
                const uint32_t hi = *reinterpret_cast<const uint32_t*>(vrow + (((c0 + 1) ^ (dd & 7)) << 4));
                const u32x4 vf = fp8x8_to_bf16x8(lo, hi);
                oacc[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, vf), __builtin_bit_cast(bf16x8, pf), oacc[db], 0, 0, 0);
            }
        }
        SA_KV8_T(td)
        __syncthreads();                                     // every wave is done with buffer `cur`
        SA_KV8_T(te)
        if (t + 2 < n_tiles) issue_tile(base + 2 * KT, cur);
        SA_KV8_T(tf)
        SA_KV8_ACC(3, tc, td)
        SA_KV8_ACC(4, td, te)
        SA_KV8_ACC(5, te, tf)
    }
    SA_KV8_T(tl1)
    SA_KV8_ACC(6, tl0, tl1)
    // ---- split-KV combine of the eight waves (the K buffers are free: comb aliases them)
    float ltot = lrun + __shfl_xor(lrun, 16, 64);
    ltot += __shfl_xor(ltot, 32, 64);
    if (hn < G) {
        float* rec = comb + (wave * MAXG + hn) * CW;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
            *reinterpret_cast<float4*>(rec + db * 16 + g4 * 4) = make_float4(oacc[db][0], oacc[db][1], oacc[db][2], oacc[db][3]);
        if (g4 == 0) { rec[D] = mrun; rec[D + 1] = ltot; }
    }
    __syncthreads();
    if (tid < G * (D / 4)) {
        const int oh = tid / (D / 4), od = (tid % (D / 4)) * 4;
        float mx = -INFINITY;
#pragma unroll
        for (int w = 0; w < NW; ++w) mx = fmaxf(mx, comb[(w * MAXG + oh) * CW + D]);
        float num[4] = {0.f, 0.f, 0.f, 0.f}, den = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const float* rec = comb + (w * MAXG + oh) * CW;
            const float e = (rec[D] == -INFINITY) ? 0.f : __expf(rec[D] - mx);       // waves past the context hold (-inf, 0, 0)
            const float4 o4 = *reinterpret_cast<const float4*>(rec + od);
            num[0] += e * o4.x; num[1] += e * o4.y; num[2] += e * o4.z; num[3] += e * o4.w;
            den += e * rec[D + 1];
        }
        const float inv = 1.0f / den;
        const long o_off = (long)a * nq * D + (long)(kvh * G + oh) * D + od;
        store4(out + o_off, num[0] * inv, num[1] * inv, num[2] * inv, num[3] * inv);
        if (out8) {   // MXFP8 copy for the fp8 o-projection (gemm_mx.h): 8 adjacent threads own one 32-wide block of a head
            const float q[4] = {Ty<T>::rnd(num[0] * inv), Ty<T>::rnd(num[1] * inv), Ty<T>::rnd(num[2] * inv), Ty<T>::rnd(num[3] * inv)};
            int e8;
            const uint32_t pk = mx_quant4_oct(q, e8);
            *reinterpret_cast<uint32_t*>(out8 + o_off) = pk;
            const int col = (kvh * G + oh) * D + od;             // K-tile-major scales (gemm_mx.h): [col / 128][srows][4]
            if ((tid & 7) == 0) sout[((long)(col >> 7) * srows + a) * 4 + ((col >> 5) & 3)] = (uint8_t)e8;
        }
    }
    SA_KV8_T(tend)
    SA_KV8_ACC(7, tl1, tend)
}

template <int D, int MAXG>
static inline size_t decode_attn_kv8_lds() {
    return (size_t)2 * 2 * 256 * D + (size_t)2 * 2 * 256 * 4 + (size_t)16 * D * 2 + (size_t)(MAXG + 2) * D * 4 + 2 * D + 16;
}

}  // namespace sa
