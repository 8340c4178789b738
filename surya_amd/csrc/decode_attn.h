// Decode-step attention, second version: one workgroup per (active slot, kv head), scores on the matrix cores.
//
// The first version (decode_attn_kernel, kernels.h) gave every key to 16 lanes and ran the online softmax per key in
// every lane: 13 of its 24 us were cross-lane reductions and redundant exp/rescale VALU work (tools/microbench). Here:
//   1. the slot's cached K and V rows are fetched as whole 16-byte chunks into LDS tiles (issued before the prologue so
//      their HBM latency overlaps it);
//   2. prologue: this row's q/k/v = sum of the split-K slabs + bias, RoPE from the table, KV append (as before);
//   3. S = K_tile . q^T on MFMA: wave w owns 32 keys, first operand = K fragment (rows = keys), second = q fragment
//      (columns = the group's heads, zero-padded to 32) -> each lane (head = lane & 31 < G) gets 16 keys' scores;
//   4. softmax over the tile per head (one wave-reduction per head), running (max, sum) across tiles for long contexts;
//   5. O += P . V on the VALU with a thread per (head, 4 output dims): P broadcast from LDS, V rows contiguous.
// Replaces cache concat + 4-D mask + SDPA of the reference (decoder/__init__.py:193-234, recognition/cache.py:57-105).
#pragma once
#include "gemm.h"

namespace sa {

// Phase timestamps for tools/microbench/decode_attn2_bench.hip (-DSA_DA_TIMING): workgroup (0, 0), thread 0 records the
// 100 MHz wall clock at phase boundaries. Compiled out of the product library.
#ifdef SA_DA_TIMING
__device__ unsigned long long sa_da_stamps[16];
#define SA_DA_STAMP(i) if (blockIdx.x == 64 && blockIdx.y == 0 && threadIdx.x == 0) sa_da_stamps[i] = wall_clock64();
#else
#define SA_DA_STAMP(i)
#endif

template <typename T, int D, int MAXG>
__global__ __launch_bounds__(256) void decode_attn_mfma_kernel(const float* __restrict__ qkv_part, int S, const T* __restrict__ qkv_bias,
                                                               T* __restrict__ out, T* __restrict__ kc, T* __restrict__ vc,
                                                               const int* __restrict__ active_slots, const int* __restrict__ row_len,
                                                               const float2* __restrict__ rope_cs, int nq, int nkv, int Tmax,
                                                               float scale) {
    constexpr int V16 = Ty<T>::V16;                          // elements per 16-byte chunk
    constexpr int CPRk = D / V16;                            // 16-byte chunks per K/V row
    constexpr int ROWB = D * (int)sizeof(T);                 // bytes per K/V row
    constexpr int KT = (ROWB >= 512) ? 64 : 128;             // keys per LDS tile (32 KiB for K, 32 KiB for V)
    constexpr int XM = CPRk >= 16 ? 15 : CPRk - 1;           // XOR mask of the K-tile chunk swizzle
    constexpr int NCH = KT * CPRk / 256;                     // chunks per thread per tile (per K and per V)
    static_assert(KT * CPRk % 256 == 0 && NCH >= 1 && D % 32 == 0 && MAXG <= 32, "geometry");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Ks = smem;                                // [KT][ROWB], chunk c of row r at c ^ (r & XM)
    unsigned char* Vs = Ks + KT * ROWB;                      // [KT][ROWB], linear
    T* qT = reinterpret_cast<T*>(Vs + KT * ROWB);            // [32][D] q heads (rows >= G are zero), storage dtype
    float* P = reinterpret_cast<float*>(qT + 32 * D);        // [MAXG][KT] scores -> probabilities
    float* xrow = P + MAXG * KT;                             // [(MAXG + 2) * D]
    float* hstat = xrow + (MAXG + 2) * D;                    // [3][MAXG]: running max, running sum, rescale factor

    SA_DA_STAMP(0)
    const int G = nq / nkv;
    const int a = blockIdx.x, kvh = blockIdx.y;
    const int slot = active_slots[a];
    const int len = row_len[a];                              // cached tokens; the new token sits at index len
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int total = len + 1;
    const unsigned char* kb = reinterpret_cast<const unsigned char*>(kc + ((long)slot * nkv + kvh) * Tmax * D);
    const unsigned char* vb = reinterpret_cast<const unsigned char*>(vc + ((long)slot * nkv + kvh) * Tmax * D);

    // ---- fetch of the first tile's cached rows (indices < len), before anything else
    u32x4 kreg[NCH], vreg[NCH];
    auto fetch_tile = [&](int base) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int id = tid + i * 256, r = id / CPRk, c = id % CPRk;
            const int j = min(base + r, max(len - 1, 0));    // clamped: rows >= len are never used
            kreg[i] = *reinterpret_cast<const u32x4*>(kb + (long)j * ROWB + c * 16);
            vreg[i] = *reinterpret_cast<const u32x4*>(vb + (long)j * ROWB + c * 16);
        }
    };
    auto stash_tile = [&]() {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int id = tid + i * 256, r = id / CPRk, c = id % CPRk;
            *reinterpret_cast<u32x4*>(Ks + r * ROWB + ((c ^ (r & XM)) << 4)) = kreg[i];
            *reinterpret_cast<u32x4*>(Vs + r * ROWB + (c << 4)) = vreg[i];
        }
    };
    fetch_tile(0);
    SA_DA_STAMP(1)

    // ---- prologue: q/k/v of this row (split-K slabs + bias), all loads issued before the first wait
    const int qkv_dim = (nq + 2 * nkv) * D;
    const int Mrows = gridDim.x;
    const int half = D / 2;
    const float2 csn = rope_cs[(long)len * half + (tid % half)];
    constexpr int NI = ((MAXG + 2) * D + 255) / 256;
    const int n_items = (G + 2) * D;
    float p8[NI][8], bias_v[NI];
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const int it = min(tid + k * 256, n_items - 1);
        const int hh = it / D, i = it % D;
        const int col = (hh < G ? (kvh * G + hh) * D : (hh == G ? (nq + kvh) * D : (nq + nkv + kvh) * D)) + i;
#pragma unroll
        for (int sidx = 0; sidx < 8; ++sidx) p8[k][sidx] = qkv_part[((long)min(sidx, S - 1) * Mrows + a) * qkv_dim + col];
        bias_v[k] = Ty<T>::ld(qkv_bias + col);
    }
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        float val = bias_v[k];
#pragma unroll
        for (int sidx = 0; sidx < 8; ++sidx) val += (sidx < S) ? p8[k][sidx] : 0.f;
        if (tid + k * 256 < n_items) xrow[tid + k * 256] = Ty<T>::rnd(val);
    }
    for (int i = tid; i < (32 - G) * D; i += 256) qT[G * D + i] = T(0);              // zero the padding heads of the q operand
    if (tid < 3 * MAXG) hstat[tid] = (tid < MAXG) ? -INFINITY : 0.f;
    SA_DA_STAMP(2)
    stash_tile();
    __syncthreads();
    SA_DA_STAMP(3)
    // RoPE (decoder/__init__.py:60-84, cos/sin rounded to the storage dtype): q -> qT (scaled), k -> cache + tile, v -> cache + tile
    T* knew_dst = kc + (((long)slot * nkv + kvh) * Tmax + len) * D;
    T* vnew_dst = vc + (((long)slot * nkv + kvh) * Tmax + len) * D;
    const int new_tile = len / KT, new_row = len % KT;       // where the new token's row lives among the tiles
    for (int it = tid; it < (G + 1) * half; it += 256) {
        const int i = it % half, hh = it / half;
        const float cs = csn.x, sn = csn.y;
        const float x1 = xrow[hh * D + i], x2 = xrow[hh * D + i + half];
        const float y1 = Ty<T>::rnd(x1 * cs - x2 * sn), y2 = Ty<T>::rnd(x2 * cs + x1 * sn);
        if (hh < G) {
            Ty<T>::st(qT + hh * D + i, y1 * scale); Ty<T>::st(qT + hh * D + i + half, y2 * scale);
        } else {
            Ty<T>::st(knew_dst + i, y1); Ty<T>::st(knew_dst + i + half, y2);
            if (new_tile == 0) {
                auto kput = [&](int e, float v) {
                    const int c = e / V16, w = e % V16;
                    Ty<T>::st(reinterpret_cast<T*>(Ks + new_row * ROWB + ((c ^ (new_row & XM)) << 4)) + w, v);
                };
                kput(i, y1); kput(i + half, y2);
            }
        }
    }
    for (int i = tid; i < D; i += 256) {
        const float val = xrow[(G + 1) * D + i];
        Ty<T>::st(vnew_dst + i, val);
        if (new_tile == 0) Ty<T>::st(reinterpret_cast<T*>(Vs + new_row * ROWB) + i, val);
    }
    __syncthreads();
    SA_DA_STAMP(4)

    // ---- per-thread output accumulator: thread (h, d4) for tid < G * D / 4
    const int oh = tid / (D / 4), od = (tid % (D / 4)) * 4;
    float oacc[4] = {0.f, 0.f, 0.f, 0.f};
    const int n_tiles = (total + KT - 1) / KT;
    for (int t = 0; t < n_tiles; ++t) {
        const int base = t * KT, nk = min(KT, total - base);
        if (t > 0) {
            // longer contexts: next tile (previous tile's LDS is free after the barrier that ended the last PV phase)
            fetch_tile(base);
            stash_tile();
            __syncthreads();
            if (new_tile == t) {                             // the new token's row falls into this tile: copy it from the cache rows
                for (int i = tid; i < D; i += 256) {         // (written above by this workgroup; re-read through LDS-visible xrow instead)
                    const float vv = xrow[(G + 1) * D + i];
                    Ty<T>::st(reinterpret_cast<T*>(Vs + new_row * ROWB) + i, vv);
                }
                for (int it = tid; it < half; it += 256) {
                    const float cs = csn.x, sn = csn.y;      // it % half == tid % half
                    const float x1 = xrow[G * D + it], x2 = xrow[G * D + it + half];
                    const float y1 = Ty<T>::rnd(x1 * cs - x2 * sn), y2 = Ty<T>::rnd(x2 * cs + x1 * sn);
                    const int c1 = it / V16, w1 = it % V16, c2 = (it + half) / V16, w2 = (it + half) % V16;
                    Ty<T>::st(reinterpret_cast<T*>(Ks + new_row * ROWB + ((c1 ^ (new_row & XM)) << 4)) + w1, y1);
                    Ty<T>::st(reinterpret_cast<T*>(Ks + new_row * ROWB + ((c2 ^ (new_row & XM)) << 4)) + w2, y2);
                }
                __syncthreads();
            }
        }
        // ---- scores on MFMA: wave w -> keys [32w, 32w+32) of the tile (KT = 128: all four waves; KT = 64: waves 0,1)
        if (wave * 32 < KT) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const int krow = wave * 32 + (lane & 31);
#pragma unroll
            for (int ks = 0; ks < CPRk / 2; ++ks) {
                const int c = ks * 2 + (lane >> 5);
                const u32x4 kf = *reinterpret_cast<const u32x4*>(Ks + krow * ROWB + ((c ^ (krow & XM)) << 4));
                const u32x4 qf = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(qT) + (lane & 31) * ROWB + (c << 4));
                Mfma<T>::run(acc, kf, qf);
            }
            const int h = lane & 31;
            if (h < G) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = wave * 32 + g * 8 + (lane >> 5) * 4 + r;
                        P[h * KT + key] = (key < nk) ? acc[4 * g + r] : -INFINITY;
                    }
            }
        }
        __syncthreads();
        SA_DA_STAMP(5)
        // ---- softmax over the tile, one wave per head (waves take heads w, w + 4, ...)
        for (int h = wave; h < G; h += 4) {
            float mx = -INFINITY;
            for (int j = lane; j < KT; j += 64) mx = fmaxf(mx, P[h * KT + j]);
            mx = wave_max(mx);
            const float m_old = hstat[h], m_new = fmaxf(m_old, mx);
            float sum = 0.f;
            for (int j = lane; j < KT; j += 64) {
                const float pj = __expf(P[h * KT + j] - m_new);     // exp(-inf) = 0 beyond nk
                P[h * KT + j] = pj;
                sum += pj;
            }
            sum = wave_sum(sum);
            if (lane == 0) {
                const float alpha = (m_old == -INFINITY) ? 0.f : __expf(m_old - m_new);
                hstat[h] = m_new;
                hstat[MAXG + h] = hstat[MAXG + h] * alpha + sum;
                hstat[2 * MAXG + h] = alpha;
            }
        }
        __syncthreads();
        SA_DA_STAMP(6)
        // ---- O = alpha * O + P . V on the VALU
        if (tid < G * (D / 4)) {
            const float alpha = hstat[2 * MAXG + oh];
#pragma unroll
            for (int e = 0; e < 4; ++e) oacc[e] *= alpha;
            // 8 keys per trip so their LDS reads are in flight together (one key per trip cost a full LDS round trip per
            // key, ~100 cycles x 100 keys); keys in [nk, nk8) have P = exp(-inf) = 0 and V rows that are clamped duplicates
            // of a real row, so they add exact zeros.
            const int nk8 = (nk + 7) & ~7;
            for (int j0 = 0; j0 < nk8; j0 += 8) {
                float pj[8], v4[8][4];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    pj[u] = P[oh * KT + j0 + u];
                    load4(reinterpret_cast<const T*>(Vs + (j0 + u) * ROWB) + od, v4[u]);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    oacc[0] += pj[u] * v4[u][0]; oacc[1] += pj[u] * v4[u][1]; oacc[2] += pj[u] * v4[u][2]; oacc[3] += pj[u] * v4[u][3];
                }
            }
        }
        __syncthreads();
        SA_DA_STAMP(7)
    }
    if (tid < G * (D / 4)) {
        const float inv = 1.0f / hstat[MAXG + oh];
        store4(out + (long)a * nq * D + (long)(kvh * G + oh) * D + od, oacc[0] * inv, oacc[1] * inv, oacc[2] * inv, oacc[3] * inv);
    }
    SA_DA_STAMP(8)
}

// Third version (bf16): the phase timers of the second version (tools/microbench/decode_attn2_bench.hip -DSA_DA_TIMING)
// showed 6.2 of its 12.8 us in three barrier-separated compute phases (scores 1.8, tile softmax 1.8, VALU P.V 2.6) and
// 4.2 us waiting for a K/V fetch that always moved 128 rows per (slot, head). Here:
//   * K/V rows go global -> LDS directly (global_load_lds, swizzle on the source address, no staging registers, no stash
//     phase), and only rows up to the context length rounded to a 16-key MFMA step are requested;
//   * wave w owns keys [32w, 32w + 32) of each 128-key tile and runs its OWN flash-attention state for them:
//     S^T = K q^T on MFMA (a lane owns one head), per-lane softmax statistics, the exp() registers are the P fragment,
//     O^T += V^T P^T on MFMA with ds_read_b64_tr_b16 V^T fragments (same recipe as attn_mfma.h) -- no barrier and no LDS
//     traffic between scores and P.V;
//   * one split-KV combine of the four waves' (max, sum, O) through LDS at the end.
template <int D, int MAXG>
__global__ __launch_bounds__(256) void decode_attn_flash_kernel(const float* __restrict__ qkv_part, int S, const bf16_t* __restrict__ qkv_bias,
                                                                bf16_t* __restrict__ out, bf16_t* __restrict__ kc, bf16_t* __restrict__ vc,
                                                                const int* __restrict__ active_slots, const int* __restrict__ row_len,
                                                                const float2* __restrict__ rope_cs, int nq, int nkv, int Tmax,
                                                                float scale, uint8_t* __restrict__ out8 = nullptr,
                                                                uint8_t* __restrict__ sout = nullptr, int srows = 0) {
    typedef bf16_t T;
    constexpr int CPR = D / 8;                               // 16-byte chunks per K/V row
    constexpr int ROWB = D * 2;                              // bytes per K/V row
    constexpr int KT = 128;                                  // keys per LDS tile: 32 per wave
    constexpr int RPI = 1024 / ROWB;                         // rows moved by one global_load_lds (64 lanes x 16 B)
    constexpr int XM = CPR >= 16 ? 15 : CPR - 1;             // XOR mask of the K-tile chunk swizzle
    constexpr int NKK = D / 16, NDB = (D + 31) / 32;
    constexpr int CW = D + 4;                                // floats per (wave, head) record of the final combine: O[D], max, sum
    static_assert(D % 32 == 0 && D <= 128 && MAXG <= 32 && 1024 % ROWB == 0, "geometry");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Ks = smem;                                // [KT][ROWB], chunk c of row r at c ^ (r & XM)
    unsigned char* Vs = Ks + KT * ROWB;                      // [KT][ROWB], linear
    T* qT = reinterpret_cast<T*>(Vs + KT * ROWB);            // [32][D] q heads (rows >= G are zero)
    float* xrow = reinterpret_cast<float*>(qT + 32 * D);     // [(MAXG + 2) * D]
    float* comb = reinterpret_cast<float*>(qT);              // [4][MAXG][CW], aliases qT + xrow after the key loop

    SA_DA_STAMP(0)
    const int G = nq / nkv;
    const int a = blockIdx.x, kvh = blockIdx.y;
    const int slot = active_slots[a];
    const int len = row_len[a];                              // cached tokens; the new token sits at index len
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int total = len + 1;
    const unsigned char* kb = reinterpret_cast<const unsigned char*>(kc + ((long)slot * nkv + kvh) * Tmax * D);
    const unsigned char* vb = reinterpret_cast<const unsigned char*>(vc + ((long)slot * nkv + kvh) * Tmax * D);
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

    // rows of tile `base` that a 16-key MFMA step can touch get finite data (cached rows, clamped duplicates past len);
    // groups of RPI rows are dealt to the four waves
    auto issue_tile = [&](int base) {
        const int rows = min(KT, (total - base + 15) & ~15);
        const int ngroups = (rows + RPI - 1) / RPI;
        for (int g = wave; g < ngroups; g += 4) {
            const int r = g * RPI + lane / CPR, pc = lane % CPR;
            const long j = min(base + r, max(len - 1, 0));
            __builtin_amdgcn_global_load_lds((gptr_t)(kb + j * ROWB + ((pc ^ (r & XM)) << 4)), (lptr_t)(Ks + g * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(vb + j * ROWB + (pc << 4)), (lptr_t)(Vs + g * 1024), 16, 0, 0);
        }
    };
    issue_tile(0);
    SA_DA_STAMP(1)

    // ---- prologue: q/k/v of this row (split-K slabs + bias), all loads issued before the first wait
    const int qkv_dim = (nq + 2 * nkv) * D;
    const int Mrows = gridDim.x;
    const int half = D / 2;
    const float2 csn = rope_cs[(long)len * half + (tid % half)];
    constexpr int NI = ((MAXG + 2) * D + 255) / 256;
    const int n_items = (G + 2) * D;
    float p8[NI][8], bias_v[NI];
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const int it = min(tid + k * 256, n_items - 1);
        const int hh = it / D, i = it % D;
        const int col = (hh < G ? (kvh * G + hh) * D : (hh == G ? (nq + kvh) * D : (nq + nkv + kvh) * D)) + i;
#pragma unroll
        for (int sidx = 0; sidx < 8; ++sidx) p8[k][sidx] = qkv_part[((long)min(sidx, S - 1) * Mrows + a) * qkv_dim + col];
        bias_v[k] = Ty<T>::ld(qkv_bias + col);
    }
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        float val = bias_v[k];
#pragma unroll
        for (int sidx = 0; sidx < 8; ++sidx) val += (sidx < S) ? p8[k][sidx] : 0.f;
        if (tid + k * 256 < n_items) xrow[tid + k * 256] = Ty<T>::rnd(val);
    }
    // zero the padding heads of the q operand: whole rows, 16 bytes per store (2-byte stores here cost ~14 conflicting LDS
    // writes per thread; r03 counters: two thirds of this kernel's LDS cycles were bank conflicts)
    for (int i = tid; i < (32 - G) * CPR; i += 256)
        *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned char*>(qT) + G * ROWB + i * 16) = u32x4{0u, 0u, 0u, 0u};
    SA_DA_STAMP(2)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the LDS-DMA rows of tile 0 (not tracked by the compiler)
    __syncthreads();
    SA_DA_STAMP(3)
    // RoPE (decoder/__init__.py:60-84, cos/sin rounded to the storage dtype): q -> qT (scaled), k -> cache + tile, v -> cache + tile
    T* knew_dst = kc + (((long)slot * nkv + kvh) * Tmax + len) * D;
    T* vnew_dst = vc + (((long)slot * nkv + kvh) * Tmax + len) * D;
    const int new_tile = len / KT, new_row = len % KT;       // where the new token's row lives among the tiles
    // q rows are stored with the K tile's chunk swizzle (chunk c of head row r at c ^ (r & XM)): the q fragment reads below take
    // the same 16-byte chunk of 32 different rows at a 256-byte pitch -- 16-way bank conflicts on the plain layout.
    // Two adjacent dims per thread -> 4-byte stores.
    auto kput2 = [&](int e, float v0, float v1) {             // elements e, e + 1 of the new K row (e even)
        const int c = e / 8, w = e % 8;
        store2(reinterpret_cast<T*>(Ks + new_row * ROWB + ((c ^ (new_row & XM)) << 4)) + w, v0, v1);
    };
    auto qput2 = [&](int hh, int e, float v0, float v1) {
        const int c = e / 8, w = e % 8;
        store2(reinterpret_cast<T*>(reinterpret_cast<unsigned char*>(qT) + hh * ROWB + ((c ^ (hh & XM)) << 4)) + w, v0, v1);
    };
    const float4 cs2 = *reinterpret_cast<const float4*>(rope_cs + (long)len * half + (tid % (half / 2)) * 2);   // (cos, sin) of dims i, i + 1
    for (int it = tid; it < (G + 1) * (half / 2); it += 256) {
        const int i = (it % (half / 2)) * 2, hh = it / (half / 2);          // it % (half / 2) == tid % (half / 2): 256 % (half / 2) == 0
        const float xa1 = xrow[hh * D + i], xa2 = xrow[hh * D + i + half], xb1 = xrow[hh * D + i + 1], xb2 = xrow[hh * D + i + 1 + half];
        const float ya1 = Ty<T>::rnd(xa1 * cs2.x - xa2 * cs2.y), ya2 = Ty<T>::rnd(xa2 * cs2.x + xa1 * cs2.y);
        const float yb1 = Ty<T>::rnd(xb1 * cs2.z - xb2 * cs2.w), yb2 = Ty<T>::rnd(xb2 * cs2.z + xb1 * cs2.w);
        if (hh < G) {
            qput2(hh, i, ya1 * scale, yb1 * scale); qput2(hh, i + half, ya2 * scale, yb2 * scale);
        } else {
            store2(knew_dst + i, ya1, yb1); store2(knew_dst + i + half, ya2, yb2);
            if (new_tile == 0) { kput2(i, ya1, yb1); kput2(i + half, ya2, yb2); }
        }
    }
    for (int i = tid; i < D; i += 256) {
        const float val = xrow[(G + 1) * D + i];
        Ty<T>::st(vnew_dst + i, val);
        if (new_tile == 0) Ty<T>::st(reinterpret_cast<T*>(Vs + new_row * ROWB) + i, val);
    }
    __syncthreads();
    SA_DA_STAMP(4)

    // ---- per-wave flash attention over keys [32 * wave, 32 * wave + 32) of every tile
    const int hl = lane & 31, h = lane >> 5;                 // this lane's head (column of S^T) and K-half
    u32x4 qf[NKK];
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk)
        qf[kk] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(qT) + hl * ROWB + (((kk * 2 + h) ^ (hl & XM)) << 4));
    f32x16 oacc[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[db][r] = 0.f;
    float mrun = -INFINITY, lrun = 0.f;
    const int tr_off = (((lane & 15) >> 2) + h * 4) * D + ((lane >> 4) & 1) * 16 + (lane & 3) * 4;   // elements, see attn_mfma.h
    const int n_tiles = (total + KT - 1) / KT;
    for (int t = 0; t < n_tiles; ++t) {
        const int base = t * KT, nk = min(KT, total - base);
        if (t > 0) {
            __syncthreads();                                 // every wave is done with the previous tile
            issue_tile(base);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (new_tile == t) {                             // the new token's row falls into this tile
                for (int i = tid; i < D; i += 256) Ty<T>::st(reinterpret_cast<T*>(Vs + new_row * ROWB) + i, xrow[(G + 1) * D + i]);
                for (int it = tid; it < half / 2; it += 256) {
                    const int i = it * 2;                    // it == tid here: cs2 holds dims i, i + 1
                    const float xa1 = xrow[G * D + i], xa2 = xrow[G * D + i + half], xb1 = xrow[G * D + i + 1], xb2 = xrow[G * D + i + 1 + half];
                    kput2(i, Ty<T>::rnd(xa1 * cs2.x - xa2 * cs2.y), Ty<T>::rnd(xb1 * cs2.z - xb2 * cs2.w));
                    kput2(i + half, Ty<T>::rnd(xa2 * cs2.x + xa1 * cs2.y), Ty<T>::rnd(xb2 * cs2.z + xb1 * cs2.w));
                }
                __syncthreads();
            }
        }
        const int k0 = wave * 32;
        if (k0 < nk) {                                       // wave-uniform
            f32x16 sacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
            const int krow = k0 + hl;
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                const u32x4 kf = *reinterpret_cast<const u32x4*>(Ks + krow * ROWB + (((kk * 2 + h) ^ (krow & XM)) << 4));
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, qf[kk]), sacc, 0, 0, 0);
            }
            float bm = -INFINITY;                            // register 4g + r = key k0 + 8g + 4h + r
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float sv = (k0 + g * 8 + h * 4 + r < nk) ? sacc[4 * g + r] : -INFINITY;
                    sacc[4 * g + r] = sv;
                    bm = fmaxf(bm, sv);
                }
            bm = fmaxf(bm, __shfl_xor(bm, 32, 64));          // finite: key k0 is visible
            const float mnew = fmaxf(mrun, bm);
            const float alpha = __expf(mrun - mnew);         // exp(-inf) = 0 on the first tile
            mrun = mnew;
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __expf(sacc[r] - mnew);
                sacc[r] = pv;
                psum += pv;
            }
            lrun = lrun * alpha + psum;
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                if (k0 + st * 16 < nk) {                     // wave-uniform; rows of a started 16-key step hold finite data
                    u32x4 pf;
                    pf[0] = pack2(sacc[8 * st + 0], sacc[8 * st + 1]);
                    pf[1] = pack2(sacc[8 * st + 2], sacc[8 * st + 3]);
                    pf[2] = pack2(sacc[8 * st + 4], sacc[8 * st + 5]);
                    pf[3] = pack2(sacc[8 * st + 6], sacc[8 * st + 7]);
#pragma unroll
                    for (int db = 0; db < NDB; ++db) {
                        const bf16_t* vp = reinterpret_cast<const bf16_t*>(Vs) + (k0 + st * 16) * D + db * 32 + tr_off;
                        typedef short s16x4_t __attribute__((ext_vector_type(4)));
                        typedef short s16x8_t __attribute__((ext_vector_type(8)));
                        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(vp));
                        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(vp + 8 * D));
                        const s16x8_t vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                        oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf), __builtin_bit_cast(bf16x8, pf),
                                                                           oacc[db], 0, 0, 0);
                    }
                }
            }
        }
    }
    SA_DA_STAMP(5)
    // ---- split-KV combine of the four waves
    const float ltot = lrun + __shfl_xor(lrun, 32, 64);
    __syncthreads();                                         // qT / xrow are dead: comb aliases them
    if (hl < G) {
        float* rec = comb + (wave * MAXG + hl) * CW;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(rec + db * 32 + g * 8 + h * 4) =
                    make_float4(oacc[db][4 * g], oacc[db][4 * g + 1], oacc[db][4 * g + 2], oacc[db][4 * g + 3]);
        if (h == 0) { rec[D] = mrun; rec[D + 1] = ltot; }
    }
    __syncthreads();
    SA_DA_STAMP(6)
    if (tid < G * (D / 4)) {
        const int oh = tid / (D / 4), od = (tid % (D / 4)) * 4;
        float mx = -INFINITY;
#pragma unroll
        for (int w = 0; w < 4; ++w) mx = fmaxf(mx, comb[(w * MAXG + oh) * CW + D]);
        float num[4] = {0.f, 0.f, 0.f, 0.f}, den = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
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
    SA_DA_STAMP(7)
    SA_DA_STAMP(8)
}

// Fourth version (bf16), round 4: the third version's phase timers at the bench's context (60-110 cached keys = ONE tile) showed a kernel
// that is all prologue: 32 scalar 4-byte slab loads per thread, an fp32 round trip of the row through LDS (xrow) with its own barrier
// before the rotary step, and a barrier in front of the combine records because they aliased the q operand. Here:
//   * thread t < (G + 2) * D / 8 owns 4 consecutive dims i..i+3 of one head AND their rotate_half partners i + D/2..: two 16-byte loads
//     per slab, the reduce, the rounding, the rotation and the q / K-new / V-new LDS writes are thread-local -- no xrow, no barrier
//     between reduce and RoPE;
//   * the K / V rows appended to the cache are stored to global memory AFTER the barrier that publishes the LDS tile (nothing in this
//     launch reads them back, so no wave waits for their acknowledgement);
//   * each wave drops its (max, sum, O) record into its OWN 32 key rows of the K tile (only that wave ever read them), so one barrier
//     separates the per-wave flash pass from the final combine.
// Arithmetic (slab order, rounding points, MFMA order, combine) is the third version's; outputs agree to the bit unless hipcc contracts
// the rotation differently (pinned with explicit fma / mul below).
// DB (round 4, long contexts -- the texify horizon of 200..970 cached keys): two K/V tile buffers. The single-buffered loop above
// exposes every tile's fetch (barrier, issue, vmcnt(0), barrier, compute: ~5.5 us per 128 keys at 590 cached keys, of which ~1.4 are
// compute); with DB tile t + 1 is requested right after the barrier that frees its buffer and lands while tile t is computed: one
// barrier per tile, fetch and compute overlapped. 2 x 64 KB + the q operand = 136 KB of LDS for D = 128, i.e. one workgroup per CU:
// the host picks DB only when some active slot's context exceeds one tile (rec_model.hip, ctx bound kept on the host), where the
// launch has at most one workgroup per CU anyway at the task's batch (128 slots x 2 kv heads). Same arithmetic, same order: outputs
// are bit-identical to the single-buffered kernel.
template <int D, int MAXG, bool DB = false>
__global__ __launch_bounds__(256) void decode_attn_flash2_kernel(const float* __restrict__ qkv_part, int S, const bf16_t* __restrict__ qkv_bias,
                                                                 bf16_t* __restrict__ out, bf16_t* __restrict__ kc, bf16_t* __restrict__ vc,
                                                                 const int* __restrict__ active_slots, const int* __restrict__ row_len,
                                                                 const float2* __restrict__ rope_cs, int nq, int nkv, int Tmax,
                                                                 float scale, uint8_t* __restrict__ out8 = nullptr,
                                                                 uint8_t* __restrict__ sout = nullptr, int srows = 0) {
    typedef bf16_t T;
    constexpr int CPR = D / 8;                               // 16-byte chunks per K/V row
    constexpr int ROWB = D * 2;                              // bytes per K/V row
    constexpr int KT = 128;                                  // keys per LDS tile: 32 per wave
    constexpr int RPI = 1024 / ROWB;                         // rows moved by one global_load_lds (64 lanes x 16 B)
    constexpr int XM = CPR >= 16 ? 15 : CPR - 1;             // XOR mask of the K-tile chunk swizzle
    constexpr int NKK = D / 16, NDB = (D + 31) / 32;
    constexpr int CW = D + 4;                                // floats per head record of the final combine: O[D], max, sum
    constexpr int IPH = D / 8;                               // prologue items (threads) per head: 4 dims + their 4 partners each
    static_assert(D % 32 == 0 && D <= 128 && MAXG <= 32 && 1024 % ROWB == 0, "geometry");
    static_assert((MAXG + 2) * IPH <= 256 && MAXG * CW * 4 <= 32 * ROWB, "one prologue pass; a wave's records fit its own K rows");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NBUF = DB ? 2 : 1;
    constexpr int BUFB = 2 * KT * ROWB;                      // one tile buffer: K rows, then V rows
    unsigned char* Ks = smem;                                // [KT][ROWB], chunk c of row r at c ^ (r & XM)      (buffer 0)
    unsigned char* Vs = Ks + KT * ROWB;                      // [KT][ROWB], linear                                 (buffer 0)
    unsigned char* qT = smem + NBUF * BUFB;                  // [32][ROWB] q heads (rows >= G are zero), chunk-swizzled like K

    SA_DA_STAMP(0)
    const int G = nq / nkv;
    const int a = blockIdx.x, kvh = blockIdx.y;
    const int slot = active_slots[a];
    const int len = row_len[a];                              // cached tokens; the new token sits at index len
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int total = len + 1;
    const unsigned char* kb = reinterpret_cast<const unsigned char*>(kc + ((long)slot * nkv + kvh) * Tmax * D);
    const unsigned char* vb = reinterpret_cast<const unsigned char*>(vc + ((long)slot * nkv + kvh) * Tmax * D);
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

    // Rows of tile `base` that a 16-key MFMA step can touch get finite data (cached rows, clamped duplicates past len) -- EXCEPT row
    // `len` itself: the lanes that would fill it are masked off (LDS-DMA honours EXEC), so the new token's K / V rows, which this
    // workgroup writes with ordinary LDS stores, never race with a late-landing duplicate and need no barrier of their own.
    auto issue_tile = [&](int base, int buf) {
        const int rows = min(KT, (total - base + 15) & ~15);
        const int ngroups = (rows + RPI - 1) / RPI;
        unsigned char* Kd = Ks + buf * BUFB;
        unsigned char* Vd = Vs + buf * BUFB;
        for (int g = wave; g < ngroups; g += 4) {
            const int r = g * RPI + lane / CPR, pc = lane % CPR;
            const long j = min(base + r, max(len - 1, 0));
            if (base + r != len) {
                __builtin_amdgcn_global_load_lds((gptr_t)(kb + j * ROWB + ((pc ^ (r & XM)) << 4)), (lptr_t)(Kd + g * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gptr_t)(vb + j * ROWB + (pc << 4)), (lptr_t)(Vd + g * 1024), 16, 0, 0);
            }
        }
    };
    issue_tile(0, 0);
    SA_DA_STAMP(1)

    // ---- prologue: this row's q / k / v = bias + sum of the split-K slabs, rounded; RoPE; all thread-local
    const int qkv_dim = (nq + 2 * nkv) * D;
    const int Mrows = gridDim.x;
    constexpr int half = D / 2;
    const int n_items = (G + 2) * IPH;
    const int item = min(tid, n_items - 1);                   // idle threads shadow the last item (no stores)
    const bool on = tid < n_items;
    const int hh = item / IPH, i0 = (item % IPH) * 4;         // head (q heads, then k, then v) and first dim (< D/2)
    const int col = (hh < G ? (kvh * G + hh) * D : (hh == G ? (nq + kvh) * D : (nq + nkv + kvh) * D)) + i0;
    const float* prow = qkv_part + (long)a * qkv_dim + col;
    const long sstride = (long)Mrows * qkv_dim;
    f32x4 plo[8], phi[8];
    // slab loads: unconditional, slab index clamped (duplicates hit L1); only as many as the launch can need
    const int SL = S <= 2 ? 2 : (S <= 4 ? 4 : 8);
    if (SL == 2) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            plo[s] = *reinterpret_cast<const f32x4*>(prow + min(s, S - 1) * sstride);
            phi[s] = *reinterpret_cast<const f32x4*>(prow + min(s, S - 1) * sstride + half);
        }
#pragma unroll
        for (int s = 2; s < 8; ++s) { plo[s] = f32x4{0.f, 0.f, 0.f, 0.f}; phi[s] = plo[s]; }
    } else if (SL == 4) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            plo[s] = *reinterpret_cast<const f32x4*>(prow + min(s, S - 1) * sstride);
            phi[s] = *reinterpret_cast<const f32x4*>(prow + min(s, S - 1) * sstride + half);
        }
#pragma unroll
        for (int s = 4; s < 8; ++s) { plo[s] = f32x4{0.f, 0.f, 0.f, 0.f}; phi[s] = plo[s]; }
    } else {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            plo[s] = *reinterpret_cast<const f32x4*>(prow + min(s, S - 1) * sstride);
            phi[s] = *reinterpret_cast<const f32x4*>(prow + min(s, S - 1) * sstride + half);
        }
    }
    float blo[4], bhi[4];
    load4(qkv_bias + col, blo);
    load4(qkv_bias + col + half, bhi);
    const float4 csA = *reinterpret_cast<const float4*>(rope_cs + (long)len * half + i0);        // (cos, sin) of dims i0, i0 + 1
    const float4 csB = *reinterpret_cast<const float4*>(rope_cs + (long)len * half + i0 + 2);    // dims i0 + 2, i0 + 3
    // zero the padding heads of the q operand: whole rows, 16 bytes per store
    for (int i = tid; i < (32 - G) * CPR; i += 256)
        *reinterpret_cast<u32x4*>(qT + G * ROWB + i * 16) = u32x4{0u, 0u, 0u, 0u};
    SA_DA_STAMP(2)
    float xlo[4], xhi[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float vl = blo[e], vh = bhi[e];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const bool use = s < S;
            vl += use ? plo[s][e] : 0.f;
            vh += use ? phi[s][e] : 0.f;
        }
        xlo[e] = Ty<T>::rnd(vl); xhi[e] = Ty<T>::rnd(vh);
    }
    const float csn[4] = {csA.x, csA.z, csB.x, csB.z}, snn[4] = {csA.y, csA.w, csB.y, csB.w};
    float ylo[4], yhi[4];                                     // what goes to the q operand / the cache rows (v: unrotated)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        // decoder/__init__.py:60-84 with cos / sin rounded to the storage dtype (the table); one rounded product, one fused multiply-add
        const float r1 = Ty<T>::rnd(__fmaf_rn(xlo[e], csn[e], -__fmul_rn(xhi[e], snn[e])));
        const float r2 = Ty<T>::rnd(__fmaf_rn(xhi[e], csn[e], __fmul_rn(xlo[e], snn[e])));
        ylo[e] = hh <= G ? r1 : xlo[e];
        yhi[e] = hh <= G ? r2 : xhi[e];
    }
    const int new_tile = len / KT, new_row = len % KT;       // where the new token's row lives among the tiles
    auto lds_put4 = [&](unsigned char* rowp, int swz, int e0, const float (&v)[4], float mul) {   // 4 consecutive elements from e0 (multiple of 4)
        const int c = e0 / 8, w = e0 % 8;
        store4(reinterpret_cast<T*>(rowp + ((c ^ swz) << 4)) + w, v[0] * mul, v[1] * mul, v[2] * mul, v[3] * mul);
    };
    auto put_new_rows = [&](int buf) {                        // K-new (rotated) and V-new rows of the tile that holds index len
        if (on && hh == G) {
            lds_put4(Ks + buf * BUFB + new_row * ROWB, new_row & XM, i0, ylo, 1.f);
            lds_put4(Ks + buf * BUFB + new_row * ROWB, new_row & XM, i0 + half, yhi, 1.f);
        } else if (on && hh == G + 1) {
            lds_put4(Vs + buf * BUFB + new_row * ROWB, 0, i0, ylo, 1.f);
            lds_put4(Vs + buf * BUFB + new_row * ROWB, 0, i0 + half, yhi, 1.f);
        }
    };
    if (on && hh < G) {
        lds_put4(qT + hh * ROWB, hh & XM, i0, ylo, scale);
        lds_put4(qT + hh * ROWB, hh & XM, i0 + half, yhi, scale);
    }
    if (new_tile == 0) put_new_rows(0);
    SA_DA_STAMP(3)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the LDS-DMA rows of tile 0 (not tracked by the compiler)
    __syncthreads();
    // append to the cache: after the barrier, so no wave's vmcnt(0) above waits for a store acknowledgement
    if (on && hh >= G) {
        T* dst = (hh == G ? kc : vc) + (((long)slot * nkv + kvh) * Tmax + len) * D;
        store4(dst + i0, ylo[0], ylo[1], ylo[2], ylo[3]);
        store4(dst + i0 + half, yhi[0], yhi[1], yhi[2], yhi[3]);
    }
    SA_DA_STAMP(4)

    // ---- per-wave flash attention over keys [32 * wave, 32 * wave + 32) of every tile
    const int hl = lane & 31, h = lane >> 5;                 // this lane's head (column of S^T) and K-half
    u32x4 qf[NKK];
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk)
        qf[kk] = *reinterpret_cast<const u32x4*>(qT + hl * ROWB + (((kk * 2 + h) ^ (hl & XM)) << 4));
    f32x16 oacc[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[db][r] = 0.f;
    float mrun = -INFINITY, lrun = 0.f;
    const int tr_off = (((lane & 15) >> 2) + h * 4) * D + ((lane >> 4) & 1) * 16 + (lane & 3) * 4;   // elements, see attn_mfma.h
    const int n_tiles = (total + KT - 1) / KT;
    if (DB && n_tiles > 1) issue_tile(KT, 1);                // lands while tile 0 is computed (buffer 1 is untouched so far)
    for (int t = 0; t < n_tiles; ++t) {
        const int base = t * KT, nk = min(KT, total - base);
        const int buf = DB ? (t & 1) : 0;
        if (t > 0) {
            if (DB) {
                // tile t was requested one iteration ago into the buffer tile t - 2 has left (every wave passed the previous barrier
                // after computing it); the new token's rows are ordinary LDS stores into rows the DMA skips
                if (new_tile == t) put_new_rows(buf);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's share of tile t (nothing younger is in flight)
                __syncthreads();                             // tile t complete for everyone; everyone is done with tile t - 1
                if (t + 1 < n_tiles) issue_tile(base + KT, buf ^ 1);
            } else {
                __syncthreads();                             // every wave is done with the previous tile
                issue_tile(base, 0);
                if (new_tile == t) put_new_rows(0);          // the new token's rows fall into this tile (the DMA above skips row `len`)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
        }
        const unsigned char* Kt = Ks + buf * BUFB;
        const unsigned char* Vt = Vs + buf * BUFB;
        const int k0 = wave * 32;
        if (k0 < nk) {                                       // wave-uniform
            f32x16 sacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
            const int krow = k0 + hl;
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                const u32x4 kf = *reinterpret_cast<const u32x4*>(Kt + krow * ROWB + (((kk * 2 + h) ^ (krow & XM)) << 4));
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, qf[kk]), sacc, 0, 0, 0);
            }
            float bm = -INFINITY;                            // register 4g + r = key k0 + 8g + 4h + r
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float sv = (k0 + g * 8 + h * 4 + r < nk) ? sacc[4 * g + r] : -INFINITY;
                    sacc[4 * g + r] = sv;
                    bm = fmaxf(bm, sv);
                }
            bm = fmaxf(bm, __shfl_xor(bm, 32, 64));          // finite: key k0 is visible
            const float mnew = fmaxf(mrun, bm);
            const float alpha = __expf(mrun - mnew);         // exp(-inf) = 0 on the first tile
            mrun = mnew;
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __expf(sacc[r] - mnew);
                sacc[r] = pv;
                psum += pv;
            }
            lrun = lrun * alpha + psum;
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                if (k0 + st * 16 < nk) {                     // wave-uniform; rows of a started 16-key step hold finite data
                    u32x4 pf;
                    pf[0] = pack2(sacc[8 * st + 0], sacc[8 * st + 1]);
                    pf[1] = pack2(sacc[8 * st + 2], sacc[8 * st + 3]);
                    pf[2] = pack2(sacc[8 * st + 4], sacc[8 * st + 5]);
                    pf[3] = pack2(sacc[8 * st + 6], sacc[8 * st + 7]);
#pragma unroll
                    for (int db = 0; db < NDB; ++db) {
                        const bf16_t* vp = reinterpret_cast<const bf16_t*>(Vt) + (k0 + st * 16) * D + db * 32 + tr_off;
                        typedef short s16x4_t __attribute__((ext_vector_type(4)));
                        typedef short s16x8_t __attribute__((ext_vector_type(8)));
                        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(vp));
                        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(vp + 8 * D));
                        const s16x8_t vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                        oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf), __builtin_bit_cast(bf16x8, pf),
                                                                           oacc[db], 0, 0, 0);
                    }
                }
            }
        }
    }
    SA_DA_STAMP(5)
    // ---- split-KV combine of the four waves: records live in each wave's own K rows of buffer 0 (rows 32w..32w+31: read by wave w
    // only, in whichever tile used that buffer, and no LDS-DMA is in flight any more), so no barrier is needed before they are written
    const float ltot = lrun + __shfl_xor(lrun, 32, 64);
    if (hl < G) {
        float* rec = reinterpret_cast<float*>(Ks + wave * 32 * ROWB) + hl * CW;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(rec + db * 32 + g * 8 + h * 4) =
                    make_float4(oacc[db][4 * g], oacc[db][4 * g + 1], oacc[db][4 * g + 2], oacc[db][4 * g + 3]);
        if (h == 0) { rec[D] = mrun; rec[D + 1] = ltot; }
    }
    __syncthreads();
    SA_DA_STAMP(6)
    if (tid < G * (D / 4)) {
        const int oh = tid / (D / 4), od = (tid % (D / 4)) * 4;
        const float* r0 = reinterpret_cast<const float*>(Ks) + oh * CW;
        constexpr int WSTR = 32 * ROWB / 4;                  // floats between the records of consecutive waves
        float mx = -INFINITY;
#pragma unroll
        for (int w = 0; w < 4; ++w) mx = fmaxf(mx, r0[w * WSTR + D]);
        float num[4] = {0.f, 0.f, 0.f, 0.f}, den = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float* rec = r0 + w * WSTR;
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
            const int colo = (kvh * G + oh) * D + od;            // K-tile-major scales (gemm_mx.h): [col / 128][srows][4]
            if ((tid & 7) == 0) sout[((long)(colo >> 7) * srows + a) * 4 + ((colo >> 5) & 3)] = (uint8_t)e8;
        }
    }
    SA_DA_STAMP(7)
    SA_DA_STAMP(8)
}

template <int D, int MAXG, bool DB = false>
static inline size_t decode_attn_flash2_lds() { return (size_t)(DB ? 2 : 1) * 2 * 128 * D * 2 + (size_t)32 * D * 2 + 64; }

template <int D, int MAXG>
static inline size_t decode_attn_flash_lds() {
    const size_t q_x = (size_t)32 * D * 2 + (size_t)(MAXG + 2) * D * 4, comb = (size_t)4 * MAXG * (D + 4) * 4;
    return (size_t)2 * 128 * D * 2 + (q_x > comb ? q_x : comb) + 64;
}

template <typename T, int D, int MAXG>
static inline size_t decode_attn_mfma_lds() {
    constexpr int ROWB = D * (int)sizeof(T);
    constexpr int KT = (ROWB >= 512) ? 64 : 128;
    return (size_t)2 * KT * ROWB + 32 * ROWB + (size_t)MAXG * KT * 4 + (size_t)(MAXG + 2) * D * 4 + 3 * MAXG * 4 + 64;
}

}  // namespace sa
