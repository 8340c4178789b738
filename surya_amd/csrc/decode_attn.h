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
    stash_tile();
    __syncthreads();
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
        // ---- O = alpha * O + P . V on the VALU
        if (tid < G * (D / 4)) {
            const float alpha = hstat[2 * MAXG + oh];
#pragma unroll
            for (int e = 0; e < 4; ++e) oacc[e] *= alpha;
            for (int j = 0; j < nk; ++j) {
                const float pj = P[oh * KT + j];
                float v4[4];
                load4(reinterpret_cast<const T*>(Vs + j * ROWB) + od, v4);
                oacc[0] += pj * v4[0]; oacc[1] += pj * v4[1]; oacc[2] += pj * v4[2]; oacc[3] += pj * v4[3];
            }
        }
        __syncthreads();
    }
    if (tid < G * (D / 4)) {
        const float inv = 1.0f / hstat[MAXG + oh];
        store4(out + (long)a * nq * D + (long)(kvh * G + oh) * D + od, oacc[0] * inv, oacc[1] * inv, oacc[2] * inv, oacc[3] * inv);
    }
}

template <typename T, int D, int MAXG>
static inline size_t decode_attn_mfma_lds() {
    constexpr int ROWB = D * (int)sizeof(T);
    constexpr int KT = (ROWB >= 512) ? 64 : 128;
    return (size_t)2 * KT * ROWB + 32 * ROWB + (size_t)MAXG * KT * 4 + (size_t)(MAXG + 2) * D * 4 + 3 * MAXG * 4 + 64;
}

}  // namespace sa
