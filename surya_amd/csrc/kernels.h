// Non-GEMM kernels of the recognition path (gfx950, wave64). All are HBM/latency-bound:
// 16-byte vector accesses, wave-shuffle reductions, LDS staging of K/V. Templated on the storage type
// T (bf16_t or float); arithmetic is fp32.
#pragma once
#include "common.h"

#ifndef SA_DBG_VARIANT
#define SA_DBG_VARIANT 0     // tools/microbench only: 1 = no cached-key loop, 2 = no split-K slab loads, 3 = no LDS merge, 4 = no K/V preload
#endif

namespace sa {

// ---------------------------------------------------------------------------------------------------
// Patch tiles fp32 [P, kin] -> T [P, kout] (zero padded), rows gathered by src_row (window order), so the
// reference's `hidden_states[window_index]` gather (encoder/__init__.py:622-627) costs nothing extra.
template <typename T>
__global__ void convert_tiles_kernel(const float* __restrict__ in, T* __restrict__ out, const int* __restrict__ src_row,
                                     int P, int kin, int kout) {
    const int row = blockIdx.x;
    const float* src = in + (long)src_row[row] * kin;
    T* dst = out + (long)row * kout;
    for (int c = threadIdx.x * 4; c < kout; c += blockDim.x * 4) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (c + r < kin) ? src[c + r] : 0.f;
        store4(dst + c, v[0], v[1], v[2], v[3]);
    }
}

// ---------------------------------------------------------------------------------------------------
// RMSNorm, Qwen2RMSNorm semantics (encoder/__init__.py:99-104): y = w * T(x * rsqrt(mean(x^2) + eps)).
// One wave per row; optional row gather (src_row) for "last token only" use.
template <typename T>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const T* __restrict__ x, long ldx, const T* __restrict__ w,
                                                      T* __restrict__ y, long ldy, const int* __restrict__ src_row,
                                                      int rows, int C, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const T* xr = x + (long)(src_row ? src_row[row] : row) * ldx;
    constexpr int V = Ty<T>::V16;
    float ss = 0.f;
    for (int c = lane * V; c < C; c += 64 * V) {
        float v[V];
        unpack16(*reinterpret_cast<const uint4*>(xr + c), v, (T*)nullptr);
#pragma unroll
        for (int i = 0; i < V; ++i) ss += v[i] * v[i];
    }
    ss = wave_sum(ss);
    const float rstd = rsqrtf(ss / (float)C + eps);
    T* yr = y + (long)row * ldy;
    for (int c = lane * V; c < C; c += 64 * V) {
        float v[V], g[V];
        unpack16(*reinterpret_cast<const uint4*>(xr + c), v, (T*)nullptr);
        unpack16(*reinterpret_cast<const uint4*>(w + c), g, (T*)nullptr);
#pragma unroll
        for (int i = 0; i < V; i += 4)
            store4(yr + c + i, g[i] * Ty<T>::rnd(v[i] * rstd), g[i + 1] * Ty<T>::rnd(v[i + 1] * rstd),
                   g[i + 2] * Ty<T>::rnd(v[i + 2] * rstd), g[i + 3] * Ty<T>::rnd(v[i + 3] * rstd));
    }
}

// ---------------------------------------------------------------------------------------------------
// Vision 2-D RoPE of the q and k parts of qkv [P, 3*H] (apply_rotary_pos_emb_vision, encoder/__init__.py:188-199). Per patch
// the angle vector (length D/2) is [pos_h*f | pos_w*f], f = inv_freq[D/4], duplicated to D (:633); rotate_half pairs element
// i with i + D/2; fp32 math. The rotation itself happens in the qkv GEMM's epilogue (EPI_ROPE, gemm.h); this kernel builds its
// (cos, sin) table tab[p * D/2 + i] once per encoder pass (the first version ran a separate in-place kernel per layer:
// 8 x 165 us per recognition step).
static __global__ void rope_vision_table_kernel(const int* __restrict__ pos_hw, const float* __restrict__ inv_freq, float2* __restrict__ tab,
                                         int P, int D) {
    const int half = D / 2, quarter = D / 4;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)P * half) return;
    const int p = (int)(idx / half), i = (int)(idx % half);
    const float ang = (i < quarter) ? (float)pos_hw[2 * p] * inv_freq[i] : (float)pos_hw[2 * p + 1] * inv_freq[i - quarter];
    float sn, cs;
    sincosf(ang, &sn, &cs);
    tab[idx] = make_float2(cs, sn);
}

// ---------------------------------------------------------------------------------------------------
// Segment attention on the vector ALU (first correct version; the score/PV products are ~1 % of the
// encoder FLOPs). One 256-thread workgroup = 64 queries of one (segment, head); 4 lanes share a query, each
// owning D/4 of the head dim. K/V stream through LDS in 64-key chunks; online softmax in fp32.
//   non-causal varlen (vision windows / whole images, encoder/__init__.py:238-261)
//   causal GQA over the slot KV cache (decoder prefill, decoder/__init__.py:101-128)
struct AttnSegs {
    const int* tile_seg;     // [n_tiles] segment of each 64-query tile
    const int* tile_q0;      // [n_tiles] first query (segment-local) of the tile
    const int* seg_len;      // [n_seg]
    const long* q_off;       // [n_seg] element offset of the segment's first query row
    const long* k_off;       // [n_seg] element offset of the segment's first key row (head 0)
    const long* v_off;       // [n_seg]
    const long* o_off;       // [n_seg]
};

template <typename T, int D>
__global__ __launch_bounds__(256) void attn_valu_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                        T* __restrict__ out, AttnSegs sg, long q_row, long q_head, long k_row,
                                                        long k_head, long o_row, long o_head, int group, int causal, float scale) {
    constexpr int DS = D / 4;                 // elements of the head dim per lane
    constexpr int V = Ty<T>::V16;
    constexpr int CPR = D / V;                // 16-byte chunks per K/V row
    constexpr int KC = (D * sizeof(T) >= 512) ? 32 : 64;   // keys per LDS chunk (<= 32 KiB for K+V)
    __shared__ __attribute__((aligned(16))) T ks[KC * D];
    __shared__ __attribute__((aligned(16))) T vs[KC * D];
    const int tile = blockIdx.x, head = blockIdx.y, kvh = head / group;
    const int seg = sg.tile_seg[tile], q0 = sg.tile_q0[tile], L = sg.seg_len[seg];
    const int tid = threadIdx.x, ql = tid >> 2, sl = tid & 3;
    const int qi = q0 + ql;
    const bool valid = qi < L;
    float qr[DS], acc[DS];
    {
        const T* qp = q + sg.q_off[seg] + (long)min(qi, L - 1) * q_row + (long)head * q_head + sl * DS;
#pragma unroll
        for (int i = 0; i < DS; i += 4) {
            float t4[4];
            load4(qp + i, t4);
#pragma unroll
            for (int r = 0; r < 4; ++r) qr[i + r] = t4[r] * scale;
        }
    }
#pragma unroll
    for (int i = 0; i < DS; ++i) acc[i] = 0.f;
    float mrun = -INFINITY, lrun = 0.f;
    const int kend = causal ? min(L, q0 + 64) : L;
    const T* kbase = k + sg.k_off[seg] + (long)kvh * k_head;
    const T* vbase = v + sg.v_off[seg] + (long)kvh * k_head;
    for (int kc = 0; kc < kend; kc += KC) {
        const int nk = min(KC, kend - kc);
        for (int c = tid; c < nk * CPR; c += 256) {
            const int r = c / CPR, cc = c % CPR;
            *reinterpret_cast<uint4*>(ks + r * D + cc * V) = *reinterpret_cast<const uint4*>(kbase + (long)(kc + r) * k_row + cc * V);
            *reinterpret_cast<uint4*>(vs + r * D + cc * V) = *reinterpret_cast<const uint4*>(vbase + (long)(kc + r) * k_row + cc * V);
        }
        __syncthreads();
        for (int jb = 0; jb < nk; jb += 16) {
            float s[16];
            float bm = -INFINITY;
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) {
                const int j = jb + jj;
                float d = 0.f;
                if (j < nk) {
                    const T* kr = ks + j * D + sl * DS;
#pragma unroll
                    for (int i = 0; i < DS; i += 4) {
                        float t4[4];
                        load4(kr + i, t4);
                        d += qr[i] * t4[0] + qr[i + 1] * t4[1] + qr[i + 2] * t4[2] + qr[i + 3] * t4[3];
                    }
                }
                d = quad_sum(d);
                const bool ok = (j < nk) && (!causal || (kc + j) <= qi);
                s[jj] = ok ? d : -INFINITY;
                bm = fmaxf(bm, s[jj]);
            }
            if (bm == -INFINITY) continue;     // uniform within the 4 lanes of a query; shuffles are above
            const float mnew = fmaxf(mrun, bm);
            const float alpha = __expf(mrun - mnew);
            lrun *= alpha;
#pragma unroll
            for (int i = 0; i < DS; ++i) acc[i] *= alpha;
            mrun = mnew;
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) {
                const int j = jb + jj;
                if (j < nk) {                  // wave-uniform bound
                    const float pj = __expf(s[jj] - mnew);   // exp(-inf) = 0 for masked keys
                    lrun += pj;
                    const T* vr = vs + j * D + sl * DS;
#pragma unroll
                    for (int i = 0; i < DS; i += 4) {
                        float t4[4];
                        load4(vr + i, t4);
                        acc[i] += pj * t4[0]; acc[i + 1] += pj * t4[1]; acc[i + 2] += pj * t4[2]; acc[i + 3] += pj * t4[3];
                    }
                }
            }
        }
        __syncthreads();
    }
    if (valid) {
        const float inv = 1.0f / lrun;
        T* op = out + sg.o_off[seg] + (long)qi * o_row + (long)head * o_head + sl * DS;
#pragma unroll
        for (int i = 0; i < DS; i += 4) store4(op + i, acc[i] * inv, acc[i + 1] * inv, acc[i + 2] * inv, acc[i + 3] * inv);
    }
}

// ---------------------------------------------------------------------------------------------------
// Merged image tokens (window order) -> decoder input embeddings:
//   embeds[dst_row[r]] = merged[r] + img_h_embed[hidx[r]] + img_w_embed[widx[r]]
// = inverse window permutation (encoder/__init__.py:669-670) + 2-D learned position embedding
// (common/surya/__init__.py:233-272,193) + masked_scatter into the <IMAGE> positions (:214-225).
// Rounding follows the reference: (h + w) -> T, feature + that -> T.
// dst[dst_row[r]] = src[r] (rows of H elements): image embeddings encoded ahead -> their <IMAGE> positions of the packed prompt.
template <typename T>
__global__ void scatter_rows_kernel(const T* __restrict__ src, const int* __restrict__ dst_row, T* __restrict__ dst, int H) {
    const int r = blockIdx.x;
    const T* s = src + (long)r * H;
    T* d = dst + (long)dst_row[r] * H;
    for (int c = threadIdx.x * Ty<T>::V16; c < H; c += blockDim.x * Ty<T>::V16)
        *reinterpret_cast<uint4*>(d + c) = *reinterpret_cast<const uint4*>(s + c);
}

template <typename T>
__global__ void scatter_image_kernel(const T* __restrict__ merged, const T* __restrict__ hemb, const T* __restrict__ wemb,
                                     const int* __restrict__ dst_row, const int* __restrict__ hidx, const int* __restrict__ widx,
                                     T* __restrict__ embeds, int H) {
    const int r = blockIdx.x;
    const T* src = merged + (long)r * H;
    const T* he = hemb + (long)hidx[r] * H;
    const T* we = wemb + (long)widx[r] * H;
    T* dst = embeds + (long)dst_row[r] * H;
    for (int c = threadIdx.x * 4; c < H; c += blockDim.x * 4) {
        float a[4], b[4], d[4];
        load4(src + c, a); load4(he + c, b); load4(we + c, d);
        store4(dst + c, a[0] + Ty<T>::rnd(b[0] + d[0]), a[1] + Ty<T>::rnd(b[1] + d[1]), a[2] + Ty<T>::rnd(b[2] + d[2]),
               a[3] + Ty<T>::rnd(b[3] + d[3]));
    }
}

// Token embedding gather: x[t] = table[ids[t]] (rows with ids[t] < 0 are left untouched: image positions).
template <typename T>
__global__ void embed_tokens_kernel(const T* __restrict__ table, const int* __restrict__ ids, T* __restrict__ x, int H) {
    const int t = blockIdx.x;
    const int id = ids[t];
    if (id < 0) return;
    const uint4* src = reinterpret_cast<const uint4*>(table + (long)id * H);
    uint4* dst = reinterpret_cast<uint4*>(x + (long)t * H);
    for (int c = threadIdx.x; c < H / Ty<T>::V16; c += blockDim.x) dst[c] = src[c];
}

// ---------------------------------------------------------------------------------------------------
// Decoder RoPE table, built once per model: (cos, sin)(pos * inv_freq[i]) in fp32, then rounded to the storage type
// exactly where the reference rounds them (Qwen2RotaryEmbedding.forward, decoder/__init__.py:346-361).
template <typename T>
__global__ void rope_table_kernel(const float* __restrict__ inv_freq, float2* __restrict__ table, int Tmax, int half) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Tmax * half) return;
    const int pos = idx / half, i = idx % half;
    float sn, cs;
    sincosf((float)pos * inv_freq[i], &sn, &cs);
    table[idx] = make_float2(Ty<T>::rnd(cs), Ty<T>::rnd(sn));
}

// ---------------------------------------------------------------------------------------------------
// Decoder prefill: RoPE on q (in place) and k, append k/v to the slot KV cache.
// qkv [Ttot, (nq + 2 nkv) * D]; cache layout [slot][kv_head][T_max][D]; cos/sin are rounded to T before use
// (decoder/__init__.py:361), rotate_half pairing (:53-84).
template <typename T>
__global__ void rope_kv_append_kernel(T* __restrict__ qkv, const int* __restrict__ tok_slot, const int* __restrict__ tok_pos,
                                      const float2* __restrict__ rope_cs, T* __restrict__ kc, T* __restrict__ vc,
                                      int nq, int nkv, int D, int Tmax) {
    const int t = blockIdx.x;
    const int slot = tok_slot[t], pos = tok_pos[t];
    const int half = D / 2;
    T* row = qkv + (long)t * (nq + 2 * nkv) * D;
    for (int it = threadIdx.x; it < (nq + nkv) * half; it += blockDim.x) {
        const int i = it % half, hh = it / half;           // hh < nq: q head, else k head
        const float2 csn = rope_cs[(long)pos * half + i];
        const float cs = csn.x, sn = csn.y;
        T* v = row + (long)hh * D;
        const float x1 = Ty<T>::ld(v + i), x2 = Ty<T>::ld(v + i + half);
        const float y1 = x1 * cs - x2 * sn, y2 = x2 * cs + x1 * sn;
        if (hh < nq) {
            Ty<T>::st(v + i, y1); Ty<T>::st(v + i + half, y2);
        } else {
            T* dst = kc + (((long)slot * nkv + (hh - nq)) * Tmax + pos) * D;
            Ty<T>::st(dst + i, y1); Ty<T>::st(dst + i + half, y2);
        }
    }
    const T* vsrc = row + (long)(nq + nkv) * D;
    for (int it = threadIdx.x; it < nkv * D; it += blockDim.x) {
        const int hh = it / D, i = it % D;
        vc[(((long)slot * nkv + hh) * Tmax + pos) * D + i] = vsrc[it];
    }
}

// ---------------------------------------------------------------------------------------------------
// Launch-boundary reduce of a split-K projection fused with the residual add and the NEXT RMSNorm:
//   x <- T(x + bias + sum_s part[s])            (what the unsplit GEMM's EPI_RESIDUAL epilogue would have stored)
//   y <- w * T(x * rsqrt(mean(x^2) + eps))      (Qwen2RMSNorm of the updated residual stream, optional)
// One wave per row, row length H (decode: hidden size). Removes the separate rmsnorm launch of every decode layer.
// Launched with blockDim = H / 4 rounded up to whole waves (<= 1024): every thread owns ONE 4-element chunk, so all its
// loads are in flight together and the normalised row is written from registers. (The first version ran 256 threads over
// H = 1280 -- a second serial trip for the first wave -- and re-read its own stores for the norm: 4.9 us per launch, 1504
// launches per recognition step.)
// Round 4: the launch may carry M extra workgroups (blockIdx.x >= M) that do nothing but request the K (or V) rows the NEXT layer's
// decode attention will read for row blockIdx.x - M (KvPrefetch). That kernel is the only one of the decode step that waits on cold HBM:
// 22 MB per layer that were last touched a whole step (1.1 GB of traffic) ago -- 11.2 us per launch against 7.5 us on a warm cache
// (profiles/r04_a_decode_attn_phases.txt). The reduce kernels move ~5 MB and leave HBM idle; with M a multiple of 8, workgroup M + r
// runs on XCD r % 8 like the attention workgroups of row r, so the lines arrive in the L2 that will be asked for them. The extra
// workgroups write nothing and share no barrier with the working ones (a prefetch WAVE inside the row's workgroup would hold the
// row's barrier until its loads return: s_endpgm drains the wave's memory queue first).
struct KvPrefetch {
    const unsigned char* base = nullptr;   // cache of the layer to warm ([slot][kv head][T_max][D]); nullptr = no prefetch wave
    const int* slots = nullptr;            // [rows] slot of each row (the active list)
    const int* lens = nullptr;             // [rows] cached tokens of each row
    long slot_stride = 0, head_stride = 0; // bytes
    int heads = 0, row_bytes = 0, max_rows = 0;
};

template <typename T, int SL = 8>      // SL: slabs requested per thread (>= S; the round-3 kernel always asked for 8, clamped duplicates included)
__global__ __launch_bounds__(1024) void splitk_residual_norm_kernel(const float* __restrict__ part, int S, int M, T* __restrict__ x,
                                                                    const T* __restrict__ bias, const T* __restrict__ w,
                                                                    T* __restrict__ y, int H, float eps,
                                                                    uint8_t* __restrict__ y8 = nullptr, uint8_t* __restrict__ sy = nullptr,
                                                                    int srows = 0, KvPrefetch pf = KvPrefetch()) {
    if ((int)blockIdx.x >= M) {
        if (pf.base) {
            const int r = (int)blockIdx.x - M, t16 = (int)threadIdx.x * 16, step = (int)blockDim.x * 16;
            const int bytes = min(pf.lens[r] + 1, pf.max_rows) * pf.row_bytes;
            const unsigned char* p0 = pf.base + (long)pf.slots[r] * pf.slot_stride + t16;
            // Ordinary loads folded into a value that an empty asm "uses": the compiler knows every destination register, keeps eight
            // requests in flight (unroll) and waits for them before the wave ends. (The first version was an asm load the compiler knew
            // nothing about: it re-used the destination quad for the next address while the load was still in flight and the late
            // write-back turned the address into garbage -- MEMORY_APERTURE_VIOLATION on the box, gpurun r04c. A volatile load is
            // compiled to a system-scope flat_load (sc0 sc1) with a full wait after each: it would bypass the L2 it is meant to fill.)
            const int per_head = bytes > t16 ? (bytes - t16 + step - 1) / step : 0;
            const int total = per_head * pf.heads;
            u32x4 acc = {0u, 0u, 0u, 0u};
            int k = 0;
            const unsigned char* p = p0;
#pragma unroll 8
            for (int i = 0; i < total; ++i) {
                acc ^= *reinterpret_cast<const u32x4*>(p + (long)k * step);
                if (++k == per_head) { k = 0; p += pf.head_stride; }
            }
            asm volatile("" ::"v"(acc));
        }
        return;
    }
    const int row = blockIdx.x, tid = threadIdx.x;       // one workgroup per row: M workgroups keep the chip busy at M = 256
    __shared__ float red[16];
    T* xr = x + (long)row * H;
    const int c = tid * 4;
    const bool on_row = c < H;
    const int cc = on_row ? c : 0;                        // idle lanes of the last wave shadow chunk 0 (no stores)
    float v[4], g[4] = {0.f, 0.f, 0.f, 0.f};
    load4(xr + cc, v);
    if (w) load4(w + cc, g);
    if (bias) {
        float b[4];
        load4(bias + cc, b);
        v[0] += b[0]; v[1] += b[1]; v[2] += b[2]; v[3] += b[3];
    }
    f32x4 p4[SL];                                  // slabs in flight
#pragma unroll
    for (int s = 0; s < SL; ++s)                   // clamped slab index: unconditional loads, masked below
        p4[s] = *reinterpret_cast<const f32x4*>(part + ((long)min(s, S - 1) * M + row) * H + cc);
#pragma unroll
    for (int s = 0; s < SL; ++s) {
        const float on = (s < S) ? 1.f : 0.f;
        v[0] += on * p4[s][0]; v[1] += on * p4[s][1]; v[2] += on * p4[s][2]; v[3] += on * p4[s][3];
    }
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = Ty<T>::rnd(v[i]); ss += on_row ? v[i] * v[i] : 0.f; }
    if (on_row) store4(xr + c, v[0], v[1], v[2], v[3]);
    if (!w) return;
    ss = wave_sum(ss);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    float tot = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) tot += red[i];
    const float rstd = rsqrtf(tot / (float)H + eps);
    const float o[4] = {g[0] * Ty<T>::rnd(v[0] * rstd), g[1] * Ty<T>::rnd(v[1] * rstd), g[2] * Ty<T>::rnd(v[2] * rstd),
                        g[3] * Ty<T>::rnd(v[3] * rstd)};
    if (on_row) store4(y + (long)row * H + c, o[0], o[1], o[2], o[3]);
    if (y8) {   // MXFP8 copy of the normalised row for the fp8 decode GEMMs (gemm_mx.h): 8 lanes = one 32-element block (H % 32 == 0)
        const float q[4] = {Ty<T>::rnd(o[0]), Ty<T>::rnd(o[1]), Ty<T>::rnd(o[2]), Ty<T>::rnd(o[3])};
        int e8;
        const uint32_t pk = mx_quant4_oct(q, e8);
        if (on_row) {
            *reinterpret_cast<uint32_t*>(y8 + (long)row * H + c) = pk;
            if ((tid & 7) == 0) sy[((long)(c >> 7) * srows + row) * 4 + ((c >> 5) & 3)] = (uint8_t)e8;   // K-tile-major scales (gemm_mx.h)
        }
    }
}

// Decode-step embedding fused with the first layer's input RMSNorm: x[a] = table[next_token[slot]], y[a] = norm(x[a]).
template <typename T>
__global__ __launch_bounds__(64) void embed_slots_norm_kernel(const T* __restrict__ table, const int* __restrict__ next_token,
                                                              const int* __restrict__ active_slots, const int* __restrict__ kv_len,
                                                              int Tmax, int* __restrict__ row_len, T* __restrict__ x,
                                                              const T* __restrict__ w, T* __restrict__ y, int H, float eps,
                                                              uint8_t* __restrict__ y8 = nullptr, uint8_t* __restrict__ sy = nullptr,
                                                              int srows = 0) {
    const int a = blockIdx.x, lane = threadIdx.x;
    const int slot = active_slots[a];
    // compact per-row context length for this step's attention kernels; clamped so a slot that keeps stepping after its
    // line finished (the host learns that one call late) overwrites its last cache row instead of its neighbour's first
    if (lane == 0) row_len[a] = min(kv_len[slot], Tmax - 1);
    const T* src = table + (long)next_token[slot] * H;
    T* xr = x + (long)a * H;
    float ss = 0.f;
    for (int c = lane * 4; c < H; c += 256) {
        float v[4];
        load4(src + c, v);
        ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
        store4(xr + c, v[0], v[1], v[2], v[3]);
    }
    ss = wave_sum(ss);
    const float rstd = rsqrtf(ss / (float)H + eps);
    T* yr = y + (long)a * H;
    for (int c = lane * 4; c < H; c += 256) {
        float v[4], g[4];
        load4(src + c, v);
        load4(w + c, g);
        const float o[4] = {g[0] * Ty<T>::rnd(v[0] * rstd), g[1] * Ty<T>::rnd(v[1] * rstd), g[2] * Ty<T>::rnd(v[2] * rstd),
                            g[3] * Ty<T>::rnd(v[3] * rstd)};
        store4(yr + c, o[0], o[1], o[2], o[3]);
        if (y8) {   // MXFP8 copy (H % 256 == 0 on this path: whole 8-lane groups stay inside the row)
            const float q[4] = {Ty<T>::rnd(o[0]), Ty<T>::rnd(o[1]), Ty<T>::rnd(o[2]), Ty<T>::rnd(o[3])};
            int e8;
            const uint32_t pk = mx_quant4_oct(q, e8);
            *reinterpret_cast<uint32_t*>(y8 + (long)a * H + c) = pk;
            if ((lane & 7) == 0) sy[((long)(c >> 7) * srows + a) * 4 + ((c >> 5) & 3)] = (uint8_t)e8;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Greedy head (process_outputs, recognition/__init__.py:294-324) on fp32 logits [rows, V]:
// pred = argmax (first max), score = max softmax prob = 1 / sum(exp(x - max)), done = pred in {eos, pad};
// bbox = trunc(sigmoid(W_b h + b) * bbox_size) (common/surya/__init__.py:329); updates the slot state for
// the next step (next input token = pad if done, kv_len += 1).
// PART = true: `logits` holds float4 partials {max, argmax bits, sum exp(v - max), 0} per (row, column tile) written by the
// lm_head GEMM's EPI_ARGMAX epilogue (ldl = V = tiles per row); the combination below is the same reduction with one
// more level: argmax = first maximum over tiles in column order, sum = sum_t s_t * exp(m_t - max).
template <typename T, bool PART>
__global__ __launch_bounds__(256) void greedy_head_kernel(const float* __restrict__ logits, long ldl, int V,
                                                          const T* __restrict__ hidden, int H, const T* __restrict__ wb,
                                                          const T* __restrict__ bb, const int* __restrict__ row_slot,
                                                          int eos_id, int pad_id, float bbox_size, int* __restrict__ out_token,
                                                          float* __restrict__ out_score, int* __restrict__ out_bbox,
                                                          int* __restrict__ next_token, int* __restrict__ kv_len, int len_inc) {
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* lr = logits + (long)r * ldl * (PART ? 4 : 1);
    __shared__ float smax[4], ssum[4];
    __shared__ int sidx[4];
    float best = -INFINITY;
    int bi = 0x7fffffff;
    if constexpr (PART) {
        for (int t = tid; t < V; t += 256) {
            const float4 v = *reinterpret_cast<const float4*>(lr + 4 * t);
            const int vi = __float_as_int(v.y);
            if (v.x > best || (v.x == best && vi < bi)) { best = v.x; bi = vi; }
        }
    } else {
        for (int c = tid * 4; c < V; c += 1024) {
            const float4 v = *reinterpret_cast<const float4*>(lr + c);
            const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (c + i < V && vv[i] > best) { best = vv[i]; bi = c + i; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) { smax[wave] = best; sidx[wave] = bi; }
    __syncthreads();
    best = smax[0]; bi = sidx[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
        if (smax[w] > best || (smax[w] == best && sidx[w] < bi)) { best = smax[w]; bi = sidx[w]; }
    float se = 0.f;
    if constexpr (PART) {
        for (int t = tid; t < V; t += 256) {
            const float4 v = *reinterpret_cast<const float4*>(lr + 4 * t);
            se += v.z * expf(v.x - best);
        }
    } else {
        for (int c = tid * 4; c < V; c += 1024) {
            const float4 v = *reinterpret_cast<const float4*>(lr + c);
            const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (c + i < V) se += expf(vv[i] - best);
        }
    }
    se = wave_sum(se);
    if (lane == 0) ssum[wave] = se;
    __syncthreads();
    const int slot = row_slot[r];
    if (tid == 0) {
        const float tot = ssum[0] + ssum[1] + ssum[2] + ssum[3];
        const bool done = (bi == eos_id) || (bi == pad_id);
        out_token[slot] = bi;
        out_score[slot] = done ? 0.f : 1.0f / tot;
        next_token[slot] = done ? pad_id : bi;
        kv_len[slot] += len_inc;
    }
    // bbox head: 6 dot products of length H, one wave each (waves 0..3 take outputs 0..3, then 4..5)
    const T* hr = hidden + (long)r * H;
    for (int o = wave; o < 6; o += 4) {
        float d = 0.f;
        for (int c = lane; c < H; c += 64) d += Ty<T>::ld(hr + c) * Ty<T>::ld(wb + (long)o * H + c);
        d = wave_sum(d);
        if (lane == 0) {
            const float lin = Ty<T>::rnd(d + Ty<T>::ld(bb + o));
            const float sg = Ty<T>::rnd(1.0f / (1.0f + expf(-lin)));
            out_bbox[slot * 6 + o] = (int)(sg * bbox_size);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Round 4: the decode step's two small per-row kernels, rebuilt around "every load in flight before the first wait".
//
// embed_norm_row: x = table row, y = w * T(x * rsqrt(mean(x^2) + eps)) for ONE row by a whole workgroup of NT threads, each owning
// up to two 4-element chunks. Used by the standalone embedding kernel (first step of a decode call) AND by the greedy head's fused
// tail (inner steps), so both produce the same bits: the sum of squares is reduced in the same tree in both places.
template <typename T, int NT>
__device__ __forceinline__ void embed_norm_row(const T* __restrict__ src, T* __restrict__ xr, const T* __restrict__ w, T* __restrict__ yr,
                                               int H, float eps, float* red, uint8_t* __restrict__ y8, uint8_t* __restrict__ sy, int srows,
                                               int row) {
    const int tid = threadIdx.x;
    float v[2][4], g[2][4];
    bool ok[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int c = (u * NT + tid) * 4;
        ok[u] = c < H;
        const int cc = ok[u] ? c : 0;
        load4(src + cc, v[u]);
        load4(w + cc, g[u]);
    }
    float ss = 0.f;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        if (ok[u]) store4(xr + (u * NT + tid) * 4, v[u][0], v[u][1], v[u][2], v[u][3]);
        ss += ok[u] ? (v[u][0] * v[u][0] + v[u][1] * v[u][1]) + (v[u][2] * v[u][2] + v[u][3] * v[u][3]) : 0.f;
    }
    ss = wave_sum(ss);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) tot += red[i];
    const float rstd = rsqrtf(tot / (float)H + eps);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int c = (u * NT + tid) * 4;
        const float o[4] = {g[u][0] * Ty<T>::rnd(v[u][0] * rstd), g[u][1] * Ty<T>::rnd(v[u][1] * rstd), g[u][2] * Ty<T>::rnd(v[u][2] * rstd),
                            g[u][3] * Ty<T>::rnd(v[u][3] * rstd)};
        if (ok[u]) store4(yr + c, o[0], o[1], o[2], o[3]);
        if (y8) {   // MXFP8 copy (H % 32 == 0: an aligned group of 8 lanes is inside the row or outside it)
            const float q[4] = {Ty<T>::rnd(o[0]), Ty<T>::rnd(o[1]), Ty<T>::rnd(o[2]), Ty<T>::rnd(o[3])};
            int e8;
            const uint32_t pk = mx_quant4_oct(q, e8);
            if (ok[u]) {
                *reinterpret_cast<uint32_t*>(y8 + (long)row * H + c) = pk;
                if ((tid & 7) == 0) sy[((long)(c >> 7) * srows + row) * 4 + ((c >> 5) & 3)] = (uint8_t)e8;
            }
        }
    }
}

constexpr int SA_HEAD_THREADS = 384;      // 6 waves: one per bbox output; 2 x 384 x 4 >= H for the fused embedding

// First step of a decode call: x[a] = table[next_token[slot]], y[a] = norm(x[a]), row_len[a] = context length of the step.
template <typename T>
__global__ __launch_bounds__(SA_HEAD_THREADS) void embed_slots_norm2_kernel(const T* __restrict__ table, const int* __restrict__ next_token,
                                                                            const int* __restrict__ active_slots, const int* __restrict__ kv_len,
                                                                            int Tmax, int* __restrict__ row_len, T* __restrict__ x,
                                                                            const T* __restrict__ w, T* __restrict__ y, int H, float eps,
                                                                            uint8_t* __restrict__ y8, uint8_t* __restrict__ sy, int srows) {
    __shared__ float red[SA_HEAD_THREADS / 64];
    const int a = blockIdx.x;
    const int slot = active_slots[a];
    if (threadIdx.x == 0) row_len[a] = min(kv_len[slot], Tmax - 1);
    embed_norm_row<T, SA_HEAD_THREADS>(table + (long)next_token[slot] * H, x + (long)a * H, w, y + (long)a * H, H, eps, red, y8, sy, srows, a);
}

// Greedy head on the lm_head GEMM's per-tile partials (greedy_head_kernel<T, true> above states the reduction). Differences:
//   * a row's partials (<= 4 per thread) and the operands of the six bbox dot products (one wave each, 16-byte loads) are requested
//     before anything is waited for, and the partials stay in registers for the second pass (the first version walked the partial
//     row twice and the hidden row 2 bytes at a time in two serial rounds: 19.5 us per launch for ~25 KB of operands);
//   * with `table` set, the workgroup continues with the NEXT step's embedding + first RMSNorm of the token it just chose (the
//     embed launch of every inner decode step disappears; row r of x / y is this row's slot because row_slot == the active list).
template <typename T>
__global__ __launch_bounds__(SA_HEAD_THREADS) void greedy_head2_kernel(const float4* __restrict__ part, int tiles_n, const T* __restrict__ hidden,
                                                                       int H, const T* __restrict__ wb, const T* __restrict__ bb,
                                                                       const int* __restrict__ row_slot, int eos_id, int pad_id, float bbox_size,
                                                                       int* __restrict__ out_token, float* __restrict__ out_score,
                                                                       int* __restrict__ out_bbox, int* __restrict__ next_token,
                                                                       int* __restrict__ kv_len, int len_inc,
                                                                       const T* __restrict__ table, const T* __restrict__ wnorm, T* __restrict__ x,
                                                                       T* __restrict__ y, int* __restrict__ row_len, int Tmax, float eps,
                                                                       uint8_t* __restrict__ y8, uint8_t* __restrict__ sy, int srows) {
    constexpr int NT = SA_HEAD_THREADS, NW = NT / 64, NP = 4, V = Ty<T>::V16, NB = 32 / V;   // NB x 64 x V = 2048 >= H
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ float smax[NW], ssum[NW], red[NW];
    __shared__ int sidx[NW];
    const float4* lr = part + (long)r * tiles_n;
    float4 pv[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) pv[k] = lr[min(tid + k * NT, tiles_n - 1)];
    // bbox head operands of output `wave`: NB x 16 bytes of the hidden row and of the weight row per lane (H <= 64 * V * NB)
    uint4 hv[NB], wv[NB];
    const T* hr = hidden + (long)r * H;
    const T* wr = wb + (long)wave * H;
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const int c = (u * 64 + lane) * V, cc = c < H ? c : 0;
        hv[u] = *reinterpret_cast<const uint4*>(hr + cc);
        wv[u] = *reinterpret_cast<const uint4*>(wr + cc);
    }
    const int slot = row_slot[r];
    const int klen = kv_len[slot];
    const float bias_o = Ty<T>::ld(bb + wave);

    float best = -INFINITY;
    int bi = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int vi = __float_as_int(pv[k].y);
        if (tid + k * NT < tiles_n && (pv[k].x > best || (pv[k].x == best && vi < bi))) { best = pv[k].x; bi = vi; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) { smax[wave] = best; sidx[wave] = bi; }
    __syncthreads();
    best = smax[0]; bi = sidx[0];
#pragma unroll
    for (int w = 1; w < NW; ++w)
        if (smax[w] > best || (smax[w] == best && sidx[w] < bi)) { best = smax[w]; bi = sidx[w]; }
    float se = 0.f;
#pragma unroll
    for (int k = 0; k < NP; ++k)
        if (tid + k * NT < tiles_n) se += pv[k].z * expf(pv[k].x - best);
    se = wave_sum(se);
    if (lane == 0) ssum[wave] = se;
    // bbox dot product of this wave's output while the sums settle
    float d = 0.f;
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        float hf[V], wf[V];
        unpack16(hv[u], hf, (T*)nullptr);
        unpack16(wv[u], wf, (T*)nullptr);
        float du = 0.f;
#pragma unroll
        for (int i = 0; i < V; ++i) du += hf[i] * wf[i];
        d += ((u * 64 + lane) * V < H) ? du : 0.f;
    }
    d = wave_sum(d);
    __syncthreads();
    const bool done = (bi == eos_id) || (bi == pad_id);
    const int tok_next = done ? pad_id : bi;
    if (tid == 0) {
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) tot += ssum[w];
        out_token[slot] = bi;
        out_score[slot] = done ? 0.f : 1.0f / tot;
        next_token[slot] = tok_next;
        kv_len[slot] = klen + len_inc;
        if (table) row_len[r] = min(klen + len_inc, Tmax - 1);
    }
    if (lane == 0 && wave < 6) {
        const float lin = Ty<T>::rnd(d + bias_o);
        const float sg = Ty<T>::rnd(1.0f / (1.0f + expf(-lin)));
        out_bbox[slot * 6 + wave] = (int)(sg * bbox_size);
    }
    if (table) embed_norm_row<T, NT>(table + (long)tok_next * H, x + (long)r * H, wnorm, y + (long)r * H, H, eps, red, y8, sy, srows, r);
}


}  // namespace sa
