// Kernels of the layout model family that the recognition / detection kernels do not already provide:
// the Donut-Swin encoder's patchify, LayerNorm with window (un)partition, window attention with relative position bias and the
// cyclic-shift mask, patch merging (surya/common/donut/encoder.py), and the ADETR decoder's box embedding, RMSNorm variant,
// single-query cross attention over the cached encoder keys / values and output heads (surya/common/adetr/decoder.py,
// surya/layout/model/decoder.py). GEMMs, the fused decode self-attention (split-K combine + RoPE + KV append + attention) and
// the RoPE table come from gemm.h / decode_attn.h / kernels.h. HBM- or latency-bound byte work; nothing here is reshaped into a
// GEMM to reach MFMA.
#pragma once
#include "common.h"
#include "kernels.h"

namespace sa {
namespace lay {

// ---------------------------------------------------------------------------------------------------------------------
// pixel_values fp32 [B, C, H, W] -> patch rows [B * gh * gw][Kpad] (storage dtype), K index = (c * P + ky) * P + kx: the flattening
// of DonutSwinPatchEmbeddings' Conv2d weight [E, C, P, P] (donut/encoder.py:195-253), so the convolution is a plain NT GEMM.
template <typename T>
__global__ void patchify_kernel(const float* __restrict__ px, T* __restrict__ rows, int B, int C, int H, int W, int P, int Kpad) {
    const int gw = W / P, gh = H / P;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;        // one thread per (patch, k)
    const long total = (long)B * gh * gw * Kpad;
    if (idx >= total) return;
    const int k = (int)(idx % Kpad);
    const long patch = idx / Kpad;
    float v = 0.f;
    if (k < C * P * P) {
        const int c = k / (P * P), ky = (k / P) % P, kx = k % P;
        const int x = (int)(patch % gw), y = (int)((patch / gw) % gh), b = (int)(patch / ((long)gw * gh));
        v = px[(((long)b * C + c) * H + y * P + ky) * W + x * P + kx];
    }
    Ty<T>::st(rows + idx, v);
}

// ---------------------------------------------------------------------------------------------------------------------
// LayerNorm over C channels, one wave per row. Destination row = dst_row[src row % rows_per_image] + image offset when a
// permutation is given (window partition with the cyclic shift folded in, donut/encoder.py:617-636), else the same row.
// fp32 statistics (two-pass: mean, then centred variance -- what F.layer_norm computes), output = (x - mean) * rstd * w + b.
template <typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, const T* __restrict__ w, const T* __restrict__ b,
                                                        T* __restrict__ y, const int* __restrict__ perm, long rows, int rows_per_image, int C,
                                                        float eps) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const T* xr = x + row * C;
    float s = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
        float v[4];
        load4(xr + c, v);
        s += v[0] + v[1] + v[2] + v[3];
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
        float v[4];
        load4(xr + c, v);
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float d = v[i] - mean; q += d * d; }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
    long drow = row;
    if (perm) drow = (row / rows_per_image) * rows_per_image + perm[row % rows_per_image];
    T* yr = y + drow * C;
    for (int c = lane * 4; c < C; c += 256) {
        float v[4], wv[4], bv[4];
        load4(xr + c, v); load4(w + c, wv); load4(b + c, bv);
        store4(yr + c, (v[0] - mean) * rstd * wv[0] + bv[0], (v[1] - mean) * rstd * wv[1] + bv[1], (v[2] - mean) * rstd * wv[2] + bv[2],
               (v[3] - mean) * rstd * wv[3] + bv[3]);
    }
}

// x[row] += tab[row % rows_per_image] (2-D sin-cos table at a stage's entry, learned position embeddings at the encoder's exit).
template <typename T>
__global__ void add_rows_kernel(T* __restrict__ x, const T* __restrict__ tab, long rows, int rows_per_image, int C) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;        // one thread per 4 channels
    const int cv = C / 4;
    if (idx >= rows * cv) return;
    const long row = idx / cv;
    const int c = (int)(idx % cv) * 4;
    float a[4], t[4];
    load4(x + row * C + c, a);
    load4(tab + (long)(row % rows_per_image) * C + c, t);
    store4(x + row * C + c, a[0] + t[0], a[1] + t[1], a[2] + t[2], a[3] + t[3]);
}

// x[row] += a[perm(row)]: window reverse + reverse cyclic shift + residual add (donut/encoder.py:654-672).
template <typename T>
__global__ void gather_add_kernel(T* __restrict__ x, const T* __restrict__ a, const int* __restrict__ perm, long rows, int rows_per_image,
                                  int C) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int cv = C / 4;
    if (idx >= rows * cv) return;
    const long row = idx / cv;
    const int c = (int)(idx % cv) * 4;
    const long src = (row / rows_per_image) * rows_per_image + perm[row % rows_per_image];
    float xv[4], av[4];
    load4(x + row * C + c, xv);
    load4(a + src * C + c, av);
    store4(x + row * C + c, xv[0] + av[0], xv[1] + av[1], xv[2] + av[2], xv[3] + av[3]);
}

// ---------------------------------------------------------------------------------------------------------------------
// Window attention (DonutSwinSelfAttention, donut/encoder.py:387-444): one workgroup per (window, head); 64 tokens, head dim 32.
//   scores = q k^T / sqrt(32) + relative_position_bias[head] (+ -100 between tokens of different cyclic-shift regions)
// qkv rows are in window order: [q (nh * 32) | k (nkv * 32) | v (nkv * 32)]; query head h reads kv head h % nkv (the reference
// tiles the kv heads with .repeat, :379-385). fp32 math throughout (the reference's SDPA accumulates in fp32 as well).
// Thread t owns query i = t / 4 and keys 16 (t % 4) .. + 15 for the scores, output dims 8 (t % 4) .. + 7 for P V.
template <typename T>
__global__ __launch_bounds__(256) void swin_window_attn_kernel(const T* __restrict__ qkv, const float* __restrict__ bias, T* __restrict__ out,
                                                               int nh, int nkv, int nwx, int nwy, int shift, int ws) {
    constexpr int N = 64, D = 32;
    __shared__ float qs[N][D + 1], ks[N][D + 1], vs[N][D + 1];
    __shared__ float ps[N][N + 1];
    const long win = blockIdx.x;
    const int head = blockIdx.y, tid = threadIdx.x;
    const int row_w = (nh + 2 * nkv) * D;
    const int kvh = head % nkv;
    const T* base = qkv + win * N * row_w;
    for (int i = tid; i < N * (D / 4); i += 256) {
        const int n = i / (D / 4), c = (i % (D / 4)) * 4;
        float a[4];
        load4(base + (long)n * row_w + head * D + c, a);
        qs[n][c] = a[0]; qs[n][c + 1] = a[1]; qs[n][c + 2] = a[2]; qs[n][c + 3] = a[3];
        load4(base + (long)n * row_w + (nh + kvh) * D + c, a);
        ks[n][c] = a[0]; ks[n][c + 1] = a[1]; ks[n][c + 2] = a[2]; ks[n][c + 3] = a[3];
        load4(base + (long)n * row_w + (nh + nkv + kvh) * D + c, a);
        vs[n][c] = a[0]; vs[n][c + 1] = a[1]; vs[n][c + 2] = a[2]; vs[n][c + 3] = a[3];
    }
    __syncthreads();
    const int i = tid >> 2, jb = (tid & 3) * 16;
    // cyclic-shift regions of this window (get_attn_mask, :560-586): only the last window row / column is cut by the shift
    const int wimg = (int)(win % ((long)nwx * nwy));
    const bool last_y = shift > 0 && (wimg / nwx) == nwy - 1, last_x = shift > 0 && (wimg % nwx) == nwx - 1;
    auto region = [&](int n) { return (last_y && (n / ws) >= ws - shift ? 2 : 0) + (last_x && (n % ws) >= ws - shift ? 1 : 0); };
    const int ri = region(i);
    const float scale = 0.17677669529663687f;                       // 32 ** -0.5
    const float* brow = bias + ((long)head * N + i) * N;
    float s[16], m = -INFINITY;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
        const int j = jb + jj;
        float d = 0.f;
#pragma unroll
        for (int c = 0; c < D; ++c) d += qs[i][c] * ks[j][c];
        d = d * scale + brow[j] + (region(j) != ri ? -100.0f : 0.f);
        s[jj] = d;
        m = fmaxf(m, d);
    }
    m = fmaxf(m, dpp_mov<0xB1>(m));                                  // max over the row's 4 threads (one quad)
    m = fmaxf(m, dpp_mov<0x4E>(m));
    float l = 0.f;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) { s[jj] = expf(s[jj] - m); l += s[jj]; }
    l = quad_sum(l);
    const float inv = 1.0f / l;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) ps[i][jb + jj] = s[jj] * inv;
    __syncthreads();
    const int d0 = (tid & 3) * 8;
    float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < N; ++j) {
        const float p = ps[i][j];
#pragma unroll
        for (int c = 0; c < 8; ++c) o[c] += p * vs[j][d0 + c];
    }
    T* op = out + (win * N + i) * (long)(nh * D) + head * D + d0;
    store4(op, o[0], o[1], o[2], o[3]);
    store4(op + 4, o[4], o[5], o[6], o[7]);
}

// ---------------------------------------------------------------------------------------------------------------------
// Patch merging (DonutSwinPatchMerging, donut/encoder.py:289-319): the four neighbours (0,0), (1,0), (0,1), (1,1) of a 2x2 block
// concatenated to 4C channels, LayerNorm(4C, eps 1e-5); the reduction Linear(4C -> 2C) is a GEMM on the rows written here.
// One wave per output row. H and W are even (checked by the host).
template <typename T>
__global__ __launch_bounds__(256) void merge_ln_kernel(const T* __restrict__ x, const T* __restrict__ w, const T* __restrict__ b,
                                                       T* __restrict__ y, int B, int H, int W, int C, float eps) {
    const int Ho = H / 2, Wo = W / 2;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= (long)B * Ho * Wo) return;
    const int ox = (int)(row % Wo), oy = (int)((row / Wo) % Ho), bi = (int)(row / ((long)Wo * Ho));
    const int C4 = 4 * C;
    auto src = [&](int c4) {                                       // channel c4 of the concatenated row
        const int part = c4 / C, c = c4 % C;
        const int dy = part & 1, dx = part >> 1;                   // order: [0::2, 0::2], [1::2, 0::2], [0::2, 1::2], [1::2, 1::2]
        return x + (((long)bi * H + 2 * oy + dy) * W + 2 * ox + dx) * C + c;
    };
    float s = 0.f;
    for (int c = lane * 4; c < C4; c += 256) {
        float v[4];
        load4(src(c), v);
        s += v[0] + v[1] + v[2] + v[3];
    }
    const float mean = wave_sum(s) / (float)C4;
    float q = 0.f;
    for (int c = lane * 4; c < C4; c += 256) {
        float v[4];
        load4(src(c), v);
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float d = v[i] - mean; q += d * d; }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C4 + eps);
    T* yr = y + row * C4;
    for (int c = lane * 4; c < C4; c += 256) {
        float v[4], wv[4], bv[4];
        load4(src(c), v); load4(w + c, wv); load4(b + c, bv);
        store4(yr + c, (v[0] - mean) * rstd * wv[0] + bv[0], (v[1] - mean) * rstd * wv[1] + bv[1], (v[2] - mean) * rstd * wv[2] + bv[2],
               (v[3] - mean) * rstd * wv[3] + bv[3]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// BboxEmbedding (surya/layout/model/decoder.py:14-60): 15 table rows per token, summed in the reference's order and rounded to the
// storage dtype after every addition (each `+` of the reference is a tensor op in the model dtype).
//   tables: [w, h, cx, cy, xskew, yskew, x1, y1, x2, y2, x3, y3, x4, y4] each [vocab][Hd], then label [label_count][Hd]
template <typename T>
__global__ __launch_bounds__(256) void box_embed_kernel(const int* __restrict__ boxes, const T* const* __restrict__ tabs, T* __restrict__ x,
                                                        int Hd, int bbox_size, int vocab, int label_count) {
    const int b = blockIdx.x;
    const int* bx = boxes + b * 7;
    auto clampv = [&](int v) { return min(max(v, 0), vocab - 1); };
    const int cx = clampv(bx[0]), cy = clampv(bx[1]), w = clampv(bx[2]), h = clampv(bx[3]), xs = clampv(bx[4]), ys = clampv(bx[5]);
    const int label = min(max(bx[6], 0), label_count - 1);
    const int xa = (int)((float)(xs - bbox_size / 2) / 2.0f), ya = (int)((float)(ys - bbox_size / 2) / 2.0f);   // float division, truncation (:39-40)
    auto cl = [&](int v) { return min(max(v, 0), bbox_size); };
    const int x1 = cl(cx - w / 2 - xa), y1 = cl(cy - h / 2 - ya), x2 = cl(cx + w / 2 - xa), y2 = cl(cy + h / 2 + ya);
    const int x3 = cl(cx + w / 2 + xa), y3 = cl(cy + h / 2 + ya), x4 = cl(cx - w / 2 + xa), y4 = cl(cy - h / 2 - ya);
    for (int c = threadIdx.x; c < Hd; c += 256) {
        auto E = [&](int t, int idx) { return Ty<T>::ld(tabs[t] + (long)idx * Hd + c); };
        auto R = [](float v) { return Ty<T>::rnd(v); };
        const float size_e = R(R(R(E(0, w) + E(1, h)) + E(2, cx)) + E(3, cy));
        const float skew_e = R(E(4, xs) + E(5, ys));
        float corner = R(E(6, x1) + E(7, y1));
        corner = R(corner + E(8, x2)); corner = R(corner + E(9, y2)); corner = R(corner + E(10, x3)); corner = R(corner + E(11, y3));
        corner = R(corner + E(12, x4)); corner = R(corner + E(13, y4));
        const float e = R(R(R(E(14, label) + size_e) + skew_e) + corner);
        Ty<T>::st(x + (long)b * Hd + c, e);
    }
}

// SuryaADETRDecoderRMSNorm (adetr/decoder.py:23-47): variance CLAMPED at eps (not added), scale (1 + weight), clamp to the storage
// dtype's finite range, NaN -> 0. One wave per row.
template <typename T>
__global__ __launch_bounds__(256) void adetr_rmsnorm_kernel(const T* __restrict__ x, const T* __restrict__ w, T* __restrict__ y, int rows, int C,
                                                            float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const T* xr = x + (long)row * C;
    float q = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
        float v[4];
        load4(xr + c, v);
        q += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    const float rstd = rsqrtf(fmaxf(wave_sum(q) / (float)C, eps));
    const float lim = sizeof(T) == 2 ? 3.3895313892515355e38f : 3.4028234663852886e38f;      // finfo(bf16 / fp32).max
    for (int c = lane * 4; c < C; c += 256) {
        float v[4], wv[4], o[4];
        load4(xr + c, v); load4(w + c, wv);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float t = v[i] * rstd * (1.0f + wv[i]);
            t = fminf(fmaxf(t, -lim), lim);
            o[i] = (t != t) ? 0.f : t;
        }
        store4(y + (long)row * C + c, o[0], o[1], o[2], o[3]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Cross attention of ONE query token per image over the cached encoder keys / values (SuryaADETRDecoderSdpaCrossAttention,
// adetr/decoder.py:151-190; no mask, no rotary embedding). One workgroup per (image, kv head): its G query heads share the K / V
// rows. kv rows: [B][Lk][2 * nkv * D] = (k heads | v heads) as the fused k|v projection writes them.
template <typename T, int D>
__global__ __launch_bounds__(256) void cross_attn_decode_kernel(const T* __restrict__ q, const T* __restrict__ kv, T* __restrict__ out, int nq,
                                                                int nkv, int Lk, float scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* sc = reinterpret_cast<float*>(smem_raw);                 // [G][Lk] scores -> probabilities
    float* qsh = sc + (nq / nkv) * Lk;                              // [G][D]
    __shared__ float red[2][8][4];                                   // per-wave partials: [max | sum][head g][wave]
    const int b = blockIdx.x, kvh = blockIdx.y, tid = threadIdx.x, G = nq / nkv;
    const int row_w = 2 * nkv * D;
    const T* kb = kv + (long)b * Lk * row_w + kvh * D;
    const T* vb = kb + nkv * D;
    for (int i = tid; i < G * D; i += 256) qsh[i] = Ty<T>::ld(q + (long)b * nq * D + (long)(kvh * G) * D + i);
    __syncthreads();
    // scores: thread -> key j (strided), all G heads from one K row read
    for (int j = tid; j < Lk; j += 256) {
        float kr[D];
#pragma unroll
        for (int c = 0; c < D; c += 4) load4(kb + (long)j * row_w + c, *reinterpret_cast<float(*)[4]>(&kr[c]));
        for (int g = 0; g < G; ++g) {
            float d = 0.f;
#pragma unroll
            for (int c = 0; c < D; ++c) d += qsh[g * D + c] * kr[c];
            sc[g * Lk + j] = d * scale;
        }
    }
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63;
    for (int g = 0; g < G; ++g) {                                   // softmax per head: block max, block sum
        float m = -INFINITY;
        for (int j = tid; j < Lk; j += 256) m = fmaxf(m, sc[g * Lk + j]);
        m = wave_max(m);
        if (lane == 0) red[0][g][wave] = m;
    }
    __syncthreads();
    for (int g = 0; g < G; ++g) {
        const float m = fmaxf(fmaxf(red[0][g][0], red[0][g][1]), fmaxf(red[0][g][2], red[0][g][3]));
        float l = 0.f;
        for (int j = tid; j < Lk; j += 256) { const float e = expf(sc[g * Lk + j] - m); sc[g * Lk + j] = e; l += e; }
        l = wave_sum(l);
        if (lane == 0) red[1][g][wave] = l;
    }
    __syncthreads();
    // P V: thread -> (head g, 4 output dims), V rows are read coalesced by the threads of a head
    for (int it = tid; it < G * (D / 4); it += 256) {
        const int g = it / (D / 4), c = (it % (D / 4)) * 4;
        const float inv = 1.0f / (red[1][g][0] + red[1][g][1] + red[1][g][2] + red[1][g][3]);
        float o[4] = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < Lk; ++j) {
            float vv[4];
            load4(vb + (long)j * row_w + c, vv);
            const float p = sc[g * Lk + j];
            o[0] += p * vv[0]; o[1] += p * vv[1]; o[2] += p * vv[2]; o[3] += p * vv[3];
        }
        store4(out + (long)b * nq * D + (long)(kvh * G + g) * D + c, o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Output heads of SuryaLayoutDecoder.forward (layout/model/decoder.py:119-131): final ADETR RMSNorm -> LayerNorm -> class logits
// (label_count rows, no bias) and sigmoid(bbox_head). One workgroup per image; every intermediate is rounded to the storage dtype
// where the reference materialises a tensor. class_logits fp32 [B][label_count], bbox fp32 [B][6].
template <typename T>
__global__ __launch_bounds__(256) void layout_heads_kernel(const T* __restrict__ x, const T* __restrict__ fnorm_w, const T* __restrict__ ln_w,
                                                           const T* __restrict__ ln_b, const T* __restrict__ lm_w, const T* __restrict__ bb_w,
                                                           const T* __restrict__ bb_b, float* __restrict__ cls, float* __restrict__ box, int Hd,
                                                           int label_count, float rms_eps, float ln_eps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* h = reinterpret_cast<float*>(smem_raw);                  // [Hd]
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const T* xr = x + (long)b * Hd;
    auto block_sum = [&](float v) {
        v = wave_sum(v);
        __syncthreads();
        if (lane == 0) red[wave] = v;
        __syncthreads();
        return red[0] + red[1] + red[2] + red[3];
    };
    float q = 0.f;
    for (int c = tid; c < Hd; c += 256) { const float v = Ty<T>::ld(xr + c); q += v * v; }
    const float rstd = rsqrtf(fmaxf(block_sum(q) / (float)Hd, rms_eps));
    const float lim = sizeof(T) == 2 ? 3.3895313892515355e38f : 3.4028234663852886e38f;
    float s = 0.f;
    for (int c = tid; c < Hd; c += 256) {
        float t = Ty<T>::ld(xr + c) * rstd * (1.0f + Ty<T>::ld(fnorm_w + c));
        t = fminf(fmaxf(t, -lim), lim);
        t = Ty<T>::rnd((t != t) ? 0.f : t);
        h[c] = t;
        s += t;
    }
    const float mean = block_sum(s) / (float)Hd;
    float vq = 0.f;
    for (int c = tid; c < Hd; c += 256) { const float d = h[c] - mean; vq += d * d; }
    const float lrstd = rsqrtf(block_sum(vq) / (float)Hd + ln_eps);
    __syncthreads();
    for (int c = tid; c < Hd; c += 256) h[c] = Ty<T>::rnd((h[c] - mean) * lrstd * Ty<T>::ld(ln_w + c) + Ty<T>::ld(ln_b + c));
    __syncthreads();
    for (int o = wave; o < label_count + 6; o += 4) {               // one wave per output neuron
        const T* wr = o < label_count ? lm_w + (long)o * Hd : bb_w + (long)(o - label_count) * Hd;
        float d = 0.f;
        for (int c = lane; c < Hd; c += 64) d += h[c] * Ty<T>::ld(wr + c);
        d = wave_sum(d);
        if (lane == 0) {
            if (o < label_count) cls[(long)b * label_count + o] = Ty<T>::rnd(d);
            else {
                const float z = Ty<T>::rnd(d + Ty<T>::ld(bb_b + o - label_count));
                box[(long)b * 6 + o - label_count] = Ty<T>::rnd(1.0f / (1.0f + expf(-z)));
            }
        }
    }
}

}  // namespace lay
}  // namespace sa
