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
                                                        float eps, int rows_per_image_out = 0) {
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
    // (the window-order side has more rows per image than the token side when a stage's grid is padded to whole windows)
    if (perm) drow = (row / rows_per_image) * (rows_per_image_out ? rows_per_image_out : rows_per_image) + perm[row % rows_per_image];
    T* yr = y + drow * C;
    for (int c = lane * 4; c < C; c += 256) {
        float v[4], wv[4], bv[4];
        load4(xr + c, v); load4(w + c, wv); load4(b + c, bv);
        store4(yr + c, (v[0] - mean) * rstd * wv[0] + bv[0], (v[1] - mean) * rstd * wv[1] + bv[1], (v[2] - mean) * rstd * wv[2] + bv[2],
               (v[3] - mean) * rstd * wv[3] + bv[3]);
    }
}

// bf16 rows of 128 ... 1024 channels (the Swin stages): the kernel above gives a whole wave to a row -- at C = 128 half its lanes idle, the
// other half move 8 bytes each and read the row three times (stage 1 of 32 pages = 1.18 M rows: 489 us for 600 MB, 1.2 TB/s). Here a row
// belongs to LPR = min(64, C / 8) lanes that hold it in registers (one 16-byte load per 8 channels), several rows share a wave, the two
// reductions run inside the LPR-lane group, and a workgroup walks 256 / LPR rows. Same two-pass statistics in fp32; the sums associate
// differently (lane-local 8, then a butterfly), i.e. agreement with the kernel above to fp32 rounding, not to the bit -- bf16 mode only.
template <int LPR, int NV>
__global__ __launch_bounds__(256) void layernorm_rows_bf16_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                                  const bf16_t* __restrict__ b, bf16_t* __restrict__ y,
                                                                  const int* __restrict__ perm, long rows, int rows_per_image, float eps,
                                                                  int rows_per_image_out) {
    constexpr int C = LPR * 8 * NV, RPB = 256 / LPR;
    const int sub = threadIdx.x % LPR;
    const long row = (long)blockIdx.x * RPB + threadIdx.x / LPR;
    const bool live = row < rows;                          // dead rows shadow the last one (the shuffles below need every lane)
    const bf16_t* xr = x + (live ? row : rows - 1) * C;
    u32x4 raw[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) raw[v] = *reinterpret_cast<const u32x4*>(xr + (v * LPR + sub) * 8);
    float f[NV][8];
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[v][2 * i] = __uint_as_float(raw[v][i] << 16);
            f[v][2 * i + 1] = __uint_as_float(raw[v][i] & 0xFFFF0000u);
            s += f[v][2 * i] + f[v][2 * i + 1];
        }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float d = f[v][i] - mean; q += d * d; }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rstd = rsqrtf(q / (float)C + eps);
    if (!live) return;
    long drow = row;
    if (perm) drow = (row / rows_per_image) * (rows_per_image_out ? rows_per_image_out : rows_per_image) + perm[row % rows_per_image];
    bf16_t* yr = y + drow * C;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int c0 = (v * LPR + sub) * 8;
        float wv[8], bv[8];
        load4(w + c0, reinterpret_cast<float(&)[4]>(wv[0])); load4(w + c0 + 4, reinterpret_cast<float(&)[4]>(wv[4]));
        load4(b + c0, reinterpret_cast<float(&)[4]>(bv[0])); load4(b + c0 + 4, reinterpret_cast<float(&)[4]>(bv[4]));
        store4(yr + c0, (f[v][0] - mean) * rstd * wv[0] + bv[0], (f[v][1] - mean) * rstd * wv[1] + bv[1],
               (f[v][2] - mean) * rstd * wv[2] + bv[2], (f[v][3] - mean) * rstd * wv[3] + bv[3]);
        store4(yr + c0 + 4, (f[v][4] - mean) * rstd * wv[4] + bv[4], (f[v][5] - mean) * rstd * wv[5] + bv[5],
               (f[v][6] - mean) * rstd * wv[6] + bv[6], (f[v][7] - mean) * rstd * wv[7] + bv[7]);
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

// Window-order rows that no token maps to = the zero padding of a stage grid that is not a multiple of the window (DonutSwinLayer.maybe_pad,
// donut/encoder.py:588-596: F.pad AFTER layernorm_before, so the padded tokens enter the projections as zero vectors -- their q / k / v
// are the biases -- and take part in the attention like any other token; the rows they produce are cut off again, :668).
template <typename T>
__global__ void zero_rows_kernel(T* __restrict__ y, const int* __restrict__ pad_rows, int n_pad, int B, int rows_per_image, int C) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int cv = C / 4;
    if (idx >= (long)B * n_pad * cv) return;
    const long r = idx / cv;
    const int c = (int)(idx % cv) * 4;
    store4(y + ((r / n_pad) * rows_per_image + pad_rows[r % n_pad]) * C + c, 0.f, 0.f, 0.f, 0.f);
}

// x[row] += a[perm(row)]: window reverse + reverse cyclic shift + residual add (donut/encoder.py:654-672).
template <typename T>
__global__ void gather_add_kernel(T* __restrict__ x, const T* __restrict__ a, const int* __restrict__ perm, long rows, int rows_per_image,
                                  int C, int rows_per_image_src = 0) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int cv = C / 4;
    if (idx >= rows * cv) return;
    const long row = idx / cv;
    const int c = (int)(idx % cv) * 4;
    const long src = (row / rows_per_image) * (rows_per_image_src ? rows_per_image_src : rows_per_image) + perm[row % rows_per_image];
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

// bf16 windows on the matrix cores. One workgroup (2 waves x 32 queries) per (window, head); the tile plan of attn_mfma.h with the
// whole window as its single 64-key chunk:
//   S^T = K Q^T   v_mfma_f32_32x32x16_bf16(K fragment, Q fragment): a lane owns ONE query (lane & 31) and 16 of each 32 keys, so
//                 bias / mask / softmax are per-lane work plus one exchange with lane ^ 32;
//   O^T = V^T P^T the lane's exp() values are its P fragment, the V^T fragment comes from two ds_read_b64_tr_b16 per 16-key step.
// The scalar kernel above spent 23.8 ms per stage-1 layer of 32 pages in LDS reads (profiles/r03_f_layout_kernel_stats_before.md); it
// stays as the fp32 reference-mode path.
__global__ __launch_bounds__(128) void swin_window_attn_mfma_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ bias,
                                                                    bf16_t* __restrict__ out, int nh, int nkv, int nwx, int nwy, int shift,
                                                                    int ws) {
    constexpr int N = 64, D = 32, PK = D + 8;
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    __shared__ __attribute__((aligned(16))) bf16_t ks[N * PK];
    __shared__ __attribute__((aligned(16))) bf16_t vs[N * PK];
    const long win = blockIdx.x;
    const int head = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ql = lane & 31, h = lane >> 5, qi = wave * 32 + ql;
    const int row_w = (nh + 2 * nkv) * D, kvh = head % nkv;
    const bf16_t* base = qkv + win * N * row_w;
    u32x4 kreg[2], vreg[2], qf[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int c = tid + it * 128, r = c >> 2, cc = c & 3;
        kreg[it] = *reinterpret_cast<const u32x4*>(base + (long)r * row_w + (nh + kvh) * D + cc * 8);
        vreg[it] = *reinterpret_cast<const u32x4*>(base + (long)r * row_w + (nh + nkv + kvh) * D + cc * 8);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) qf[kk] = *reinterpret_cast<const u32x4*>(base + (long)qi * row_w + head * D + h * 8 + kk * 16);
    // relative position bias of this lane's query row: register 4g + r of key block kb = key kb * 32 + g * 8 + h * 4 + r
    const float* brow = bias + ((long)head * N + qi) * N + h * 4;
    f32x4 b4[2][4];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int g = 0; g < 4; ++g) b4[kb][g] = *reinterpret_cast<const f32x4*>(brow + kb * 32 + g * 8);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int c = tid + it * 128, r = c >> 2, cc = c & 3;
        *reinterpret_cast<u32x4*>(ks + r * PK + cc * 8) = kreg[it];
        *reinterpret_cast<u32x4*>(vs + r * PK + cc * 8) = vreg[it];
    }
    __syncthreads();
    f32x16 sacc[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[kb][r] = 0.f;
        const bf16_t* kp = ks + (kb * 32 + ql) * PK + h * 8;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const u32x4 kf = *reinterpret_cast<const u32x4*>(kp + kk * 16);
            sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, qf[kk]), sacc[kb], 0, 0, 0);
        }
    }
    const int wimg = (int)(win % ((long)nwx * nwy));
    const bool last_y = shift > 0 && (wimg / nwx) == nwy - 1, last_x = shift > 0 && (wimg % nwx) == nwx - 1;
    auto region = [&](int n) { return (last_y && (n / ws) >= ws - shift ? 2 : 0) + (last_x && (n % ws) >= ws - shift ? 1 : 0); };
    const int ri = region(qi);
    const float scale = 0.17677669529663687f;                       // 32 ** -0.5
    float m = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = kb * 32 + g * 8 + h * 4 + r;
                const float sv = sacc[kb][4 * g + r] * scale + b4[kb][g][r] + (region(j) != ri ? -100.0f : 0.f);    // scores stay fp32
                sacc[kb][4 * g + r] = sv;
                m = fmaxf(m, sv);
            }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float l = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pv = exp2f((sacc[kb][r] - m) * 1.44269504088896340736f);
            sacc[kb][r] = pv;
            l += pv;
        }
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    const int tr_off = ((lane & 15) >> 2) * PK + ((lane >> 4) & 1) * 16 + (lane & 3) * 4 + h * 4 * PK;
    f32x16 oacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int kb = t >> 1, o8 = (t & 1) * 8;
        u32x4 pf;                                                    // probabilities rounded to bf16 as softmax(...).to(bf16) does
        pf[0] = pack2(sacc[kb][o8 + 0] * inv, sacc[kb][o8 + 1] * inv);
        pf[1] = pack2(sacc[kb][o8 + 2] * inv, sacc[kb][o8 + 3] * inv);
        pf[2] = pack2(sacc[kb][o8 + 4] * inv, sacc[kb][o8 + 5] * inv);
        pf[3] = pack2(sacc[kb][o8 + 6] * inv, sacc[kb][o8 + 7] * inv);
        const bf16_t* vp = vs + t * 16 * PK + tr_off;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vp));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vp + 8 * PK));
        const s16x8 vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        oacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf), __builtin_bit_cast(bf16x8, pf), oacc, 0, 0, 0);
    }
    bf16_t* op = out + (win * N + qi) * (long)(nh * D) + head * D;
#pragma unroll
    for (int g = 0; g < 4; ++g) store4(op + g * 8 + h * 4, oacc[4 * g], oacc[4 * g + 1], oacc[4 * g + 2], oacc[4 * g + 3]);
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
// `emit(c, e)` receives column c of the row's embedding (already rounded to the storage dtype): the stand-alone kernel stores it, the
// fused heads kernel also keeps it in LDS for the next layer's norm.
template <typename T, typename Emit>
__device__ __forceinline__ void box_embed_row(const int* __restrict__ bx, const T* const* __restrict__ tabs, int Hd, int bbox_size, int vocab,
                                              int label_count, Emit emit) {
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
        emit(c, R(R(R(E(14, label) + size_e) + skew_e) + corner));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void box_embed_kernel(const int* __restrict__ boxes, const T* const* __restrict__ tabs, T* __restrict__ x,
                                                        int Hd, int bbox_size, int vocab, int label_count) {
    const int b = blockIdx.x;
    box_embed_row<T>(boxes + b * 7, tabs, Hd, bbox_size, vocab, label_count, [&](int c, float e) { Ty<T>::st(x + (long)b * Hd + c, e); });
}

// LabelEmbedding of the table-recognition decoder (surya/table_rec/model/decoder.py:12-73): 10-number tokens (cx, cy, w, h, xskew, yskew,
// category, merges, colspan, is_header); columns [0, BE) = the box embedding (w + h + cx + cy) + (xskew + yskew) + (x1 + y1 + x3 + y3) from
// tables of width BE, columns [BE, Hd) = category + merge + colspan from tables of width Hd - BE; is_header is not embedded. Rounded to
// the storage dtype after every addition, in the reference's order. tables: the 14 box tables (x2 / y2 / x4 / y4 exist but are not read,
// :25-30 vs :63), then category, merge, colspan.
template <typename T, typename Emit>
__device__ __forceinline__ void table_embed_row(const int* __restrict__ bx, const T* const* __restrict__ tabs, int Hd, int BE, int bbox_size,
                                                int vocab, int category_count, int merge_count, Emit emit) {
    auto clampv = [&](int v) { return min(max(v, 0), vocab - 1); };          // boxes.clamp(0, vocab_size) (:48); index vocab would be out of range
    const int cx = clampv(bx[0]), cy = clampv(bx[1]), w = clampv(bx[2]), h = clampv(bx[3]), xs = clampv(bx[4]), ys = clampv(bx[5]);
    const int cat = min(clampv(bx[6]), category_count - 1), mer = min(clampv(bx[7]), merge_count - 1), col = clampv(bx[8]);
    const int xa = (int)((float)(xs - bbox_size / 2) / 2.0f), ya = (int)((float)(ys - bbox_size / 2) / 2.0f);
    auto cl = [&](int v) { return min(max(v, 0), bbox_size); };
    const int x1 = cl(cx - w / 2 - xa), y1 = cl(cy - h / 2 - ya), x3 = cl(cx + w / 2 + xa), y3 = cl(cy + h / 2 + ya);
    const int P = Hd - BE;
    auto R = [](float v) { return Ty<T>::rnd(v); };
    for (int c = threadIdx.x; c < Hd; c += 256) {
        float e;
        if (c < BE) {
            auto E = [&](int t, int idx) { return Ty<T>::ld(tabs[t] + (long)idx * BE + c); };
            const float size_e = R(R(R(E(0, w) + E(1, h)) + E(2, cx)) + E(3, cy));
            const float skew_e = R(E(4, xs) + E(5, ys));
            const float corner = R(R(R(E(6, x1) + E(7, y1)) + E(10, x3)) + E(11, y3));
            e = R(R(size_e + skew_e) + corner);
        } else {
            const int pc = c - BE;
            auto E = [&](int t, int idx) { return Ty<T>::ld(tabs[t] + (long)idx * P + pc); };
            e = R(R(E(14, cat) + E(15, mer)) + E(16, col));
        }
        emit(c, e);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void table_embed_kernel(const int* __restrict__ boxes, const T* const* __restrict__ tabs, T* __restrict__ x,
                                                          int Hd, int BE, int bbox_size, int vocab, int category_count, int merge_count) {
    const int b = blockIdx.x;
    table_embed_row<T>(boxes + b * 10, tabs, Hd, BE, bbox_size, vocab, category_count, merge_count,
                       [&](int c, float e) { Ty<T>::st(x + (long)b * Hd + c, e); });
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
// adetr/decoder.py:151-190; no mask, no rotary embedding). kv rows: [B][Lk][2 * nkv * D] = (k heads | v heads) as the fused k|v
// projection writes them. The first version ran one workgroup per (image, kv head) -- B * nkv = 128 workgroups whose P V phase kept
// 64 threads busy for Lk serial steps: 88 us per layer at Lk = 576 (profiles/r03_f_layout_kernel_stats_before.md). Now
// one workgroup per (image, kv head, key range), NS ranges per (image, kv head):
//   * the query row is the sum of the q projection's split-K slabs (qpart [S][M][nq * D] fp32, rounded to T like the unsplit
//     GEMM's output), so that projection needs no reduce launch;
//   * scores: 4 lanes per key (coalesced 2 D-byte row reads), quad-reduced; softmax statistics per range;
//   * P V: the 256 threads are (key slice, head, 4 output dims), slices summed through LDS;
//   * the range's (max, sum, un-normalised output) goes to `scratch`; cross_attn_merge_kernel merges the NS ranges.
template <typename T, int D>
__global__ __launch_bounds__(256) void cross_attn_split_kernel(const float* __restrict__ qpart, int S, int M, const T* __restrict__ kv,
                                                               float* __restrict__ scratch, const int* __restrict__ item_map, int nq, int nkv,
                                                               int Lk, int chunk, float scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int G = nq / nkv;
    float* sc = reinterpret_cast<float*>(smem_raw);                 // [G][chunk] scores -> exp()
    float* qsh = sc + G * chunk;                                    // [G][D]
    float* osh = qsh + G * D;                                       // [256][4] P V partials
    __shared__ float red[2][8][4];                                   // [max | sum][head g][wave]
    const int b = blockIdx.x, kvh = blockIdx.y, sp = blockIdx.z, NS = gridDim.z, tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int row_w = 2 * nkv * D, Hq = nq * D;
    const int j0 = sp * chunk, nk = min(chunk, Lk - j0);
    const T* kb = kv + ((long)item_map[b] * Lk + j0) * row_w + kvh * D;       // row b attends the encoder states of image item_map[b]
    const T* vb = kb + nkv * D;
    for (int i = tid; i < G * D; i += 256) {
        float a = 0.f;
        if (S == 0) a = Ty<T>::ld(reinterpret_cast<const T*>(qpart) + (long)b * Hq + kvh * G * D + i);     // prompt prefill: a plain [M][nq * D] matrix
        for (int s = 0; s < S; ++s) a += qpart[((long)s * M + b) * Hq + kvh * G * D + i];
        qsh[i] = Ty<T>::rnd(a);
    }
    __syncthreads();
    {
        constexpr int DQ = D / 4;                                   // dims per lane of a key's quad
        const int quarter = tid & 3;
        for (int jb = 0; jb < nk; jb += 64) {
            const int j = jb + (tid >> 2), jc = min(j, nk - 1);
            float kr[DQ];
#pragma unroll
            for (int c = 0; c < DQ; c += 4) load4(kb + (long)jc * row_w + quarter * DQ + c, *reinterpret_cast<float(*)[4]>(&kr[c]));
            for (int g = 0; g < G; ++g) {
                float d = 0.f;
#pragma unroll
                for (int c = 0; c < DQ; ++c) d += qsh[g * D + quarter * DQ + c] * kr[c];
                d = quad_sum(d);
                if (quarter == 0 && j < nk) sc[g * chunk + j] = d * scale;
            }
        }
    }
    __syncthreads();
    for (int g = 0; g < G; ++g) {
        float m = -INFINITY;
        for (int j = tid; j < nk; j += 256) m = fmaxf(m, sc[g * chunk + j]);
        m = wave_max(m);
        if (lane == 0) red[0][g][wave] = m;
    }
    __syncthreads();
    for (int g = 0; g < G; ++g) {
        const float m = fmaxf(fmaxf(red[0][g][0], red[0][g][1]), fmaxf(red[0][g][2], red[0][g][3]));
        float l = 0.f;
        for (int j = tid; j < nk; j += 256) { const float e = expf(sc[g * chunk + j] - m); sc[g * chunk + j] = e; l += e; }
        l = wave_sum(l);
        if (lane == 0) red[1][g][wave] = l;
    }
    __syncthreads();
    const int combos = G * (D / 4), nslice = 256 / combos;          // combos <= 128 (G <= 8, D <= 64)
    {
        const int combo = tid % combos, slice = tid / combos;
        const int g = combo / (D / 4), c = (combo % (D / 4)) * 4;
        float o[4] = {0.f, 0.f, 0.f, 0.f};
        if (slice < nslice) {
            int j = slice;
            for (; j + 3 * nslice < nk; j += 4 * nslice) {           // four V rows in flight
                float v0[4], v1[4], v2[4], v3[4];
                load4(vb + (long)j * row_w + c, v0);
                load4(vb + (long)(j + nslice) * row_w + c, v1);
                load4(vb + (long)(j + 2 * nslice) * row_w + c, v2);
                load4(vb + (long)(j + 3 * nslice) * row_w + c, v3);
                const float p0 = sc[g * chunk + j], p1 = sc[g * chunk + j + nslice], p2 = sc[g * chunk + j + 2 * nslice],
                            p3 = sc[g * chunk + j + 3 * nslice];
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] += p0 * v0[i] + p1 * v1[i] + p2 * v2[i] + p3 * v3[i];
            }
            for (; j < nk; j += nslice) {
                float v0[4];
                load4(vb + (long)j * row_w + c, v0);
                const float p0 = sc[g * chunk + j];
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] += p0 * v0[i];
            }
        }
        *reinterpret_cast<float4*>(osh + tid * 4) = make_float4(o[0], o[1], o[2], o[3]);
    }
    __syncthreads();
    // range record per head: [m, l, o[D]]
    const int REC = D + 2;
    if (tid < combos) {
        const int g = tid / (D / 4), c = (tid % (D / 4)) * 4;
        float o[4] = {0.f, 0.f, 0.f, 0.f};
        for (int sl = 0; sl < nslice; ++sl)
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] += osh[(sl * combos + tid) * 4 + i];
        float* rec = scratch + (((long)b * nq + kvh * G + g) * NS + sp) * REC;
        *reinterpret_cast<float2*>(rec + 2 + c) = make_float2(o[0], o[1]);
        *reinterpret_cast<float2*>(rec + 4 + c) = make_float2(o[2], o[3]);
        if (c == 0) {
            rec[0] = fmaxf(fmaxf(red[0][g][0], red[0][g][1]), fmaxf(red[0][g][2], red[0][g][3]));
            rec[1] = red[1][g][0] + red[1][g][1] + red[1][g][2] + red[1][g][3];
        }
    }
}

// Merge of the NS key ranges of cross_attn_split_kernel: one thread per (image, head, 4 output dims). (A ticket counter that let
// the last range's workgroup merge in place cost more than this launch: its device-scope fences write back and invalidate the
// XCD's L2 once per wave, 65 us per layer against 88 for the unsplit kernel.)
template <typename T, int D>
__global__ __launch_bounds__(256) void cross_attn_merge_kernel(const float* __restrict__ scratch, T* __restrict__ out, int rows, int NS) {
    constexpr int REC = D + 2;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * (D / 4)) return;
    const int r = i / (D / 4), c = (i % (D / 4)) * 4;              // r = image * nq + head
    const float* rec = scratch + (long)r * NS * REC;
    float mm = -INFINITY;
    for (int s2 = 0; s2 < NS; ++s2) mm = fmaxf(mm, rec[s2 * REC]);
    float l = 0.f, o[4] = {0.f, 0.f, 0.f, 0.f};
    for (int s2 = 0; s2 < NS; ++s2) {
        const float f = expf(rec[s2 * REC] - mm);
        l += f * rec[s2 * REC + 1];
        const float2 a = *reinterpret_cast<const float2*>(rec + s2 * REC + 2 + c), b = *reinterpret_cast<const float2*>(rec + s2 * REC + 4 + c);
        o[0] += f * a.x; o[1] += f * a.y; o[2] += f * b.x; o[3] += f * b.y;
    }
    const float inv = 1.0f / l;
    store4(out + (long)r * D + c, o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv);
}

// bf16 cross attention on the matrix cores, straight from global memory into MFMA operands (the encoder keys / values of a layer are
// read once per (row, kv head) per step: nothing to stage or share through LDS). 512 threads = 8 waves; wave w owns a contiguous run of
// keys; v_mfma_f32_16x16x32_bf16 with a lane owning head (lane & 15) and, per 16-key block, keys 4g .. 4g + 3 (g = lane >> 4):
//   S^T block = K (16 keys x 32 dims per step) . q^T : A = 16 bytes of a key row (kv rows as the fused k | v projection wrote them),
//                                                      B = 16 bytes of the lane's q head (LDS, summed from the split-K slabs);
//   O^T      += V^T (16 dims x 32 keys) . P^T        : A = 16 bytes = 8 consecutive keys of one dim row of vT, the TRANSPOSED values
//                                                      [row][kv head][D][Lkp] written once per encode by transpose_cross_v_kernel,
//                                                      B = the lane's 8 probabilities -- with the key order of the A operand:
//                                                      element i <-> key 8g + i of the 32-key step, so S^T is computed on keys
//                                                      permuted accordingly (block b, row m = key 8 (m >> 2) + (m & 3) + 4 b ... see kperm).
// All loads of a 3-step chunk (96 keys per wave) are issued before the first MFMA: one memory round trip per chunk. Replaces
// cross_attn_split_kernel + cross_attn_merge_kernel (15 + 4.9 us per layer, profiles/r03_t_layout_table_kernel_stats.md) for bf16.
template <int D>
__global__ __launch_bounds__(512) void cross_attn_mfma_kernel(const float* __restrict__ qpart, int S, int M, const bf16_t* __restrict__ kv,
                                                              const bf16_t* __restrict__ vT, bf16_t* __restrict__ out,
                                                              const int* __restrict__ item_map, int nq, int nkv, int Lk, int Lkp, float scale) {
    constexpr int NW = 8, NKS = D / 32, NDB = D / 16, CW = D + 4, STEPS = 3;
    __shared__ __attribute__((aligned(16))) bf16_t qsh[16 * D];      // q heads of this kv head, rows >= G zero
    __shared__ __attribute__((aligned(16))) float comb[NW * 8 * CW]; // per-wave (O[D], max, sum) of up to 8 heads
    const int b = blockIdx.x, kvh = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = nq / nkv, Hq = nq * D, row_w = 2 * nkv * D, img = item_map[b];
    const int hn = lane & 15, g4 = lane >> 4;
    for (int i = tid; i < 16 * D; i += 512) {
        const int h = i / D, c = i % D;
        float a = 0.f;
        const int hc = min(h, G - 1);                                 // rows >= G: load a valid address, store zero
        const long col = (long)(kvh * G + hc) * D + c;
        if (S == 0) {
            a = bf2f(reinterpret_cast<const bf16_t*>(qpart)[(long)b * Hq + col]);
        } else {                                                      // all slab loads in flight together (a `for s < S` loop is one round trip per slab)
            float p[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) p[s] = qpart[((long)min(s, S - 1) * M + b) * Hq + col];
#pragma unroll
            for (int s = 0; s < 8; ++s) a += (s < S) ? p[s] : 0.f;
        }
        qsh[i] = f2bf(h < G ? a : 0.f);
    }
    __syncthreads();
    u32x4 qf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) qf[ks] = *reinterpret_cast<const u32x4*>(qsh + hn * D + ks * 32 + g4 * 8);
    // keys of this wave: [k_lo, k_hi), a multiple of 32 long except at the end of the sequence
    const int kpw = ((Lk + NW - 1) / NW + 31) & ~31;
    const int k_lo = wave * kpw, k_hi = min(Lk, k_lo + kpw);
    const bf16_t* kbase = kv + (long)img * Lk * row_w + kvh * D;
    const bf16_t* vbase = vT + ((long)img * nkv + kvh) * D * Lkp;
    f32x4 oacc[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db) oacc[db] = f32x4{0.f, 0.f, 0.f, 0.f};
    float mrun = -INFINITY, lrun = 0.f;
    // A-operand row m of S^T block kb (16 keys) of a 32-key step is key  8 (m >> 2) + 4 kb + (m & 3)  of the step, so that the lane's
    // four scores of block kb (rows 4 g4 + i) are keys 8 g4 + 4 kb + i: its 8 probabilities [block 0 | block 1] are keys 8 g4 .. 8 g4 + 7,
    // the k order of the V^T operand's 16 contiguous bytes.
    const int krow = 8 * (hn >> 2) + (hn & 3);
    for (int c0 = k_lo; c0 < k_hi; c0 += 32 * STEPS) {
        u32x4 kf[STEPS][2][NKS], vf[STEPS][NDB];
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            const int s0 = c0 + 32 * st;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const int key = min(s0 + krow + 4 * kb, Lk - 1);                      // clamped: rows past the end are masked below
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks)
                    kf[st][kb][ks] = *reinterpret_cast<const u32x4*>(kbase + (long)key * row_w + ks * 32 + g4 * 8);
            }
            const int vk = min(s0, Lkp - 32) + g4 * 8;                                // vT rows are padded to whole 32-key steps with zeros
#pragma unroll
            for (int db = 0; db < NDB; ++db) vf[st][db] = *reinterpret_cast<const u32x4*>(vbase + (long)(db * 16 + hn) * Lkp + vk);
        }
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            const int s0 = c0 + 32 * st;
            if (s0 >= k_hi) break;                                                    // wave-uniform
            f32x4 sacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
                    sacc[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kf[st][kb][ks]), __builtin_bit_cast(bf16x8, qf[ks]),
                                                                       sacc[kb], 0, 0, 0);
            float bm = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int key = s0 + 8 * g4 + 4 * kb + i;
                    const float sv = key < k_hi ? sacc[kb][i] * scale : -INFINITY;
                    sacc[kb][i] = sv;
                    bm = fmaxf(bm, sv);
                }
            bm = fmaxf(bm, __shfl_xor(bm, 16, 64));
            bm = fmaxf(bm, __shfl_xor(bm, 32, 64));                                   // finite: key s0 < k_hi belongs to g4 = 0
            const float mnew = fmaxf(mrun, bm), alpha = __expf(mrun - mnew);
            mrun = mnew;
            float psum = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float pv = __expf(sacc[kb][i] - mnew);
                    sacc[kb][i] = pv;
                    psum += pv;
                }
            lrun = lrun * alpha + psum;
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int i = 0; i < 4; ++i) oacc[db][i] *= alpha;
            u32x4 pf;
            pf[0] = pack2(sacc[0][0], sacc[0][1]); pf[1] = pack2(sacc[0][2], sacc[0][3]);
            pf[2] = pack2(sacc[1][0], sacc[1][1]); pf[3] = pack2(sacc[1][2], sacc[1][3]);
#pragma unroll
            for (int db = 0; db < NDB; ++db)
                oacc[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, vf[st][db]), __builtin_bit_cast(bf16x8, pf), oacc[db], 0, 0, 0);
        }
    }
    float ltot = lrun + __shfl_xor(lrun, 16, 64);
    ltot += __shfl_xor(ltot, 32, 64);
    if (hn < G) {
        float* rec = comb + (wave * 8 + hn) * CW;
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
        for (int w = 0; w < NW; ++w) mx = fmaxf(mx, comb[(w * 8 + oh) * CW + D]);
        float num[4] = {0.f, 0.f, 0.f, 0.f}, den = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const float* rec = comb + (w * 8 + oh) * CW;
            const float e = (rec[D] == -INFINITY) ? 0.f : __expf(rec[D] - mx);        // waves past the sequence hold (-inf, 0, 0)
            const float4 o4 = *reinterpret_cast<const float4*>(rec + od);
            num[0] += e * o4.x; num[1] += e * o4.y; num[2] += e * o4.z; num[3] += e * o4.w;
            den += e * rec[D + 1];
        }
        const float inv = 1.0f / den;
        store4(out + (long)b * Hq + (long)(kvh * G + oh) * D + od, num[0] * inv, num[1] * inv, num[2] * inv, num[3] * inv);
    }
}

// vT[row][kv head][d][key] <- the v half of the fused k | v projection rows [row][key][2 * nkv * D]; keys >= Lk (padding to whole 32-key
// steps) are zero. Once per encode and layer.
template <typename T>
__global__ __launch_bounds__(256) void transpose_cross_v_kernel(const T* __restrict__ kv, T* __restrict__ vT, int Lk, int Lkp, int nkv, int D) {
    const long n = (long)gridDim.y * nkv * D * Lkp;                  // gridDim.y = images
    const int img = blockIdx.y;
    const long per = (long)nkv * D * Lkp;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < per; i += (long)gridDim.x * 256) {
        const int key = (int)(i % Lkp), d = (int)((i / Lkp) % D), h = (int)(i / ((long)Lkp * D));
        T v = key < Lk ? kv[((long)img * Lk + key) * (2 * nkv * D) + (nkv + h) * D + d] : T(0);
        vT[(long)img * per + i] = v;
    }
    (void)n;
}

// Launch-boundary reduce of a split-K projection of the ADETR decoder, fused with bias, the residual add and the NEXT
// SuryaADETRDecoderRMSNorm (adetr_rmsnorm_kernel's arithmetic):
//   x_out <- T(res + T(bias + sum_s part[s]))          (Linear output rounded, then the residual add, as the unsplit epilogue does)
//   y     <- clamp(x_out * rsqrt(max(mean(x_out^2), eps)) * (1 + w))        (skipped when w == nullptr)
// One workgroup per row, blockDim = H / 4 rounded up to whole waves: every thread owns one 4-element chunk.
template <typename T>
__global__ __launch_bounds__(1024) void splitk_residual_adetr_norm_kernel(const float* __restrict__ part, int S, int M, const T* res,
                                                                          const T* __restrict__ bias, T* x_out,     // res may be x_out
                                                                          const T* __restrict__ w, T* __restrict__ y, int H, float eps) {
    const int row = blockIdx.x, tid = threadIdx.x;
    __shared__ float red[16];
    const int c = tid * 4;
    const bool on_row = c < H;
    const int cc = on_row ? c : 0;
    float r4[4], g[4] = {0.f, 0.f, 0.f, 0.f}, b4[4] = {0.f, 0.f, 0.f, 0.f};
    load4(res + (long)row * H + cc, r4);
    if (w) load4(w + cc, g);
    if (bias) load4(bias + cc, b4);
    f32x4 p4[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) p4[s] = *reinterpret_cast<const f32x4*>(part + ((long)min(s, S - 1) * M + row) * H + cc);
    float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const float on = (s < S) ? 1.f : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] += on * p4[s][i];
    }
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[i] = Ty<T>::rnd(r4[i] + Ty<T>::rnd(v[i] + b4[i]));
        ss += on_row ? v[i] * v[i] : 0.f;
    }
    if (on_row) store4(x_out + (long)row * H + c, v[0], v[1], v[2], v[3]);
    if (!w) return;
    ss = wave_sum(ss);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    float tot = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) tot += red[i];
    const float rstd = rsqrtf(fmaxf(tot / (float)H, eps));
    const float lim = sizeof(T) == 2 ? 3.3895313892515355e38f : 3.4028234663852886e38f;
    float o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float t = v[i] * rstd * (1.0f + g[i]);
        t = fminf(fmaxf(t, -lim), lim);
        o[i] = (t != t) ? 0.f : t;
    }
    if (on_row) store4(y + (long)row * H + c, o[0], o[1], o[2], o[3]);
}

// ---------------------------------------------------------------------------------------------------------------------
// Causal self-attention of a short decoder PROMPT (table recognition: [bos, query, query_end] + one token per column, Tn <= 64 tokens per
// row; SuryaADETRDecoderSdpaAttention with prefill = True, adetr/decoder.py:239-284): RoPE at positions 0 .. Tn - 1, K / V written to the
// cache rows the decode steps continue from, every query attends keys <= its own position. One workgroup per (row, kv head); the prompt is
// tiny (Tn^2 * G * D MACs), so this is plain fp32 VALU work from LDS. qkv rows: [row * Tn + t][(nq + 2 nkv) * D] (no bias in this model).
template <typename T, int D>
__global__ __launch_bounds__(256) void adetr_prefill_attn_kernel(const T* __restrict__ qkv, T* __restrict__ out, T* __restrict__ kc, T* __restrict__ vc,
                                                                 const float2* __restrict__ rope_cs, int Tn, int nq, int nkv, int Tmax, float scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* ksh = reinterpret_cast<float*>(smem_raw);               // [Tn][D]
    float* vsh = ksh + Tn * D;                                      // [Tn][D]
    const int b = blockIdx.x, kvh = blockIdx.y, tid = threadIdx.x, G = nq / nkv, half = D / 2;
    const long qkv_d = (long)(nq + 2 * nkv) * D;
    const T* rows = qkv + (long)b * Tn * qkv_d;
    T* kdst = kc + ((long)b * nkv + kvh) * Tmax * D;
    T* vdst = vc + ((long)b * nkv + kvh) * Tmax * D;
    auto R = [](float v) { return Ty<T>::rnd(v); };
    for (int it = tid; it < Tn * half; it += 256) {
        const int t = it / half, i = it % half;
        const float2 cs = rope_cs[(long)t * half + i];
        const T* kr = rows + t * qkv_d + (long)(nq + kvh) * D;
        const float x1 = Ty<T>::ld(kr + i), x2 = Ty<T>::ld(kr + i + half);
        const float y1 = R(x1 * cs.x - x2 * cs.y), y2 = R(x2 * cs.x + x1 * cs.y);
        ksh[t * D + i] = y1; ksh[t * D + i + half] = y2;
        Ty<T>::st(kdst + (long)t * D + i, y1); Ty<T>::st(kdst + (long)t * D + i + half, y2);
    }
    for (int it = tid; it < Tn * D; it += 256) {
        const int t = it / D, i = it % D;
        const T v = rows[t * qkv_d + (long)(nq + nkv + kvh) * D + i];
        vsh[it] = Ty<T>::ld(&v);
        vdst[(long)t * D + i] = v;
    }
    __syncthreads();
    for (int it = tid; it < Tn * G; it += 256) {                    // one (query position, head) per thread
        const int t = it / G, g = it % G, head = kvh * G + g;
        const T* qr = rows + t * qkv_d + (long)head * D;
        float q[D], o[D];
#pragma unroll
        for (int i = 0; i < half; ++i) {
            const float2 cs = rope_cs[(long)t * half + i];
            const float x1 = Ty<T>::ld(qr + i), x2 = Ty<T>::ld(qr + i + half);
            q[i] = R(R(x1 * cs.x - x2 * cs.y) * scale); q[i + half] = R(R(x2 * cs.x + x1 * cs.y) * scale);
        }
#pragma unroll
        for (int i = 0; i < D; ++i) o[i] = 0.f;
        float m = -INFINITY, l = 0.f;
        for (int j = 0; j <= t; ++j) {
            float sdot = 0.f;
#pragma unroll
            for (int i = 0; i < D; ++i) sdot += q[i] * ksh[j * D + i];
            const float mn = fmaxf(m, sdot), al = __expf(m - mn), pv = __expf(sdot - mn);
            l = l * al + pv;
            const float pr = R(pv);                                 // the decode kernels round P to the storage dtype before P V as well
#pragma unroll
            for (int i = 0; i < D; ++i) o[i] = o[i] * al + pr * vsh[j * D + i];
            m = mn;
        }
        const float inv = 1.0f / l;
        T* op = out + ((long)b * Tn + t) * nq * D + (long)head * D;
#pragma unroll
        for (int i = 0; i < D; i += 4) store4(op + i, o[i] * inv, o[i + 1] * inv, o[i + 2] * inv, o[i + 3] * inv);
    }
}

// row b * Tn + t of a prompt cross-attends the image of decoder row b
__global__ void expand_map_kernel(const int* __restrict__ src, int* __restrict__ dst, int B, int Tn) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B * Tn) dst[i] = src[i / Tn];
}

// ---------------------------------------------------------------------------------------------------------------------
// Output heads of SuryaLayoutDecoder.forward (layout/model/decoder.py:119-131): final ADETR RMSNorm -> LayerNorm -> class logits
// (label_count rows, no bias) and sigmoid(bbox_head). One workgroup per image; every intermediate is rounded to the storage dtype
// where the reference materialises a tensor. class_logits fp32 [B][label_count], bbox fp32 [B][6].
// Round 4: with FB the workgroup goes on to what the reference's host loop does with these outputs (surya/layout/__init__.py:110-131,
// 158-169; surya/table_rec/__init__.py:80-118, shaper.py:12-51) -- it forms the token that is fed back, embeds it and applies the first
// layer's cross_pre_norm -- so the next decode step starts at its first GEMM with no host round trip:
//   layout: token = (trunc(box * bbox_size) x 6, argmax class); a PageHeader / PageFooter whose polygon lies in the middle of its page
//           takes the next-best class (logit of the first choice set to 0, then argmax), the polygon in float64 exactly as
//           prediction_to_polygon computes it (IEEE double operations, no contraction);
//   table:  token = (trunc(clamp(box * bbox_size, 0, bbox_size)) x 6, argmax category, argmax merges, round-half-even(max(colspan, 1)),
//           argmax is_header), the classification values with their special-token offset = the raw argmax.
// The host receives every step's class logits / boxes / fed token from rings and re-derives the token itself: a mismatch is an error.
struct LayoutFeedback {
    int* boxes = nullptr;              // [B][tokw] token of the next step (also read by a host-fed first step)
    int* len = nullptr;                // [B] cache position, incremented once per step
    const int* page_sizes = nullptr;   // layout: [B][2] (width, height) of every slice; nullptr = no header / footer rule
    int* tok_ring = nullptr;           // this step's fed tokens [B][tokw]
    int family = 0, tokw = 7, bbox_size = 1024, vocab = 0, skew_scaler = 512, relabel_a = -1, relabel_b = -1;
    int wcat = 0, wmer = 0, whdr = 0;  // table: widths of the category / merges / is_header heads (colspan has one row between merges and is_header)
    int box_embed = 0, category_count = 0, merge_count = 0, embed_labels = 0;
};

__device__ __forceinline__ int argmax_first(const float* v, int n, int zeroed = -1) {
    int best = 0;
    float bv = zeroed == 0 ? 0.f : v[0];
    for (int i = 1; i < n; ++i) {
        const float x = i == zeroed ? 0.f : v[i];
        if (x > bv) { bv = x; best = i; }
    }
    return best;
}

// The reference's prediction_to_polygon corners 0 and 2 against the page box (surya/layout/util.py:4-40, __init__.py:158-164). The
// corner arithmetic there is TENSOR arithmetic in the model dtype (each operation rounded to T), the final `.item() * scale` and the
// comparisons are Python floats (IEEE double, no contraction).
template <typename T>
__device__ inline bool header_footer_in_page_middle(const float* bp, int pw, int ph, int bbox_size, int skew_scaler) {
#pragma clang fp contract(off)
    auto R = [](float v) { return Ty<T>::rnd(v); };
    const float hw = R(bp[2] / 2.0f), hh = R(bp[3] / 2.0f);
    const float x1 = R(bp[0] - hw), y1 = R(bp[1] - hh), x2 = R(bp[0] + hw), y2 = R(bp[1] + hh);
    float sx = floorf(R(R(bp[4] - (float)skew_scaler) / 2.0f)), sy = floorf(R(R(bp[5] - (float)skew_scaler) / 2.0f));
    if (fabsf(sx) < 0.001f) sx = 0.f;
    if (fabsf(sy) < 0.001f) sy = 0.f;
    const double bs = (double)bbox_size;
    const double w_scale = (double)pw / bs, h_scale = (double)ph / bs;
    const double p0x = (double)R(x1 - sx) * w_scale, p0y = (double)R(y1 - sy) * h_scale;
    const double p2x = (double)R(x2 + sx) * w_scale, p2y = (double)R(y2 + sy) * h_scale;
    return p0y < (double)ph * .8 && p2y > (double)ph * .2 && p0x < (double)pw * .8 && p2x > (double)pw * .2;
}

template <typename T, bool FB = false>
__global__ __launch_bounds__(256) void layout_heads_kernel(const T* __restrict__ x, const T* __restrict__ fnorm_w, const T* __restrict__ ln_w,
                                                           const T* __restrict__ ln_b, const T* __restrict__ lm_w, const T* __restrict__ bb_w,
                                                           const T* __restrict__ bb_b, float* __restrict__ cls, float* __restrict__ box, int Hd,
                                                           int label_count, float rms_eps, float ln_eps, long ldx,
                                                           LayoutFeedback fb = LayoutFeedback(), const T* const* __restrict__ tabs = nullptr,
                                                           T* __restrict__ xn = nullptr, const T* __restrict__ cnorm_w = nullptr,
                                                           T* __restrict__ yn = nullptr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* h = reinterpret_cast<float*>(smem_raw);                  // [Hd]
    __shared__ float red[4];
    __shared__ float outs[FB ? 64 : 1];                             // FB: this row's class logits then its 6 box values
    __shared__ int tok_s[10];
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const T* xr = x + (long)b * ldx;
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
        for (int c = lane * 4; c < Hd; c += 256) {                  // Hd % 4 == 0 (checked by the host)
            float wv[4];
            load4(wr + c, wv);
            d += h[c] * wv[0] + h[c + 1] * wv[1] + h[c + 2] * wv[2] + h[c + 3] * wv[3];
        }
        d = wave_sum(d);
        if (lane == 0) {
            float r;
            if (o < label_count) cls[(long)b * label_count + o] = r = Ty<T>::rnd(d);
            else {
                const float z = Ty<T>::rnd(d + Ty<T>::ld(bb_b + o - label_count));
                box[(long)b * 6 + o - label_count] = r = Ty<T>::rnd(1.0f / (1.0f + expf(-z)));
            }
            if constexpr (FB) outs[o] = r;
        }
    }
    if constexpr (FB) {
        __syncthreads();
        if (tid == 0) {
            const float* cl = outs;
            const float* bx = outs + label_count;
            int tok[10];
            float bp[6];
            for (int i = 0; i < 6; ++i) bp[i] = Ty<T>::rnd(__fmul_rn(bx[i], (float)fb.bbox_size));      // a tensor op in the model dtype
            if (fb.family == 0) {
                for (int i = 0; i < 6; ++i) tok[i] = (int)bp[i];                    // .astype(int64): truncation
                int label = argmax_first(cl, label_count);
                if (fb.page_sizes && (label == fb.relabel_a || label == fb.relabel_b) &&
                    header_footer_in_page_middle<T>(bp, fb.page_sizes[2 * b], fb.page_sizes[2 * b + 1], fb.bbox_size, fb.skew_scaler))
                    label = argmax_first(cl, label_count, label);
                tok[6] = label;
            } else {
                for (int i = 0; i < 6; ++i) tok[i] = (int)fminf(fmaxf(bp[i], 0.f), (float)fb.bbox_size);
                tok[6] = argmax_first(cl, fb.wcat);
                tok[7] = argmax_first(cl + fb.wcat, fb.wmer);
                tok[8] = (int)rintf(fmaxf(cl[fb.wcat + fb.wmer], 1.0f));
                tok[9] = argmax_first(cl + fb.wcat + fb.wmer + 1, fb.whdr);
            }
            for (int i = 0; i < fb.tokw; ++i) {
                tok_s[i] = tok[i];
                fb.boxes[b * fb.tokw + i] = tok[i];
                fb.tok_ring[b * fb.tokw + i] = tok[i];
            }
            fb.len[b] += 1;
        }
        __syncthreads();
        // the fed token's embedding: this row of the residual stream for the next step (h is free: the heads are done with it)
        auto emit = [&](int c, float e) { h[c] = e; Ty<T>::st(xn + (long)b * Hd + c, e); };
        if (fb.family == 0) box_embed_row<T>(tok_s, tabs, Hd, fb.bbox_size, fb.vocab, fb.embed_labels, emit);
        else table_embed_row<T>(tok_s, tabs, Hd, fb.box_embed, fb.bbox_size, fb.vocab, fb.category_count, fb.merge_count, emit);
        __syncthreads();
        if (wave == 0) {                                            // adetr_rmsnorm_kernel's arithmetic and summation order, one wave per row
            float q2 = 0.f;
            for (int c = lane * 4; c < Hd; c += 256) q2 += h[c] * h[c] + h[c + 1] * h[c + 1] + h[c + 2] * h[c + 2] + h[c + 3] * h[c + 3];
            const float r2 = rsqrtf(fmaxf(wave_sum(q2) / (float)Hd, rms_eps));
            for (int c = lane * 4; c < Hd; c += 256) {
                float wv[4], o[4];
                load4(cnorm_w + c, wv);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float t = h[c + i] * r2 * (1.0f + wv[i]);
                    t = fminf(fmaxf(t, -lim), lim);
                    o[i] = (t != t) ? 0.f : t;
                }
                store4(yn + (long)b * Hd + c, o[0], o[1], o[2], o[3]);
            }
        }
    }
}

}  // namespace lay
}  // namespace sa
