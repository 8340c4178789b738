// The folded decode head on the matrix cores (surya/detection/model/encoderdecoder.py:699-722 in the folded form of detection/plan.py):
//   y[px][ch] = relu( A0 x0[px] + c + up2(z1)[px] + up4(z2)[px] + up8(z3)[px] ),   out[px][l] = sigmoid( w_l . y[px] + b_l ).
// head_z0_kernel (det_fused.h) computes z0 on the MFMA and everything else on the vector ALU at one wave per SIMD (468 registers: the tap tiles of a
// 4 x 2 pixel block per lane): ~700 us per 16 pages of 1024^2, the largest launch of the forward. Here the three bilinear up-samplings are what they
// are algebraically -- a CONSTANT linear map from the coarse taps around an aligned pixel tile to its pixels -- and ride on the same accumulators as z0:
//   workgroup tile = 8 x 16 pixels of the 1/4-resolution map (aligned to 8 / 16, so the coarse cells of all three scales line up); its taps are
//     6 x 10 pixels of z1 (2x coarser), 4 x 6 of z2, 3 x 4 of z3 = 96 = six K steps of 16; border taps are fetched from clamped coordinates, which
//     reproduces PyTorch's align_corners = False clamping with the SAME weights on every tile;
//   D^T[ch][px] = A0 . x0^T (K = 64) + Z^T . Wt^T (K = 96) per 128-channel slab: 4 waves x 32 pixels x 4 channel tiles, 40 MFMAs per wave and slab.
//     Wt[px][tap] = wy . wx (products of k/16 fractions: exact in bf16) is made once per lane at kernel start; the taps travel as [tap][128 ch] rows through
//     a two-buffer LDS ring (global_load_lds, 16-byte chunks XOR-swizzled by the row) and come out as MFMA A fragments through ds_read_b64_tr_b16 (the
//     V^T recipe of attn_mfma.h); A0 comes fragment-major (det_mbconv.h), re-requested for the next slab right after its last use;
//   epilogue per slab on the accumulators: + c, ReLU, round to bf16, the two classifier dot products (v_dot2_f32_bf16); after four slabs the two halves of a
//     pixel's channels meet by one lane exchange; + b_l, round, sigmoid, round -- the op list's arithmetic from the ReLU on.
// Against the op list: z0 is not rounded to bf16 on its own and the interpolation sums run in the MFMA's order instead of the separable fp32 chain -- a
// re-association of fp32 sums (the class of the two LiteMLA forms), NOT bit-identical; tests/test_gpu_det_fused.py bounds it against the op list and
// the fp32 oracle.
#pragma once
#include "det_fused.h"
#include "attn_mfma.h"

namespace sa {

#ifndef SA_HEAD_ASM_DMA
#define SA_HEAD_ASM_DMA 1
#endif

__device__ __forceinline__ void head_axis_weights(int p, int R, int& i0, float& f) {
    // fine pixel p of the tile (tile origin aligned to R): source coordinate relative to the tile's first coarse cell
    const float src = ((float)p + 0.5f) / (float)R - 0.5f;
    const float fl = floorf(src);
    i0 = (int)fl;                  // -1 .. : tap rows i0 and i0 + 1 (tap index = i0 + 1, i0 + 2 in a tile that starts one cell early)
    f = src - fl;
}

__global__ __launch_bounds__(256, 3) void head_mfma_kernel(const bf16_t* __restrict__ x0, const bf16_t* __restrict__ A0f, const bf16_t* __restrict__ zb,
                                                           const bf16_t* __restrict__ z1, const bf16_t* __restrict__ z2, const bf16_t* __restrict__ z3,
                                                           const bf16_t* __restrict__ w, const bf16_t* __restrict__ bias, float* __restrict__ out,
                                                           int B, int H0, int W0, int C, int L, int ntiles) {
    // Second version (the first held a 128-channel slab's A0 fragments in registers: 16 KiB requested by EACH of the four waves per slab -- 22
    // vector-memory requests per wave and step, 88 KiB through the CU's 64 B/clk L1 path per workgroup-step against 2560 cycles of MFMA -- and ran two
    // workgroups per CU at 250 registers): 64-channel slabs, the taps AND the slab's A0 fragments through two-buffer LDS rings (5 LDS-DMA requests
    // per wave and step, 20 KiB per workgroup-step), 32 accumulator registers, three workgroups per CU.
    constexpr int K0 = 64, TH = 8, TW = 16, NT = 96, SLAB = 64;
    constexpr int ZB = NT * SLAB * 2;                                     // one tap buffer [96 taps][64 ch] bf16 = 12 KiB
    constexpr int AB = 2 * 4 * 1024;                                      // one A0 buffer: [2 channel tiles][4 K steps][64 lanes][16 bytes] = 8 KiB
    constexpr int OFF_A = 2 * ZB, OFF_TAB = OFF_A + 2 * AB;               // then c | w0 | w1 as bf16 [C] each
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 31, lh = lane >> 5;
    const int nslab = C / SLAB;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    for (int i = tid; i < C / 8; i += 256) {
        *reinterpret_cast<uint4*>(smem + OFF_TAB + i * 16) = *reinterpret_cast<const uint4*>(zb + i * 8);
        *reinterpret_cast<uint4*>(smem + OFF_TAB + C * 2 + i * 16) = *reinterpret_cast<const uint4*>(w + i * 8);
        *reinterpret_cast<uint4*>(smem + OFF_TAB + C * 4 + i * 16) = *reinterpret_cast<const uint4*>(w + (long)(L > 1 ? 1 : 0) * C + i * 8);
    }
    // ---- this lane's pixel of the tile and its interpolation row Wt[m][0..95] as B fragments in the tr-read's tap order: element e of step s is tap
    // 16 s + 4 lh + e (e < 4) or 16 s + 8 + 4 lh + (e - 4). Tap order: z1's 6 x 10, then z2's 4 x 6, then z3's 3 x 4.
    const int m = wv * 32 + lr, py = m >> 4, px = m & 15;
    u32x4 wt[6];
    {
        int iy[3], ix[3];
        float fy[3], fx[3];
        head_axis_weights(py, 2, iy[0], fy[0]); head_axis_weights(px, 2, ix[0], fx[0]);
        head_axis_weights(py, 4, iy[1], fy[1]); head_axis_weights(px, 4, ix[1], fx[1]);
        head_axis_weights(py, 8, iy[2], fy[2]); head_axis_weights(px, 8, ix[2], fx[2]);
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = 16 * s + (e < 4 ? 4 * lh + e : 8 + 4 * lh + (e - 4));
                const int sc = k < 60 ? 0 : (k < 84 ? 1 : 2), kk = k - (sc == 0 ? 0 : (sc == 1 ? 60 : 84)), nc = sc == 0 ? 10 : (sc == 1 ? 6 : 4);
                const int ty = kk / nc, tx = kk - ty * nc;
                const int iys = sc == 0 ? iy[0] : (sc == 1 ? iy[1] : iy[2]), ixs = sc == 0 ? ix[0] : (sc == 1 ? ix[1] : ix[2]);
                const float fys = sc == 0 ? fy[0] : (sc == 1 ? fy[1] : fy[2]), fxs = sc == 0 ? fx[0] : (sc == 1 ? fx[1] : fx[2]);
                const int dy = ty - (iys + 1), dx = tx - (ixs + 1);
                const float wy = dy == 0 ? 1.0f - fys : (dy == 1 ? fys : 0.0f), wx = dx == 0 ? 1.0f - fxs : (dx == 1 ? fxs : 0.0f);
                v[e] = wy * wx;
            }
            wt[s] = u32x4{pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
        }
    }
    // ---- tap requests: request i of this wave (q = wv * 3 + i) fills LDS rows q * 8 + (lane >> 3) of the tap buffer (128-byte rows), physical chunk
    // lane & 7 <- logical chunk (lane & 7) ^ (((row >> 1) & 1) << 1) (rows r, r + 2 of a transposed read share banks otherwise). Request 7 straddles
    // z1 / z2 (tap 60), request 10 z2 / z3 (tap 84): the source pointer is per lane.
    const int h1 = H0 / 2, w1 = W0 / 2, h2 = H0 / 4, w2 = W0 / 4, h3 = H0 / 8, w3 = W0 / 8;
    const int tiles_x = (W0 + TW - 1) / TW, tiles_y = H0 / TH;      // W0 is a multiple of 8: the last tile of a row may be half outside (clamped loads, no stores)
    const bf16_t* zptr[3];
    const int zchunk = (((lane & 7) ^ ((((lane >> 3) >> 1) & 1) << 1)) << 3);     // element offset of this lane's 16 bytes inside a 64-channel slab row
    // tr-read addressing (attn_mfma.h): 16-lane group gi covers channels (gi & 1) * 16 .. + 15 of a 32-channel tile for tap half lh; lane i of the group
    // supplies the address of tap row (i >> 2), channels (i & 3) * 4 .. + 3. Row pitch 128 bytes; the chunk XOR flips bit 1 only (never the tile bit).
    int troff[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int row = ((lane & 15) >> 2) + lh * 4 + hh * 8;              // within a 16-tap step (16 s keeps (row >> 1) & 1)
        const int col = ((lane >> 4) & 1) * 16 + (lane & 3) * 4;             // channel inside the 32-channel tile
        troff[hh] = row * 128 + (((col >> 3) ^ (((row >> 1) & 1) << 1)) << 4) + ((col & 7) << 1);
    }
    const int xcd = blockIdx.x & 7, gx = (int)gridDim.x >> 3, wx_ = (int)blockIdx.x >> 3;
    const int per = ntiles >> 3, rem = ntiles & 7;
    const int t_begin = xcd * per + min(xcd, rem), t_cnt = per + (xcd < rem ? 1 : 0);
    const long HW = (long)H0 * W0;
    const float blv[2] = {Ty<bf16_t>::ld(bias), Ty<bf16_t>::ld(bias + (L > 1 ? 1 : 0))};

#define HM_TILE(TL, IMG, Y0, X0)                                                                                        \
    {                                                                                                                   \
        const int bid_ = t_begin + (TL);                                                                                \
        IMG = bid_ / (tiles_x * tiles_y);                                                                               \
        const int tr_ = bid_ - IMG * tiles_x * tiles_y;                                                                 \
        Y0 = (tr_ / tiles_x) * TH; X0 = (tr_ % tiles_x) * TW;                                                           \
    }
    // tap pixel of request i for the tile at (Y0, X0): scale by the tap index, coordinates clamped into the coarse map
#define HM_ROWS(IMG, Y0, X0)                                                                                            \
    {                                                                                                                   \
        _Pragma("unroll") for (int i_ = 0; i_ < 3; ++i_) {                                                              \
            const int t_ = (wv * 3 + i_) * 8 + (lane >> 3);                                                             \
            const int sc_ = t_ < 60 ? 0 : (t_ < 84 ? 1 : 2), kk_ = t_ - (sc_ == 0 ? 0 : (sc_ == 1 ? 60 : 84));          \
            const int nc_ = sc_ == 0 ? 10 : (sc_ == 1 ? 6 : 4), hs_ = sc_ == 0 ? h1 : (sc_ == 1 ? h2 : h3), ws_ = sc_ == 0 ? w1 : (sc_ == 1 ? w2 : w3); \
            const int ty_ = kk_ / nc_, tx_ = kk_ - ty_ * nc_;                                                           \
            const int cy_ = min(max(((Y0) >> (sc_ + 1)) - 1 + ty_, 0), hs_ - 1), cx_ = min(max(((X0) >> (sc_ + 1)) - 1 + tx_, 0), ws_ - 1); \
            zptr[i_] = (sc_ == 0 ? z1 : (sc_ == 1 ? z2 : z3)) + ((((IMG) * hs_ + cy_) * ws_ + cx_) * C + zchunk);      \
        }                                                                                                               \
    }
    // The requests are inline asm: behind __builtin_amdgcn_global_load_lds hipcc orders EVERY later LDS read of the step behind the pending LDS-DMA
    // (s_waitcnt vmcnt(0) in front of the first ds_read: it cannot see that the request fills the OTHER buffer). The asm form is invisible to that
    // bookkeeping; the wait at the top of the next step is explicit anyway.
#define HM_DMA(G, L)                                                                                                    \
    {                                                                                                                   \
        const void* g_ = (G);                                                                                           \
        const unsigned l_ = (L);                                                                                        \
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g_), "s"(l_) : "memory", "m0"); \
    }
#define HM_ISSUE(BUF, SLB)                                                                                              \
    {                                                                                                                   \
        _Pragma("unroll") for (int i_ = 0; i_ < 3; ++i_) HM_DMA(zptr[i_] + (SLB) * SLAB, lds0 + (unsigned)((BUF) * ZB + (wv * 3 + i_) * 1024)); \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                                                                \
            HM_DMA(A0f + ((long)(SLB) * 8 + wv * 2 + i_) * 512 + lane * 8, lds0 + (unsigned)(OFF_A + (BUF) * AB + (wv * 2 + i_) * 1024)); \
    }
#define HM_LOADX(XF, IMG, Y0, X0)                                                                                       \
    {                                                                                                                   \
        const bf16_t* xp_ = x0 + (((long)(IMG) * H0 + (Y0) + py) * W0 + min((X0) + px, W0 - 1)) * K0 + lh * 8;           \
        _Pragma("unroll") for (int ks_ = 0; ks_ < 4; ++ks_) XF[ks_] = *reinterpret_cast<const u32x4*>(xp_ + ks_ * 16);  \
    }

    if (wx_ >= t_cnt) return;                                // (uniform per workgroup; before any barrier)
    int img, y0, x0c;
    HM_TILE(wx_, img, y0, x0c);
    HM_ROWS(img, y0, x0c);
    HM_ISSUE(0, 0);
    u32x4 xf[4];
    HM_LOADX(xf, img, y0, x0c);
    int step = 0;
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    for (int tl = wx_; tl < t_cnt; tl += gx) {
        float p0 = 0.f, p1 = 0.f;
        const bool more_tiles = tl + gx < t_cnt;
        int nimg = img, ny0 = y0, nx0 = x0c;
        if (more_tiles) HM_TILE(tl + gx, nimg, ny0, nx0);
        for (int sl = 0; sl < nslab; ++sl, ++step) {
            const int buf = step & 1;
            // this step's taps and A0 fragments (and x0 fragments) have landed. The BUILTIN (not asm): hipcc's own counter bookkeeping then knows the x0
            // fragments' loads are complete. With an asm wait it kept them pending and put s_waitcnt vmcnt(0) in front of the step's first MFMA -- behind
            // the five LDS-DMA requests just issued for the NEXT step (invisible to it), every step: the ring never ran ahead
            __builtin_amdgcn_s_waitcnt(0x0F70);
            __syncthreads();                                 // ... for every wave, and every wave is done with the other buffers
            const bool last = sl + 1 == nslab;
            if (last) HM_ROWS(nimg, ny0, nx0);               // the next step belongs to the next tile (re-reads this tile at the very end: harmless)
            HM_ISSUE(buf ^ 1, last ? 0 : sl + 1);
            const unsigned char* as = smem + OFF_A + buf * AB + lane * 16;
            const unsigned char* zs = smem + buf * ZB;
            f32x16 acc[2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
            // z0 part: A0 fragments from the LDS ring (fragment-major: conflict-free 16-byte reads), all eight requested at once
            {
                u32x4 af[2][4];
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) af[j][ks] = *reinterpret_cast<const u32x4*>(as + (j * 4 + ks) * 1024);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[j][ks]), __builtin_bit_cast(bf16x8, xf[ks]), acc[j], 0, 0, 0);
            }
            if (last) HM_LOADX(xf, nimg, ny0, nx0);          // (this tile's last use of xf was the z0 part above)
            // interpolation part: Z^T fragments of (step, tile) f + 1 are read while f multiplies
#define HM_RZ(ZF, F)                                                                                                    \
    {                                                                                                                   \
        const unsigned char* vp_ = zs + ((F) >> 1) * 16 * 128 + ((F) & 1) * 64;                                         \
        const s16x4 lo_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vp_ + troff[0])); \
        const s16x4 hi_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vp_ + troff[1])); \
        ZF = __builtin_shufflevector(lo_, hi_, 0, 1, 2, 3, 4, 5, 6, 7);                                                 \
    }
#define HM_MZ(ZF, F) acc[(F) & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ZF), __builtin_bit_cast(bf16x8, wt[(F) >> 1]), acc[(F) & 1], 0, 0, 0);
            {
                s16x8 za, zb2;
                HM_RZ(za, 0);
#pragma unroll
                for (int f = 0; f < 12; f += 2) {
                    HM_RZ(zb2, f + 1);
                    HM_MZ(za, f);
                    if (f + 2 < 12) HM_RZ(za, f + 2);
                    HM_MZ(zb2, f + 1);
                }
            }
#undef HM_RZ
#undef HM_MZ
            // ---- + c, ReLU, round, classifier dot products: lane = pixel m, quad q of tile j = channels sl * 64 + j * 32 + 8 q + 4 lh + (0..3); a
            // tile's twelve table reads are requested together
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                uint2 cb[4], wa[4], wb[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c0 = (sl * SLAB + j * 32 + q * 8 + lh * 4) * 2;
                    cb[q] = *reinterpret_cast<const uint2*>(smem + OFF_TAB + c0);
                    wa[q] = *reinterpret_cast<const uint2*>(smem + OFF_TAB + C * 2 + c0);
                    wb[q] = *reinterpret_cast<const uint2*>(smem + OFF_TAB + C * 4 + c0);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float bq[4];
                    load4(reinterpret_cast<const bf16_t*>(&cb[q]), bq);
                    const f32x2 v01 = f32x2{fmaxf(acc[j][4 * q] + bq[0], 0.f), fmaxf(acc[j][4 * q + 1] + bq[1], 0.f)};
                    const f32x2 v23 = f32x2{fmaxf(acc[j][4 * q + 2] + bq[2], 0.f), fmaxf(acc[j][4 * q + 3] + bq[3], 0.f)};
                    const bf16x2_t y01 = __builtin_convertvector(v01, bf16x2_t), y23 = __builtin_convertvector(v23, bf16x2_t);
                    p0 = __builtin_amdgcn_fdot2_f32_bf16(y01, __builtin_bit_cast(bf16x2_t, wa[q].x), p0, false);
                    p0 = __builtin_amdgcn_fdot2_f32_bf16(y23, __builtin_bit_cast(bf16x2_t, wa[q].y), p0, false);
                    p1 = __builtin_amdgcn_fdot2_f32_bf16(y01, __builtin_bit_cast(bf16x2_t, wb[q].x), p1, false);
                    p1 = __builtin_amdgcn_fdot2_f32_bf16(y23, __builtin_bit_cast(bf16x2_t, wb[q].y), p1, false);
                }
            }
        }
        // ---- the two channel halves of the pixel meet; + b_l, round, sigmoid, round; lane half l writes label l
        p0 += __shfl_xor(p0, 32, 64);
        p1 += __shfl_xor(p1, 32, 64);
        if (lh < L && x0c + px < W0) {
            const float z = Ty<bf16_t>::rnd((lh ? p1 : p0) + blv[lh]);
            out[((long)img * L + lh) * HW + (long)(y0 + py) * W0 + x0c + px] = Ty<bf16_t>::rnd(1.0f / (1.0f + expf(-z)));
        }
        img = nimg; y0 = ny0; x0c = nx0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the trailing requests land before the workgroup's LDS is released
#undef HM_TILE
#undef HM_ROWS
#undef HM_DMA
#undef HM_ISSUE
#undef HM_LOADX
}

static inline bool head_mfma_shape_ok(int H0, int W0, int K, int C, int L) {
    return K == 64 && C % 64 == 0 && C <= 1024 && L >= 1 && L <= 2 && H0 % 8 == 0 && W0 % 8 == 0 && H0 >= 16 && W0 >= 32;
}

static inline int launch_head_mfma(const bf16_t* x0, const bf16_t* A0f, const bf16_t* zb, const bf16_t* z1, const bf16_t* z2, const bf16_t* z3,
                                   const bf16_t* w, const bf16_t* bias, float* planes, int B, int H0, int W0, int K, int C, int L, hipStream_t s) {
    if (!head_mfma_shape_ok(H0, W0, K, C, L) || !zb || (long)B * (H0 / 2) * (W0 / 2) * C >= (1L << 31)) return SA_ERR_SHAPE;
    const int ntiles = B * (H0 / 8) * ((W0 + 15) / 16);
    const size_t lds = 2 * 96 * 128 + 2 * 8192 + (size_t)C * 6;
    auto kern = head_mfma_kernel;
    static AttrOnce attr;
    attr.ensure(kern, lds);
    int dev = 0, n_cu = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
    const unsigned grid = (unsigned)std::max(8, std::min(ntiles, 3 * n_cu) / 8 * 8);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, x0, A0f, zb, z1, z2, z3, w, bias, planes, B, H0, W0, C, L, ntiles);
    return (int)hipGetLastError();
}
template <typename T>
static inline int launch_head_mfma(const T*, const T*, const T*, const T*, const T*, const T*, const T*, const T*, float*, int, int, int, int, int,
                                   int, hipStream_t) { return SA_ERR_UNSUPPORTED; }

}  // namespace sa
