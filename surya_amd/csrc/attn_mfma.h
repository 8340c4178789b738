// Segment attention on the matrix cores (bf16): vision windows / whole images (non-causal varlen,
// encoder/__init__.py:238-261) and the decoder prefill (causal GQA over the slot KV cache, decoder/__init__.py:101-128).
// Same interface and tile plan as attn_valu_kernel (kernels.h), which stays the fp32 reference-mode path.
//
// One workgroup = one 64-query tile of one (segment, head); 2 waves, each owning 32 queries. K/V stream through LDS in
// 64-key chunks ([key][d] rows, pitch D + 8 elements so the 16-byte fragment reads are bank-conflict-free).
//   S^T = K Q^T   v_mfma_f32_32x32x16_bf16(K fragment, Q fragment): a lane owns ONE query (lane & 31) and 16 of each 32 keys,
//                 so the online-softmax statistics are per-lane scalars plus one exchange with lane ^ 32 per chunk.
//   O^T = V^T P^T the lane's exp() values ARE its P fragment: for the 16-key step t it holds keys 16t + 4h + r and
//                 16t + 8 + 4h + r (h = lane >> 5), a permutation of the step's 16 keys -- harmless as long as the V^T
//                 fragment uses the same key order, which two ds_read_b64_tr_b16 transpose reads (4 keys x 16 d each per
//                 16-lane group) deliver straight from the row-major V tile (semantics checked by tools/microbench/tr_probe).
//                 O accumulates per lane for the same query, so the rescale by exp(m_old - m_new) is a per-lane scalar too.
// Scores stay fp32 (the reference rounds QK^T to bf16 before its fp32 softmax); P is rounded to bf16 before PV as the
// reference does (softmax(...).to(q.dtype)), un-normalised here, normalised there: covered by the bf16 tolerance tests.
#pragma once
#include "common.h"
#include "kernels.h"

namespace sa {

typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int D>
__global__ __launch_bounds__(128) void attn_mfma_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                        const bf16_t* __restrict__ v, bf16_t* __restrict__ out, AttnSegs sg,
                                                        long q_row, long q_head, long k_row, long k_head, long o_row, long o_head,
                                                        int group, int causal, float scale) {
    static_assert(D % 16 == 0 && D <= 128, "head dim");
    constexpr int KC = 64;                     // keys per LDS chunk
    constexpr int PK = D + 8;                  // LDS row pitch in elements
    constexpr int NKK = D / 16;                // QK^T MFMA steps over the head dim
    constexpr int NDB = (D + 31) / 32;         // 32-wide output blocks over the head dim (D = 80: the last one is half used)
    constexpr int CPR = D / 8;                 // 16-byte chunks per K/V row
    __shared__ __attribute__((aligned(16))) bf16_t ks[(KC + 1) * PK];
    __shared__ __attribute__((aligned(16))) bf16_t vs[(KC + 1) * PK];   // +1 row: the padded d-block of D = 80 reads past a row

    const int tile = blockIdx.x, head = blockIdx.y, kvh = head / group;
    const int seg = sg.tile_seg[tile], q0 = sg.tile_q0[tile], L = sg.seg_len[seg];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ql = lane & 31, h = lane >> 5;
    const int qi = q0 + wave * 32 + ql;        // this lane's query (segment-local)

    // Q fragments: 16-byte chunk (kk * 2 + h) of the query row, straight from global memory
    u32x4 qf[NKK];
    {
        const bf16_t* qp = q + sg.q_off[seg] + (long)min(qi, L - 1) * q_row + (long)head * q_head + h * 8;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) qf[kk] = *reinterpret_cast<const u32x4*>(qp + kk * 16);
    }
    f32x16 oacc[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[db][r] = 0.f;
    float mrun = -INFINITY, lrun = 0.f;        // lrun: this lane's share of the row sum (its 16 of every 32 keys)
    const float sl2 = scale * 1.44269504088896340736f;

    const int kend = causal ? min(L, q0 + 64) : L;
    const bf16_t* kbase = k + sg.k_off[seg] + (long)kvh * k_head;
    const bf16_t* vbase = v + sg.v_off[seg] + (long)kvh * k_head;
    // transpose-read addressing: 16-lane group gi covers d columns (gi & 1) * 16 .. + 15 of the 32-wide d-block for key half
    // h = gi >> 1; lane i of the group supplies the address of key (i >> 2), columns (i & 3) * 4 .. + 3
    const int tr_off = ((lane & 15) >> 2) * PK + ((lane >> 4) & 1) * 16 + (lane & 3) * 4 + h * 4 * PK;

    // K/V chunks travel global -> registers -> LDS, one chunk AHEAD: the loads of chunk kc + KC are issued (all of them, unconditional,
    // row index clamped) before chunk kc is multiplied and are written to LDS after it. Round 2 loaded each 16-byte piece under
    // `if (r < nk)` right where it was stored: hipcc branched around every load and waited vmcnt(0) behind it -- 5-8 dependent
    // round trips per chunk, all exposed (r03 ISA reading of gemm.h's epilogue; same pattern).
    constexpr int NST = KC * CPR / 128;                    // 16-byte pieces per thread per chunk (per K and per V)
    static_assert(KC * CPR % 128 == 0, "staging split");
    u32x4 kreg[NST], vreg[NST];
    auto fetch = [&](int kc) {
        const int nk = min(KC, kend - kc);
#pragma unroll
        for (int it = 0; it < NST; ++it) {
            const int c = tid + it * 128, r = min(c / CPR, nk - 1), cc = c % CPR;
            kreg[it] = *reinterpret_cast<const u32x4*>(kbase + (long)(kc + r) * k_row + cc * 8);
            vreg[it] = *reinterpret_cast<const u32x4*>(vbase + (long)(kc + r) * k_row + cc * 8);
        }
    };
    if (kend > 0) fetch(0);
    for (int kc = 0; kc < kend; kc += KC) {
        const int nk = min(KC, kend - kc);
#pragma unroll
        for (int it = 0; it < NST; ++it) {
            const int c = tid + it * 128, r = c / CPR, cc = c % CPR;
            const bool in = r < nk;                          // rows past the segment are zero: 0 * garbage must not be NaN
            const u32x4 z = {0u, 0u, 0u, 0u};
            *reinterpret_cast<u32x4*>(ks + r * PK + cc * 8) = in ? kreg[it] : z;
            *reinterpret_cast<u32x4*>(vs + r * PK + cc * 8) = in ? vreg[it] : z;
        }
        __syncthreads();
        if (kc + KC < kend) fetch(kc + KC);                  // wave-uniform; in flight during the MFMAs below

        // scores: two 32-key blocks
        f32x16 sacc[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[kb][r] = 0.f;
            const bf16_t* kp = ks + (kb * 32 + ql) * PK + h * 8;
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                const u32x4 kf = *reinterpret_cast<const u32x4*>(kp + kk * 16);
                sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, qf[kk]),
                                                                   sacc[kb], 0, 0, 0);
            }
        }
        // mask + running max (register 4g + r of block kb = key kb * 32 + g * 8 + h * 4 + r)
        float bm = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = kb * 32 + g * 8 + h * 4 + r;
                    const bool ok = (j < nk) && (!causal || (kc + j) <= qi);
                    const float sv = ok ? sacc[kb][4 * g + r] : -INFINITY;
                    sacc[kb][4 * g + r] = sv;
                    bm = fmaxf(bm, sv);
                }
        bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
        const float mnew = fmaxf(mrun, bm);                 // finite: every chunk holds at least one visible key per query
        const float alpha = exp2f((mrun - mnew) * sl2);
        mrun = mnew;
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = exp2f((sacc[kb][r] - mnew) * sl2);      // exp(-inf) = 0 for masked keys
                sacc[kb][r] = pv;
                psum += pv;
            }
        lrun = lrun * alpha + psum;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;

        // O^T += V^T P^T over four 16-key steps
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int kb = t >> 1, o8 = (t & 1) * 8;
            u32x4 pf;
            pf[0] = pack2(sacc[kb][o8 + 0], sacc[kb][o8 + 1]);
            pf[1] = pack2(sacc[kb][o8 + 2], sacc[kb][o8 + 3]);
            pf[2] = pack2(sacc[kb][o8 + 4], sacc[kb][o8 + 5]);
            pf[3] = pack2(sacc[kb][o8 + 6], sacc[kb][o8 + 7]);
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
                const bf16_t* vp = vs + t * 16 * PK + db * 32 + tr_off;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vp));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vp + 8 * PK));
                typedef short s16x8 __attribute__((ext_vector_type(8)));
                const s16x8 vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf), __builtin_bit_cast(bf16x8, pf),
                                                                   oacc[db], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    const float ltot = lrun + __shfl_xor(lrun, 32, 64);
    if (qi < L) {
        const float inv = 1.0f / ltot;
        bf16_t* op = out + sg.o_off[seg] + (long)qi * o_row + (long)head * o_head;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = db * 32 + g * 8 + h * 4;
                if (d0 < D) store4(op + d0, oacc[db][4 * g] * inv, oacc[db][4 * g + 1] * inv, oacc[db][4 * g + 2] * inv,
                                   oacc[db][4 * g + 3] * inv);
            }
    }
}

}  // namespace sa
