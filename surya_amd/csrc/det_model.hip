// Text-detection forward pass on MI355X: an op-list interpreter over NHWC activation buffers.
// The Python loader (surya_amd/detection/plan.py) walks the EfficientViT-L + decode-head structure
// (surya/detection/model/encoderdecoder.py:484-753), folds BatchNorm into weights/bias, lays the weights out for
// the kernels and emits one `surya_det_op` per launch; this file owns the buffers and sequences the kernels.
#include <algorithm>
#include <cstring>
#include <memory>
#include <vector>

#include "../../include/surya_amd.h"
#include "det_kernels.h"
#include "det_fused.h"
#include "det_mbconv.h"
#include "det_head.h"
#include "det_post.h"
#include "resample.h"

namespace sa {

struct DetBase {
    virtual ~DetBase() {}
    virtual int forward(const float* pixels, const unsigned char* pixels_u8, const float* mean_std, int B, float* heat, float* lowres,
                        hipStream_t s, int pix = 3, float* op_ms = nullptr) = 0;
    virtual int n_ops() const = 0;
};

template <typename T>
struct DetModel : DetBase {
    std::vector<surya_det_op> ops;
    std::vector<const void*> w;
    std::vector<size_t> buf_elems;       // per image
    std::vector<T*> bufs;
    float* planes = nullptr;             // [max_batch, labels, H/4, W/4] fp32
    float* kvp = nullptr;                // LiteMLA partial kv matrices [max_batch, heads, chunks, 32*33] fp32
    T* zero_page = nullptr;              // 256 zero bytes: source of the convolution gather outside the image
    char* arena = nullptr;
    int max_batch = 0, H = 0, W = 0, labels = 0, in_cp = 8;
    std::vector<hipEvent_t> evs;         // per-op timing (surya_det_forward_timed): one event in front of every op + one behind the last

    int init(const surya_det_config& c, const surya_det_op* o, const void* const* weights, int n_weights, const size_t* be,
             int n_bufs) {
        ops.assign(o, o + c.n_ops);
        w.assign(weights, weights + n_weights);
        buf_elems.assign(be, be + n_bufs);
        max_batch = c.max_batch; H = c.height; W = c.width; labels = c.num_labels;
        size_t total = 0;
        std::vector<size_t> offs(n_bufs);
        for (int i = 0; i < n_bufs; ++i) { offs[i] = total; total += ((buf_elems[i] * max_batch * sizeof(T)) + 255) & ~(size_t)255; }
        const size_t planes_off = total;
        total += (size_t)max_batch * labels * (H / 4) * (W / 4) * sizeof(float) + 256;
        const size_t kvp_off = total;
        size_t kv_floats = 0;
        for (const surya_det_op& op : ops)
            if (op.type == SA_DET_LITEMLA)
                kv_floats = std::max(kv_floats, (size_t)max_batch * (op.cout / op.p0) * cdiv(op.hin * op.win, LITEMLA_CHUNK) * LITEMLA_KV);
        total += kv_floats * sizeof(float) + 256;
        const size_t zero_off = total;
        total += 256;
        SA_HIP(hipMalloc((void**)&arena, total));
        poison_arena(arena, total);
        SA_HIP(hipMemset(arena + zero_off, 0, 256));
        zero_page = reinterpret_cast<T*>(arena + zero_off);
        bufs.resize(n_bufs);
        for (int i = 0; i < n_bufs; ++i) bufs[i] = reinterpret_cast<T*>(arena + offs[i]);
        planes = reinterpret_cast<float*>(arena + planes_off);
        kvp = reinterpret_cast<float*>(arena + kvp_off);
        find_fusions();
        return prepare_fused_weights();
    }
    ~DetModel() override {
        for (hipEvent_t e : evs) (void)hipEventDestroy(e);
        for (T* p : mb_w2f) if (p) (void)hipFree(p);
        for (T* p : head_a0f) if (p) (void)hipFree(p);
        if (arena) (void)hipFree(arena);
    }

    const T* WT(int idx) const { return idx < 0 ? nullptr : reinterpret_cast<const T*>(w[idx]); }

    int n_ops() const override { return (int)ops.size(); }

    // ---- fused forms (round 6), found once by a peephole pass over the op list. The list itself is unchanged (every op still owns its
    // output buffer), so sa::Tuning::det_fuse can switch each form on and off at run time: the op-by-op path stays the checker of
    // tests/test_gpu_det_fused.py and the A/B arm of tools/det_op_times.py.
    //   bit 0  depthwise 5x5 + grouped 1x1 of LiteMLA's multi-scale branch in one kernel (the depthwise result lives in LDS only)
    //   bit 1  LiteMLA's kv reduction and output in one launch per (image, head) on the fp32 MFMA
    //   bit 2  the full-resolution stage's 1x1 convolution z0 inside the sum + classify pass (the [P, 512] tensor is never written)
    //   bit 3  MBConv's depthwise 3x3 + projection 1x1 in one kernel (the depthwise result is the projection's A operand, in LDS only)
    //   bit 4  FusedMBConv's 3x3 expand + Hardswish + 1x1 projection in one kernel (the expanded tensor exists per 64-channel chunk, in registers / LDS only)
    //   bit 5  (not a fusion: a kernel choice) the three 32-channel stem convolutions on the patch-in-LDS kernel instead of the implicit GEMM
    //   bit 8  (with bit 5) the first convolution reads the caller's pixels itself (fp32 planes or uint8 pages): the input-layout launch and its
    //          8-channel copy of the page are gone; same conversions, same bits
    //   bit 9  the stem's residual block (two 32 -> 32 3x3 convolutions, x + conv2(hswish(conv1(x)))) in one kernel: the tensor between them lives in
    //          LDS (det_fused.h stem_res_kernel); the two launches' K orders and rounding points: bit-identical
    //   bit 7  (with bit 2) the folded head entirely on the matrix cores: the three bilinear up-samplings as a constant K = 96 map on z0's accumulators
    //          (det_head.h); re-associates fp32 sums, not bit-identical to the op list
    //   bit 6  MBConv's expand 1x1 + depthwise 3x3 + projection 1x1 in one kernel (det_mbconv.h: the expanded tensor exists per 64-channel chunk, in LDS
    //          only); takes the stride-2 transitions, where bit 3 alone leaves the 2048- / 6144-channel tensor written and read back once
    enum { FUSE_MLA_AGG = 1, FUSE_MLA_ATTN = 2, FUSE_HEAD_Z0 = 4, FUSE_DWPROJ = 8, FUSE_FMB = 16, FUSE_STEM = 32, FUSE_MBCONV = 64, FUSE_HEAD_MFMA = 128, FUSE_INPUT = 256, FUSE_STEM_RES = 512 };
    std::vector<int> fuse_kind;        // per op: the fused form that STARTS here (0 = none)
    std::vector<int> fuse_with;        // per op: index of the partner op (the one skipped / the producer folded in), -1 = none
    int input_fold = -1;               // the SA_DET_INPUT op whose only reader is the first stem convolution (that convolution is op input_fold + 1)
    std::vector<char> mb_start;        // per op: an expand 1x1 whose depthwise (op + 1) and projection (op + 2) the MBConv kernel takes with it
    std::vector<T*> mb_w2f;            // per DEPTHWISE op of a depthwise + projection pair: the projection weight, fragment-major (det_mbconv.h), made once at init; owned
    std::vector<T*> head_a0f;          // per op (FUSE_HEAD_Z0): the folded z0 convolution's weight A0, fragment-major; owned

    // an op that does not run under the fused forms switched on by `fuse`
    bool folded(int oi, int fuse) const {
        const int n = (int)ops.size();
        if (oi == input_fold && (fuse & FUSE_INPUT) && (fuse & FUSE_STEM)) return true;
        if ((fuse & FUSE_MBCONV) && ((oi >= 1 && mb_start[oi - 1]) || (oi >= 2 && mb_start[oi - 2]))) return true;
        if (oi > 0 && fuse_with[oi - 1] == oi && (fuse & fuse_kind[oi - 1])) return true;
        for (int j = oi + 1; j < n; ++j) if (fuse_kind[j] == FUSE_HEAD_Z0 && fuse_with[j] == oi && (fuse & FUSE_HEAD_Z0)) return true;
        return false;
    }

    void find_fusions() {
        const int n = (int)ops.size();
        fuse_kind.assign(n, 0); fuse_with.assign(n, -1); mb_start.assign(n, 0);
        constexpr bool BF = std::is_same<T, bf16_t>::value;
        for (int i = 0; i < n; ++i) {
            const surya_det_op& a = ops[i];
            if (a.type == SA_DET_LITEMLA && a.p0 == 32) fuse_kind[i] = FUSE_MLA_ATTN;
            if (!BF || i + 1 >= n) continue;
            const surya_det_op& b = ops[i + 1];
            if (a.type == SA_DET_DWCONV && a.k == 5 && a.stride == 1 && a.b_idx < 0 && a.act == SA_ACT_NONE && b.type == SA_DET_GROUPED1X1 &&
                b.in0 == a.out && b.p0 == 32 && a.cin % 32 == 0) { fuse_kind[i] = FUSE_MLA_AGG; fuse_with[i] = i + 1; }
            if (a.type == SA_DET_DWCONV && a.k == 3 && (a.stride == 1 || a.stride == 2) && b.type == SA_DET_CONV && b.k == 1 && b.stride == 1 &&
                b.in0 == a.out && b.act == SA_ACT_NONE && b.p1 == b.cin && a.cin % 128 == 0 && (b.cout == 256 || b.cout == 512) && b.b_idx >= 0 &&
                a.b_idx >= 0 && a.act == SA_ACT_HSWISH) { fuse_kind[i] = FUSE_DWPROJ; fuse_with[i] = i + 1; }
            if (a.type == SA_DET_CONV && a.k == 3 && a.act == SA_ACT_HSWISH && a.res < 0 && b.type == SA_DET_CONV && b.k == 1 && b.stride == 1 &&
                b.in0 == a.out && b.act == SA_ACT_NONE && b.p1 == b.cin && fmb_supported(a, b)) { fuse_kind[i] = FUSE_FMB; fuse_with[i] = i + 1; }
            if (a.type == SA_DET_CONV && a.k == 3 && a.stride == 1 && a.p0 == 1 && a.cin == 32 && a.cout == 32 && a.act == SA_ACT_HSWISH && a.res < 0 &&
                a.b_idx >= 0 && b.type == SA_DET_CONV && b.k == 3 && b.stride == 1 && b.p0 == 1 && b.cin == 32 && b.cout == 32 && b.act == SA_ACT_NONE &&
                b.b_idx >= 0 && b.in0 == a.out && b.res == a.in0 && b.p1 == a.p1 && (long)a.hin * a.win * 64 < (1L << 31) &&
                (long)max_batch * cdiv(a.win, 32) * cdiv(a.hin, 8) < (1L << 31)) {
                bool only_reader = true;
                for (int k2 = 0; k2 < n; ++k2)
                    if (k2 != i + 1 && (ops[k2].in0 == a.out || ops[k2].in1 == a.out || ops[k2].res == a.out)) only_reader = false;
                if (only_reader) { fuse_kind[i] = FUSE_STEM_RES; fuse_with[i] = i + 1; }
            }
        }
        // the input-layout op feeding ONLY the first stem convolution (3 -> 8 padded channels, 3x3 stride 2, 32 outputs)
        input_fold = -1;
        if (BF && n >= 2 && ops[0].type == SA_DET_INPUT && ops[1].type == SA_DET_CONV && ops[1].in0 == ops[0].out && ops[0].cin == 3 && ops[0].cout == 8 &&
            ops[1].cin == 8 && ops[1].cout == 32 && ops[1].k == 3 && ops[1].stride == 2 && ops[1].p0 == 1 && ops[1].act == SA_ACT_HSWISH && ops[1].res < 0 &&
            ops[1].b_idx >= 0) {
            bool only = true;
            for (int k2 = 2; k2 < n; ++k2) if (ops[k2].in0 == ops[0].out || ops[k2].in1 == ops[0].out || ops[k2].res == ops[0].out) only = false;
            if (only) input_fold = 0;
        }
        // whole MBConv blocks: expand 1x1 (+ Hardswish) whose only reader is the depthwise of a depthwise + projection pair found above
        for (int i = 0; i + 2 < n && BF; ++i) {
            const surya_det_op& a = ops[i];
            const surya_det_op& dw = ops[i + 1];
            const surya_det_op& pj = ops[i + 2];
            if (fuse_kind[i + 1] != FUSE_DWPROJ || a.type != SA_DET_CONV || a.k != 1 || a.stride != 1 || a.act != SA_ACT_HSWISH || a.res >= 0 ||
                a.p1 != a.cin || a.b_idx < 0 || dw.in0 != a.out || !mbconv_shape_ok(a.cin, a.cout, pj.cout, dw.stride)) continue;
            bool only_reader = true;
            for (int k = 0; k < n; ++k) if (k != i + 1 && (ops[k].in0 == a.out || ops[k].in1 == a.out || ops[k].res == a.out)) only_reader = false;
            if (only_reader) mb_start[i] = 1;
        }
        // the folded head: UPSUM_CLASSIFY whose full-resolution operand comes from a plain 1x1 convolution
        for (int j = 0; j < n && BF; ++j) {
            if (ops[j].type != SA_DET_UPSUM_CLASSIFY) continue;
            int srcs = 0, r_ok = 1, prod = -1;
            for (int i = j - 1; i >= 0 && ops[i].type == SA_DET_UPSUM_SRC; --i) ++srcs;
            for (int k = 0; k < srcs; ++k) {                // the addends are declared finest first: addend k is 2 << k times coarser
                const surya_det_op& u = ops[j - srcs + k];
                r_ok &= u.hin * (2 << k) == ops[j].hin && u.win * (2 << k) == ops[j].win;
            }
            for (int i = 0; i < j; ++i) if (ops[i].type == SA_DET_CONV && ops[i].out == ops[j].in0) prod = i;
            if (prod < 0 || srcs != 3 || !r_ok) continue;
            const surya_det_op& c = ops[prod];
            bool only_reader = true;
            for (int i = 0; i < n; ++i) if (i != j && (ops[i].in0 == c.out || ops[i].in1 == c.out || ops[i].res == c.out)) only_reader = false;
            if (c.k == 1 && c.stride == 1 && c.act == SA_ACT_NONE && c.res < 0 && c.p1 == c.cin && c.cin == 64 && c.cout % 128 == 0 && c.cout <= 1024 &&
                ops[j].cout <= 2 && ops[j].hin % 8 == 0 && ops[j].win % 8 == 0 && only_reader) { fuse_kind[j] = FUSE_HEAD_Z0; fuse_with[j] = prod; }
        }
    }
    int prepare_fused_weights() {
        mb_w2f.assign(ops.size(), nullptr);
        head_a0f.assign(ops.size(), nullptr);
        if constexpr (std::is_same<T, bf16_t>::value) {
            for (size_t j = 0; j < ops.size(); ++j) {
                if (fuse_kind[j] != FUSE_HEAD_Z0) continue;
                const surya_det_op& zc = ops[fuse_with[j]];
                SA_HIP(hipMalloc((void**)&head_a0f[j], (size_t)zc.cout * zc.cin * sizeof(T)));
                int rc = mbconv_w2_fragments(WT(zc.w_idx), head_a0f[j], zc.cout, zc.cin, nullptr);
                if (rc) return rc;
            }
            // every depthwise + projection pair (dwproj_kernel and mbconv_kernel read the projection weight as MFMA fragments straight from L2):
            // indexed by the DEPTHWISE op
            for (size_t i = 0; i + 1 < ops.size(); ++i) {
                if (fuse_kind[i] != FUSE_DWPROJ && fuse_kind[i] != FUSE_FMB) continue;      // (FusedMBConv: indexed by the 3x3 op)
                const surya_det_op& pj = ops[i + 1];
                SA_HIP(hipMalloc((void**)&mb_w2f[i], (size_t)pj.cout * pj.cin * sizeof(T)));
                int rc = mbconv_w2_fragments(WT(pj.w_idx), mb_w2f[i], pj.cout, pj.cin, nullptr);
                if (rc) return rc;
            }
            SA_HIP(hipDeviceSynchronize());
        }
        return SA_OK;
    }
    static bool fmb_supported(const surya_det_op& a, const surya_det_op& b) { return fmb_shape_ok(a.cin, a.cout, b.cout, a.stride, a.hout, a.wout); }

    int forward(const float* pixels, const unsigned char* pixels_u8, const float* ms, int B, float* heat, float* lowres,
                hipStream_t s, int pix = 3, float* op_ms = nullptr) override {
        if (B <= 0 || B > max_batch) return SA_ERR_ARG;
        int rc;
        const int fuse = tuning().det_fuse;
        const int n = (int)ops.size();
        if (op_ms && (int)evs.size() < n + 1) {
            const size_t have = evs.size();
            evs.resize(n + 1);
            for (size_t i = have; i < evs.size(); ++i) SA_HIP(hipEventCreate(&evs[i]));
        }
        UpsumSrc upsum{};                                   // low-resolution addends declared for the next SA_DET_UPSUM_CLASSIFY
        for (int oi = 0; oi < n; ++oi) {
            const surya_det_op& op = ops[oi];
            if (op_ms) SA_HIP(hipEventRecord(evs[oi], s));
            // an op folded into a fused form that is switched on does not run
            if (folded(oi, fuse)) continue;
            const int fk = fuse_kind[oi] & fuse;
            switch (op.type) {
                case SA_DET_UPSUM_SRC: {
                    if (upsum.n >= 3) return SA_ERR_UNSUPPORTED;
                    upsum.p[upsum.n] = bufs[op.in0]; upsum.h[upsum.n] = op.hin; upsum.w[upsum.n] = op.win;
                    ++upsum.n;
                    break;
                }
                case SA_DET_UPSUM_CLASSIFY: {
                    const long HWl = (long)op.hin * op.win, P = (long)B * HWl;
                    if (op.cin % Ty<T>::V16 || op.cout > 4) return SA_ERR_SHAPE;
                    if constexpr (std::is_same<T, bf16_t>::value) {
                        if (fk == FUSE_HEAD_Z0 && upsum.n == 3) {
                            const surya_det_op& zc = ops[fuse_with[oi]];
                            if ((fuse & FUSE_HEAD_MFMA) && head_mfma_shape_ok(op.hin, op.win, zc.cin, op.cin, op.cout)) {
                                if ((rc = launch_head_mfma(bufs[zc.in0], head_a0f[oi], WT(zc.b_idx), reinterpret_cast<const T*>(upsum.p[0]),
                                                           reinterpret_cast<const T*>(upsum.p[1]), reinterpret_cast<const T*>(upsum.p[2]), WT(op.w_idx),
                                                           WT(op.b_idx), planes, B, op.hin, op.win, zc.cin, op.cin, op.cout, s))) return rc;
                                upsum.n = 0;
                                if (lowres)
                                    SA_HIP(hipMemcpyAsync(lowres, planes, (size_t)P * op.cout * sizeof(float), hipMemcpyDeviceToDevice, s));
                                break;
                            }
                            if ((rc = launch_head_z0(bufs[zc.in0], head_a0f[oi], WT(zc.b_idx), reinterpret_cast<const T*>(upsum.p[0]),
                                                     reinterpret_cast<const T*>(upsum.p[1]), reinterpret_cast<const T*>(upsum.p[2]), WT(op.w_idx),
                                                     WT(op.b_idx), planes, B, op.hin, op.win, zc.cin, op.cin, op.cout, s))) return rc;
                            upsum.n = 0;
                            if (lowres)
                                SA_HIP(hipMemcpyAsync(lowres, planes, (size_t)P * op.cout * sizeof(float), hipMemcpyDeviceToDevice, s));
                            break;
                        }
                    }
                    // the register-blocked kernel takes addends exactly 2 / 4 / 8 times coarser (every shipped configuration: the stage
                    // strides are 2) and <= 2 labels; anything else runs the per-pixel kernel
                    bool blk = tuning().det_head_blk && upsum.n == 3 && op.cout <= 2 && op.hin % 8 == 0 && op.win % 8 == 0;
                    for (int i = 0; i < 3 && blk; ++i)
                        blk = upsum.h[i] * (2 << i) == op.hin && upsum.w[i] * (2 << i) == op.win;
                    if (blk && tuning().det_head_blk == 2) {     // 4 x 1 blocks
                        const long nblk = (long)B * op.hin * (op.win / 4);
                        hipLaunchKernelGGL((head_upsum_classify_blk_kernel<T, 2, 4, 8, 1>), dim3((unsigned)cdivl(nblk, 16)), dim3(256), 0, s,
                                           bufs[op.in0], reinterpret_cast<const T*>(upsum.p[0]), reinterpret_cast<const T*>(upsum.p[1]),
                                           reinterpret_cast<const T*>(upsum.p[2]), WT(op.w_idx), WT(op.b_idx), planes, B, op.hin, op.win,
                                           op.cin, op.cout);
                    } else if (blk) {                            // 4 x 2 blocks
                        const long nblk = (long)B * (op.hin / 2) * (op.win / 4);
                        hipLaunchKernelGGL((head_upsum_classify_blk_kernel<T, 2, 4, 8, 2>), dim3((unsigned)cdivl(nblk, 16)), dim3(256), 0, s,
                                           bufs[op.in0], reinterpret_cast<const T*>(upsum.p[0]), reinterpret_cast<const T*>(upsum.p[1]),
                                           reinterpret_cast<const T*>(upsum.p[2]), WT(op.w_idx), WT(op.b_idx), planes, B, op.hin, op.win,
                                           op.cin, op.cout);
                    } else {
                        hipLaunchKernelGGL(head_upsum_classify_kernel<T>, dim3((unsigned)cdivl(P, 16)), dim3(256), 0, s, bufs[op.in0], upsum,
                                           WT(op.w_idx), WT(op.b_idx), planes, P, op.hin, op.win, op.cin, op.cout);
                    }
                    upsum.n = 0;
                    if (lowres)
                        SA_HIP(hipMemcpyAsync(lowres, planes, (size_t)P * op.cout * sizeof(float), hipMemcpyDeviceToDevice, s));
                    break;
                }
                case SA_DET_INPUT: {
                    const long P = (long)B * op.hin * op.win;
                    if (pixels_u8) {
                        if (op.cin != 3) return SA_ERR_SHAPE;
                        hipLaunchKernelGGL(u8_to_nhwc_kernel<T>, dim3((unsigned)cdivl(P, 256)), dim3(256), 0, s, pixels_u8, bufs[op.out], P,
                                           op.cout, ms[0], ms[1], ms[2], ms[3], ms[4], ms[5], pix);
                    } else {
                        hipLaunchKernelGGL(nchw_to_nhwc_kernel<T>, dim3((unsigned)cdivl(P, 256)), dim3(256), 0, s, pixels, bufs[op.out], B,
                                           op.cin, op.hin, op.win, op.cout);
                    }
                    break;
                }
                case SA_DET_CONV: {
                    if constexpr (std::is_same<T, bf16_t>::value) {
                        if (mb_start[oi] && (fuse & FUSE_MBCONV)) {
                            const surya_det_op& dw = ops[oi + 1];
                            const surya_det_op& pj = ops[oi + 2];
                            if ((rc = launch_mbconv(bufs[op.in0], WT(op.w_idx), WT(op.b_idx), WT(dw.w_idx), WT(dw.b_idx), mb_w2f[oi + 1], WT(pj.b_idx),
                                                    pj.res >= 0 ? bufs[pj.res] : nullptr, bufs[pj.out], B, op.hin, op.win, op.cin, op.cout, dw.hout,
                                                    dw.wout, pj.cout, dw.stride, s))) return rc;
                            break;
                        }
                        if (fk == FUSE_STEM_RES) {
                            const surya_det_op& c2 = ops[fuse_with[oi]];
                            if ((rc = launch_stem_res(bufs[op.in0], WT(op.w_idx), WT(op.b_idx), WT(c2.w_idx), WT(c2.b_idx), bufs[c2.out], B, op.hin, op.win,
                                                      op.p1, s))) return rc;
                            break;
                        }
                        if (fk == FUSE_FMB) {
                            const surya_det_op& pj = ops[fuse_with[oi]];
                            if ((rc = launch_fmb(bufs[op.in0], WT(op.w_idx), WT(op.b_idx), mb_w2f[oi], WT(pj.b_idx), pj.res >= 0 ? bufs[pj.res] : nullptr,
                                                 bufs[pj.out], zero_page, B, op.hin, op.win, op.cin, op.hout, op.wout, op.cout, pj.cout, op.stride, op.p0,
                                                 op.p1, s))) return rc;
                            break;
                        }
                    }
                    if constexpr (std::is_same<T, bf16_t>::value) {
                        if (input_fold >= 0 && oi == input_fold + 1 && (fuse & FUSE_INPUT) && (fuse & FUSE_STEM)) {
                            StemSrc src{pixels_u8 ? nullptr : pixels, pixels_u8, 0.f, 0.f, 0.f, 1.f, 1.f, 1.f, pix};
                            if (pixels_u8) { src.m0 = ms[0]; src.m1 = ms[1]; src.m2 = ms[2]; src.s0 = ms[3]; src.s1 = ms[4]; src.s2 = ms[5]; }
                            if ((rc = launch_stem_conv_pixels(src, WT(op.w_idx), WT(op.b_idx), bufs[op.out], B, op.hin, op.win, op.hout, op.wout, op.cout,
                                                              op.k, op.stride, op.p0, op.p1, op.act, s))) return rc;
                            break;
                        }
                        // the 32-channel stem convolutions: patch-in-LDS kernel (det_fuse bit 5; the implicit-GEMM path below is the checker)
                        if ((fuse & FUSE_STEM) && op.cout == 32 && op.k == 3) {
                            rc = launch_stem_conv(bufs[op.in0], WT(op.w_idx), WT(op.b_idx), op.res >= 0 ? bufs[op.res] : nullptr, bufs[op.out], B, op.hin,
                                                  op.win, op.cin, op.hout, op.wout, op.cout, op.k, op.stride, op.p0, op.p1, op.act, s);
                            if (rc == SA_OK) break;
                            if (rc != SA_ERR_UNSUPPORTED) return rc;
                        }
                    }
                    if (op.k == 1 && op.stride == 1 && op.cin % Ty<T>::KE == 0 && op.p1 == op.cin) {
                        // 1x1 convolution on NHWC == plain NT GEMM over the B*H*W pixel rows: no gather arithmetic, XCD-aware
                        // tile order (55 % of the network's FLOPs go this way)
                        GemmArgs<T, T> g{bufs[op.in0], op.cin, WT(op.w_idx), op.cin, bufs[op.out], op.cout, WT(op.b_idx),
                                         op.res >= 0 ? bufs[op.res] : nullptr, op.cout, B * op.hin * op.win, op.cout, op.cin};
                        if (op.res >= 0) rc = launch_gemm<T, T, EPI_RESIDUAL>(g, s);
                        else if (op.act == SA_ACT_HSWISH) rc = launch_gemm<T, T, EPI_HARDSWISH>(g, s);
                        else if (op.act == SA_ACT_RELU) rc = launch_gemm<T, T, EPI_RELU>(g, s);
                        else rc = launch_gemm<T, T, EPI_BIAS>(g, s);
                        if (rc) return rc;
                        break;
                    }
                    ConvArgs<T> a{bufs[op.in0], WT(op.w_idx), bufs[op.out], WT(op.b_idx), op.res >= 0 ? bufs[op.res] : nullptr,
                                  B, op.hin, op.win, op.cin, op.hout, op.wout, op.cout, op.k, op.k, op.stride, op.p0, op.p1, op.act, zero_page};
                    if ((rc = launch_conv<T>(a, s))) return rc;
                    break;
                }
                case SA_DET_DWCONV: {
                    if constexpr (std::is_same<T, bf16_t>::value) {
                        if (fk == FUSE_MLA_AGG) {
                            const surya_det_op& gp = ops[fuse_with[oi]];
                            if ((rc = launch_dw5_g1x1(bufs[op.in0], WT(op.w_idx), WT(gp.w_idx), bufs[gp.out], zero_page, B, op.hin, op.win, op.cin, s))) return rc;
                            break;
                        }
                        if (fk == FUSE_DWPROJ) {
                            const surya_det_op& pj = ops[fuse_with[oi]];
                            if ((rc = launch_dwproj(bufs[op.in0], WT(op.w_idx), WT(op.b_idx), op.act, mb_w2f[oi], WT(pj.b_idx),
                                                    pj.res >= 0 ? bufs[pj.res] : nullptr, bufs[pj.out], B, op.hin, op.win, op.cin, op.hout, op.wout,
                                                    pj.cout, op.stride, s))) return rc;
                            break;
                        }
                    }
                    constexpr int TX = 4;
                    const long nt = (long)B * op.hout * cdiv(op.wout, TX) * (op.cin / Ty<T>::V16);
#define SA_DWTX(KK, SS)                                                                                                     \
    do {                                                                                                                    \
        const FastDiv fcv_ = make_fastdiv((unsigned)(op.cin / Ty<T>::V16)), fwx_ = make_fastdiv((unsigned)cdiv(op.wout, TX)), \
                      fho_ = make_fastdiv((unsigned)op.hout);                                                               \
        const int v_ = nt < (1L << 31) ? tuning().dwconv_pipe : 0;                                                          \
        if (v_ == 1)                                                                                                        \
            hipLaunchKernelGGL((dwconv_pipe_kernel<T, KK, SS, TX>), dim3((unsigned)cdivl(nt, 256)), dim3(256), 0, s, bufs[op.in0], WT(op.w_idx), \
                               WT(op.b_idx), bufs[op.out], B, op.hin, op.win, op.cin, op.hout, op.wout, op.p0, op.act, fcv_, fwx_, fho_); \
        else if (v_ == 2)                                                                                                   \
            hipLaunchKernelGGL((dwconv_tx_kernel<T, KK, SS, TX, true>), dim3((unsigned)cdivl(nt, 256)), dim3(256), 0, s, bufs[op.in0], WT(op.w_idx), \
                               WT(op.b_idx), bufs[op.out], B, op.hin, op.win, op.cin, op.hout, op.wout, op.p0, op.act, fcv_, fwx_, fho_); \
        else                                                                                                                \
            hipLaunchKernelGGL((dwconv_tx_kernel<T, KK, SS, TX, false>), dim3((unsigned)cdivl(nt, 256)), dim3(256), 0, s, bufs[op.in0], WT(op.w_idx), \
                               WT(op.b_idx), bufs[op.out], B, op.hin, op.win, op.cin, op.hout, op.wout, op.p0, op.act, fcv_, fwx_, fho_); \
    } while (0)
                    if (op.k == 3 && op.stride == 1) SA_DWTX(3, 1);
                    else if (op.k == 3 && op.stride == 2) SA_DWTX(3, 2);
                    else if (op.k == 5 && op.stride == 1) SA_DWTX(5, 1);
                    else if (op.k == 5 && op.stride == 2) SA_DWTX(5, 2);
                    else return SA_ERR_UNSUPPORTED;         // EfficientViT-L uses 3x3 (s1, s2) and 5x5 (s1) depthwise convs only
#undef SA_DWTX
                    break;
                }
                case SA_DET_GROUPED1X1: {
                    const long P = (long)B * op.hin * op.win;
                    hipLaunchKernelGGL(grouped1x1_kernel<T>, dim3((unsigned)cdivl(P, 64), op.cin / op.p0), dim3(256), 0, s, bufs[op.in0],
                                       WT(op.w_idx), bufs[op.out], P, op.cin, op.p0);
                    break;
                }
                case SA_DET_LITEMLA: {
                    const int HW = op.hin * op.win, heads = op.cout / op.p0, heads_a = heads / 2;
                    dim3 grid(B, heads, cdiv(HW, LITEMLA_CHUNK));
                    if (fk == FUSE_MLA_ATTN) {
                        hipLaunchKernelGGL((litemla_fused_kernel<T, 32>), dim3(heads, B), dim3(256), 0, s, bufs[op.in0], bufs[op.in1], bufs[op.out],
                                           HW, heads_a, heads, 1e-5f);
                    } else if (op.p0 == 32) {
                        hipLaunchKernelGGL((litemla_kv_kernel<T, 32>), grid, dim3(256), 0, s, bufs[op.in0], bufs[op.in1], kvp, HW, heads_a, heads);
                        hipLaunchKernelGGL((litemla_out_kernel<T, 32>), grid, dim3(256), 0, s, bufs[op.in0], bufs[op.in1], kvp, bufs[op.out],
                                           HW, heads_a, heads, 1e-5f);
                    } else if (op.p0 == 16) {
                        hipLaunchKernelGGL((litemla_kv_kernel<T, 16>), grid, dim3(256), 0, s, bufs[op.in0], bufs[op.in1], kvp, HW, heads_a, heads);
                        hipLaunchKernelGGL((litemla_out_kernel<T, 16>), grid, dim3(256), 0, s, bufs[op.in0], bufs[op.in1], kvp, bufs[op.out],
                                           HW, heads_a, heads, 1e-5f);
                    } else {
                        return SA_ERR_UNSUPPORTED;
                    }
                    break;
                }
                case SA_DET_UPCAT: {
                    const long n = (long)B * op.hout * op.wout * (op.cin / Ty<T>::V16);
                    hipLaunchKernelGGL(upsample_concat_kernel<T>, dim3((unsigned)cdivl(n, 256)), dim3(256), 0, s, bufs[op.in0],
                                       bufs[op.out], B, op.hin, op.win, op.cin, op.hout, op.wout, op.cout, op.p0);
                    break;
                }
                case SA_DET_CLASSIFY: {
                    const long HWl = (long)op.hin * op.win, P = (long)B * HWl;
                    hipLaunchKernelGGL(classify_sigmoid_kernel<T>, dim3((unsigned)cdivl(P, 16)), dim3(256), 0, s, bufs[op.in0],
                                       WT(op.w_idx), WT(op.b_idx), planes, P, HWl, op.cin, op.cout);
                    if (lowres)
                        SA_HIP(hipMemcpyAsync(lowres, planes, (size_t)P * op.cout * sizeof(float), hipMemcpyDeviceToDevice, s));
                    break;
                }
                case SA_DET_UPSAMPLE_OUT: {
                    if (!heat) break;
                    const long n = (long)B * op.cout * op.hout * op.wout;
                    if (op.hout == 4 * op.hin && op.wout == 4 * op.win && B * op.cout <= 65535 && cdiv(op.hin, 4) <= 65535 && tuning().det_up4)
                        hipLaunchKernelGGL(upsample_planes_x4_kernel, dim3((unsigned)cdiv(op.win, 64), (unsigned)cdiv(op.hin, 4), (unsigned)(B * op.cout)),
                                           dim3(64, 4), 0, s, planes, heat, op.hin, op.win);
                    else if (op.wout % 4 == 0)
                        hipLaunchKernelGGL(upsample_planes4_kernel, dim3((unsigned)cdivl(n / 4, 256)), dim3(256), 0, s, planes, heat,
                                           B * op.cout, op.hin, op.win, op.hout, op.wout);
                    else
                    hipLaunchKernelGGL(upsample_planes_kernel, dim3((unsigned)cdivl(n, 256)), dim3(256), 0, s, planes, heat,
                                       B * op.cout, op.hin, op.win, op.hout, op.wout);
                    break;
                }
                default: return SA_ERR_UNSUPPORTED;
            }
            if ((rc = (int)hipGetLastError())) return rc;
        }
        if (op_ms) {
            // per-op event times: op i = [event i, the next recorded event); ops that did not run (folded into a fused form) report 0
            SA_HIP(hipEventRecord(evs[n], s));
            SA_HIP(hipEventSynchronize(evs[n]));
            std::vector<int> ran;
            for (int oi = 0; oi < n; ++oi) {
                op_ms[oi] = 0.f;
                if (!folded(oi, fuse)) ran.push_back(oi);
            }
            for (size_t k = 0; k < ran.size(); ++k) {
                float ms_ = 0.f;
                SA_HIP(hipEventElapsedTime(&ms_, evs[ran[k]], evs[k + 1 < ran.size() ? ran[k + 1] : n]));
                op_ms[ran[k]] = ms_;
            }
        }
        return SA_OK;
    }
};

}  // namespace sa

using namespace sa;
struct surya_det { std::unique_ptr<DetBase> impl; };

extern "C" {

int surya_det_create(const surya_det_config* cfg, const surya_det_op* ops, const void* const* weights, int n_weights,
                     const size_t* buf_elems, int n_bufs, surya_det** out) {
    if (!cfg || !ops || !weights || !buf_elems || !out || cfg->n_ops <= 0 || n_bufs <= 0 || cfg->max_batch <= 0) return SA_ERR_ARG;
    if (cfg->height % 32 || cfg->width % 32 || cfg->num_labels < 1 || cfg->num_labels > 4) return SA_ERR_SHAPE;
    for (int i = 0; i < cfg->n_ops; ++i) {
        const surya_det_op& o = ops[i];
        if (o.out >= n_bufs || o.in0 >= n_bufs || o.in1 >= n_bufs || o.res >= n_bufs || o.w_idx >= n_weights || o.b_idx >= n_weights)
            return SA_ERR_ARG;
    }
    auto* h = new surya_det();
    int rc;
    if (cfg->dtype == SA_DTYPE_F32) {
        auto m = std::make_unique<DetModel<float>>();
        rc = m->init(*cfg, ops, weights, n_weights, buf_elems, n_bufs);
        h->impl = std::move(m);
    } else if (cfg->dtype == SA_DTYPE_BF16) {
        auto m = std::make_unique<DetModel<bf16_t>>();
        rc = m->init(*cfg, ops, weights, n_weights, buf_elems, n_bufs);
        h->impl = std::move(m);
    } else {
        rc = SA_ERR_UNSUPPORTED;
    }
    if (rc) { delete h; return rc; }
    *out = h;
    return SA_OK;
}

int surya_det_destroy(surya_det* h) {
    if (!h) return SA_ERR_ARG;
    (void)hipDeviceSynchronize();
    delete h;
    return SA_OK;
}

size_t surya_det_boxes_workspace_bytes(int batch, int height, int width, int max_boxes) {
    if (batch <= 0 || height <= 0 || width <= 0 || max_boxes <= 0) return 0;
    return sa::post::post_layout(batch, height, width, max_boxes).total;
}

int surya_det_boxes(const float* heat, long page_stride, int batch, int height, int width, float text_threshold, float low_text,
                    int max_boxes, float* boxes, float* conf, int32_t* count, void* workspace, size_t workspace_bytes, void* stream) {
    return sa::post::post_run(heat, page_stride, batch, height, width, text_threshold, low_text, max_boxes, boxes, conf, count,
                              workspace, workspace_bytes, (hipStream_t)stream);
}

int surya_det_forward(surya_det* h, const float* pixel_values, int batch, float* heatmaps, float* lowres, void* stream) {
    if (!h || !pixel_values || (!heatmaps && !lowres)) return SA_ERR_ARG;
    return h->impl->forward(pixel_values, nullptr, nullptr, batch, heatmaps, lowres, (hipStream_t)stream);
}

int surya_det_op_count(surya_det* h) { return h ? h->impl->n_ops() : SA_ERR_ARG; }

int surya_det_forward_timed(surya_det* h, const float* pixel_values, int batch, float* heatmaps, float* lowres, void* stream, float* op_ms,
                            int n_op_ms) {
    if (!h || !pixel_values || (!heatmaps && !lowres) || !op_ms || n_op_ms < h->impl->n_ops()) return SA_ERR_ARG;
    return h->impl->forward(pixel_values, nullptr, nullptr, batch, heatmaps, lowres, (hipStream_t)stream, 3, op_ms);
}

int surya_det_forward_u8(surya_det* h, const uint8_t* pixels_nhwc, int pixel_stride, const float* mean, const float* std, int batch,
                         float* heatmaps, float* lowres, void* stream) {
    if (!h || !pixels_nhwc || !mean || !std || (!heatmaps && !lowres) || (pixel_stride != 3 && pixel_stride != 4)) return SA_ERR_ARG;
    const float ms[6] = {mean[0], mean[1], mean[2], std[0], std[1], std[2]};
    return h->impl->forward(nullptr, pixels_nhwc, ms, batch, heatmaps, lowres, (hipStream_t)stream, pixel_stride);
}

int surya_resample_lanczos_u8(const uint8_t* src, int src_w, int src_h, int src_pix, uint8_t* dst, int dst_w, int dst_h, int dst_pix,
                              const int32_t* bounds_x, const int32_t* taps_x, int ksize_x, const int32_t* bounds_y,
                              const int32_t* taps_y, int ksize_y, uint8_t* tmp, void* stream) {
    if (!src || !dst || src_w <= 0 || src_h <= 0 || dst_w <= 0 || dst_h <= 0 || src_h > 65535 || dst_h > 65535) return SA_ERR_ARG;
    if ((src_pix != 3 && src_pix != 4) || (dst_pix != 3 && dst_pix != 4)) return SA_ERR_ARG;
    return sa::rs::resample_run(src, src_w, src_h, src_pix, dst, dst_w, dst_h, dst_pix, bounds_x, taps_x, ksize_x, bounds_y, taps_y,
                                ksize_y, tmp, (hipStream_t)stream);
}

}  // extern "C"
