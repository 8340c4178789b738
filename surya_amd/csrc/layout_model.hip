// Layout model family on MI355X: Donut-Swin window-attention encoder + ADETR decoder with cross- and
// self-attention, behind surya_layout_* of include/surya_amd.h. Replaces DonutSwinLayoutModel.forward
// (surya/layout/model/encoder.py:36-80, surya/common/donut/encoder.py) and SuryaLayoutDecoder.forward
// (surya/layout/model/decoder.py:96-131, surya/common/adetr/decoder.py) as LayoutPredictor calls them
// (surya/layout/__init__.py:95-131).
//
// Design notes (vs the PyTorch path):
//   * window partition, cyclic shift and their inverses are ONE permutation table per (stage, shift): LayerNorm writes its rows
//     straight into window order, the projection output is added back through the same table -- no roll / view / permute passes;
//   * the shift mask is computed from the window's position (only the last window row / column is cut), the relative position
//     bias is gathered once at load time into [head][64][64];
//   * cross-attention K / V of the encoder states are projected once per batch in `encode` (the reference caches them at the
//     first decoder call); the decode step's self-attention reuses the recogniser's fused split-K-combine + RoPE + KV-append +
//     attention kernel (decode_attn.h) on a [layer][image][kv_head][max_boxes][d] cache;
//   * every dense layer runs on gemm.h's MFMA tiles (patch embedding = GEMM over patch rows, merge reduction, GeGLU MLP).
#include <algorithm>
#include <cstring>
#include <map>
#include <memory>
#include <set>
#include <vector>

#include "../../include/surya_amd.h"
#include "gemm.h"
#include "kernels.h"
#include "decode_attn.h"
#include "layout_kernels.h"

namespace sa {

struct LayoutBase {
    virtual ~LayoutBase() {}
    virtual int encode(const float* pixels, int B, hipStream_t s) = 0;
    virtual int decode_step(const int32_t* boxes, int B, int pos, float* cls, float* box, hipStream_t s) = 0;
    virtual int encoder_states(void* out, int B, hipStream_t s) = 0;
    virtual int select(const int32_t* src, int n) = 0;
    virtual int prefill(const int32_t* boxes, int B, int Tn, float* cls, float* box, hipStream_t s) = 0;
    virtual int set_feedback(const surya_layout_feedback* fb, int B, hipStream_t s) = 0;
    virtual int decode_steps(const int32_t* boxes, int B, int pos0, int n_steps, int ring, hipStream_t s) = 0;
    virtual int wait_steps(int ring, int B, int n_steps, float* cls, float* box, int32_t* tok) = 0;
};

static size_t lalign(size_t v) { return (v + 255) & ~(size_t)255; }

__global__ void fill_int_kernel(int* p, int v, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

template <typename T>
struct LayoutModel : LayoutBase {
    surya_layout_config c;
    std::vector<const void*> w;
    std::vector<int> stage_base;                 // weight-table index of each stage's first entry
    int dec_base = 0;
    char* arena = nullptr;
    // encoder workspaces
    T *patch_rows, *x, *hbuf, *qkv, *att, *mlp;
    std::vector<int*> perm;                      // [stage * 2 + (shift > 0)] device permutation tables (token -> window-order row)
    std::vector<int*> pad_rows;                  // [stage * 2 + (shift > 0)] window-order rows of the zero padding (grids that are not whole windows)
    std::vector<int> pad_count;
    std::vector<int> grid_hp, grid_wp;           // per stage: the grid rounded up to whole windows
    // decoder
    T *ckv;                                      // [layer][B][Lk][2 * kvd]
    T *cvT = nullptr;                            // bf16: [layer][B][nkv][hd][Lkp] transposed cross-attention values (cross_attn_mfma_kernel)
    int Lkp = 0;
    T *kcache, *vcache;                          // [layer][B][nkv][Tmax][hd]
    T *dx, *dh, *dq, *dattn, *dres, *dmlp;
    float* part;
    float* cross_scratch;                        // [B][nq][ranges][hd + 2] partial cross-attention records
    static constexpr int MAX_PROMPT = 64;        // tokens of a decoder prompt that prefill() takes in one pass
    size_t enc_mlp_bytes = 0, enc_qkv_bytes = 0; // sizes of the encoder workspaces prefill() borrows between two encodes
    int cross_ranges = 1, cross_chunk = 0;
    int* cross_map_dev = nullptr;                // [max_batch] decoder row -> encoded image whose K / V it cross-attends
    int batch_active = 0;                        // decoder rows of the current decode (encode: = batch; select: any re-batching)
    float2* rope_cs;
    int *boxes_dev, *slots_dev, *len_dev;
    float *cls_dev, *box_dev;
    const T** tabs_dev = nullptr;                // 17 embedding table pointers
    int tokw() const { return c.family == SA_FAMILY_TABLE ? 10 : 7; }
    char* pinned = nullptr;
    int Lk = 0, enc_rows_final = 0, batch_encoded = 0;

    const T* W(int i) const { return reinterpret_cast<const T*>(w[i]); }
    int gh() const { return c.img_h / c.patch; }
    int gw() const { return c.img_w / c.patch; }
    int kvd() const { return c.dec_kv_heads * (c.dec_hidden / c.dec_heads); }
    int hd() const { return c.dec_hidden / c.dec_heads; }

    int init(const surya_layout_config& cfg, const void* const* weights, int n) {
        c = cfg;
        w.assign(weights, weights + n);
        int idx = SA_LW_GLOBALS;
        for (int s = 0; s < c.n_stages; ++s) { stage_base.push_back(idx); idx += SA_LS_COUNT + c.depths[s] * SA_LB_COUNT; }
        dec_base = idx;
        if (n != dec_base + c.dec_layers * SA_LD_COUNT) return SA_ERR_ARG;
        const size_t B = c.max_batch, rows0 = B * gh() * gw(), E = c.embed_dim;
        int fh = gh(), fw = gw();
        for (int s = 1; s < c.n_stages; ++s) { fh /= 2; fw /= 2; }
        Lk = fh * fw;
        if (Lk > c.encoder_length) return SA_ERR_SHAPE;
        const size_t He = E << (c.n_stages - 1), Hd = c.dec_hidden, I = c.dec_inter, qd = Hd, kv = kvd();
        const size_t qkv_d = qd + 2 * kv;
        size_t off = 0;
        auto take = [&](size_t bytes) { size_t o = off; off = lalign(off + bytes); return o; };
        // window-order workspaces hold the PADDED grid of a stage (rows of whole windows); rows x width shrinks by two per stage unless the
        // padding outgrows it, so take the largest stage
        size_t win_elems = 0;
        {
            size_t h2 = gh(), w2 = gw(), dim2 = E;
            for (int s = 0; s < c.n_stages; ++s) {
                const size_t hp = (h2 + c.window - 1) / c.window * c.window, wp = (w2 + c.window - 1) / c.window * c.window;
                win_elems = std::max(win_elems, B * hp * wp * dim2);
                h2 = (h2 + 1) / 2; w2 = (w2 + 1) / 2; dim2 *= 2;
            }
        }
        const size_t o_patch = take(rows0 * 64 * sizeof(T)), o_x = take(rows0 * E * sizeof(T)), o_h = take(win_elems * sizeof(T));
        const size_t o_qkv = take(win_elems * 3 * sizeof(T)), o_att = take(win_elems * sizeof(T)), o_mlp = take(rows0 * 4 * E * sizeof(T));
        enc_qkv_bytes = win_elems * 3 * sizeof(T); enc_mlp_bytes = rows0 * 4 * E * sizeof(T);
        const size_t o_ckv = take((size_t)c.dec_layers * B * Lk * 2 * kv * sizeof(T));
        Lkp = (Lk + 31) & ~31;
        const size_t o_cvt = take((size_t)c.dec_layers * B * kv * Lkp * sizeof(T));
        const size_t kv_elems = (size_t)c.dec_layers * B * c.dec_kv_heads * c.max_boxes * hd();
        const size_t o_k = take(kv_elems * sizeof(T)), o_v = take(kv_elems * sizeof(T));
        const size_t o_dx = take(B * Hd * sizeof(T)), o_dh = take(B * Hd * sizeof(T)), o_dq = take(B * qd * sizeof(T));
        const size_t o_da = take(B * qd * sizeof(T)), o_dr = take(B * Hd * sizeof(T)), o_dm = take(B * I * sizeof(T));
        const size_t o_part = take((size_t)8 * B * std::max(qkv_d, Hd) * sizeof(float));
        const size_t o_rope = take((size_t)c.max_boxes * (hd() / 2) * sizeof(float2));
        cross_chunk = (Lk + std::max(1, std::min(8, Lk / 128)) - 1) / std::max(1, std::min(8, Lk / 128));     // >= 128 keys per range
        cross_ranges = (Lk + cross_chunk - 1) / cross_chunk;
        const size_t o_cscr = take(B * c.dec_heads * cross_ranges * (hd() + 2) * sizeof(float)), o_cmap = take(B * sizeof(int));
        const size_t o_boxes = take(B * 10 * sizeof(int)), o_slots = take(B * sizeof(int)), o_len = take(B * sizeof(int));
        const size_t o_cls = take(B * c.label_count * sizeof(float)), o_box = take(B * 6 * sizeof(float));
        const size_t o_tabs = take(17 * sizeof(void*));
        // permutation tables: two per stage
        std::vector<size_t> o_perm;
        {
            int h = gh(), wd = gw();
            for (int s = 0; s < c.n_stages; ++s) {
                o_perm.push_back(take((size_t)h * wd * sizeof(int)));
                o_perm.push_back(take((size_t)h * wd * sizeof(int)));
                h /= 2; wd /= 2;
            }
        }
        SA_HIP(hipMalloc((void**)&arena, off));
        poison_arena(arena, off);
        // The self-attention caches must hold FINITE data from the start: the first decode step of a row (no cached token yet) loads
        // cache row 0 as the clamped duplicate behind its masked key columns, P = 0 there, and 0 x NaN = NaN. hipMalloc does not clear
        // memory -- on a box whose last tenant left NaN bit patterns in HBM every output of that model was NaN (gpurun r04f: three tests
        // failed on one box only; SURYA_AMD_POISON=1 reproduces it everywhere). Rows only ever get overwritten with finite values.
        SA_HIP(hipMemset(arena + o_k, 0, kv_elems * sizeof(T)));
        SA_HIP(hipMemset(arena + o_v, 0, kv_elems * sizeof(T)));
        char* b = arena;
        patch_rows = (T*)(b + o_patch); x = (T*)(b + o_x); hbuf = (T*)(b + o_h); qkv = (T*)(b + o_qkv); att = (T*)(b + o_att); mlp = (T*)(b + o_mlp);
        ckv = (T*)(b + o_ckv); cvT = (T*)(b + o_cvt); kcache = (T*)(b + o_k); vcache = (T*)(b + o_v);
        dx = (T*)(b + o_dx); dh = (T*)(b + o_dh); dq = (T*)(b + o_dq); dattn = (T*)(b + o_da); dres = (T*)(b + o_dr); dmlp = (T*)(b + o_dm);
        part = (float*)(b + o_part); rope_cs = (float2*)(b + o_rope);
        cross_scratch = (float*)(b + o_cscr); cross_map_dev = (int*)(b + o_cmap);
        boxes_dev = (int*)(b + o_boxes); slots_dev = (int*)(b + o_slots); len_dev = (int*)(b + o_len);
        cls_dev = (float*)(b + o_cls); box_dev = (float*)(b + o_box); tabs_dev = (const T**)(b + o_tabs);
        {   // window-order row of every token, per stage and shift (window_partition after F.pad to whole windows and torch.roll(-shift),
            // donut/encoder.py:588-636), and the window-order rows nothing maps to (the padding)
            int h = gh(), wd = gw();
            for (int s = 0; s < c.n_stages; ++s) {
                const int ws = c.window;
                if ((h % 2 || wd % 2) && s + 1 < c.n_stages) return SA_ERR_SHAPE;      // (the reference's sin-cos table is sized for grid >> stage)
                if (std::min(h, wd) < ws) return SA_ERR_UNSUPPORTED;       // a map below the window: the reference's [64][64] bias would not fit either
                const bool part_ok = std::min(h, wd) > ws;                 // else no shift (:551-559)
                const int hp = (h + ws - 1) / ws * ws, wp = (wd + ws - 1) / ws * ws;
                grid_hp.push_back(hp); grid_wp.push_back(wp);
                for (int sh = 0; sh < 2; ++sh) {
                    const int shift = (sh && part_ok) ? ws / 2 : 0;
                    std::vector<int> p((size_t)h * wd);
                    std::vector<char> hit((size_t)hp * wp, 0);
                    for (int y = 0; y < h; ++y)
                        for (int xx = 0; xx < wd; ++xx) {
                            const int ys = ((y - shift) % hp + hp) % hp, xs = ((xx - shift) % wp + wp) % wp;
                            const int r = ((ys / ws) * (wp / ws) + xs / ws) * ws * ws + (ys % ws) * ws + xs % ws;
                            p[(size_t)y * wd + xx] = r;
                            hit[r] = 1;
                        }
                    int* d = (int*)(b + o_perm[2 * s + sh]);
                    SA_HIP(hipMemcpy(d, p.data(), p.size() * sizeof(int), hipMemcpyHostToDevice));
                    perm.push_back(d);
                    std::vector<int> pads;
                    for (int r = 0; r < hp * wp; ++r)
                        if (!hit[r]) pads.push_back(r);
                    int* pd = nullptr;
                    if (!pads.empty()) {
                        SA_HIP(hipMalloc((void**)&pd, pads.size() * sizeof(int)));
                        SA_HIP(hipMemcpy(pd, pads.data(), pads.size() * sizeof(int), hipMemcpyHostToDevice));
                    }
                    pad_rows.push_back(pd);
                    pad_count.push_back((int)pads.size());
                }
                h /= 2; wd /= 2;
            }
        }
        {
            std::vector<int> ident(B);
            for (size_t i = 0; i < B; ++i) ident[i] = (int)i;
            SA_HIP(hipMemcpy(slots_dev, ident.data(), B * sizeof(int), hipMemcpyHostToDevice));
            const void* tabs[17];
            for (int i = 0; i < 17; ++i) tabs[i] = w[SA_LW_EMB_TABLES + i];
            SA_HIP(hipMemcpy((void*)tabs_dev, tabs, sizeof(tabs), hipMemcpyHostToDevice));
            const int half = hd() / 2, nn = c.max_boxes * half;
            hipLaunchKernelGGL(rope_table_kernel<T>, dim3(cdiv(nn, 256)), dim3(256), 0, 0, reinterpret_cast<const float*>(w[SA_LW_DEC_INVFREQ]),
                               rope_cs, c.max_boxes, half);
            SA_HIP(hipGetLastError());
        }
        SA_HIP(hipHostMalloc((void**)&pinned, B * (MAX_PROMPT * 10 * sizeof(int) + (c.label_count + 6) * sizeof(float)) + 256, hipHostMallocDefault));
        int rrc = init_rings();
        if (rrc) return rrc;
        SA_HIP(hipDeviceSynchronize());
        return SA_OK;
    }
    ~LayoutModel() override {
        free_rings();
        for (int* pd : pad_rows) if (pd) (void)hipFree(pd);
        if (arena) (void)hipFree(arena);
        if (pinned) (void)hipHostFree(pinned);
    }

    template <int EPI>
    int gemm(const T* X, long ldx, const T* Wt, long ldw, T* C, long ldc, const T* bias, const T* R, long ldr, int M, int N, int K,
             hipStream_t s) {
        GemmArgs<T, T> a{X, ldx, Wt, ldw, C, ldc, bias, R, ldr, M, N, K};
        return launch_gemm<T, T, EPI>(a, s);
    }
    int layernorm(const T* in, int wi, int bi, T* out, const int* pm, long rows, int rpi, int C, float eps, hipStream_t s, int rpi_out = 0) {
        if constexpr (std::is_same<T, bf16_t>::value) {      // rows held in registers by C / 8 lanes (layout_kernels.h); 16-byte aligned rows
            if (rows > 0 && tuning().lay_ln) {
#define SA_LN_ROWS(LPR, NV)                                                                                                              \
    {                                                                                                                                    \
        hipLaunchKernelGGL((lay::layernorm_rows_bf16_kernel<LPR, NV>), dim3((unsigned)cdivl(rows, 256 / LPR)), dim3(256), 0, s, in, W(wi), W(bi), \
                           out, pm, rows, rpi, eps, rpi_out);                                                                            \
        return (int)hipGetLastError();                                                                                                   \
    }
                if (C == 128) SA_LN_ROWS(16, 1)
                if (C == 256) SA_LN_ROWS(32, 1)
                if (C == 512) SA_LN_ROWS(64, 1)
                if (C == 1024) SA_LN_ROWS(64, 2)
#undef SA_LN_ROWS
            }
        }
        hipLaunchKernelGGL(lay::layernorm_kernel<T>, dim3((unsigned)cdivl(rows, 4)), dim3(256), 0, s, in, W(wi), W(bi), out, pm, rows, rpi, C, eps, rpi_out);
        return (int)hipGetLastError();
    }

    int encode(const float* pixels, int B, hipStream_t s) override {
        if (B <= 0 || B > c.max_batch) return SA_ERR_ARG;
        int rc;
        int h = gh(), wd = gw(), dim = c.embed_dim;
        long rows = (long)B * h * wd;
        {
            const long total = rows * 64;
            hipLaunchKernelGGL(lay::patchify_kernel<T>, dim3((unsigned)cdivl(total, 256)), dim3(256), 0, s, pixels, patch_rows, B, 3, c.img_h,
                               c.img_w, c.patch, 64);
            if ((rc = gemm<EPI_BIAS>(patch_rows, 64, W(SA_LW_PATCH_W), 64, hbuf, dim, W(SA_LW_PATCH_B), nullptr, 0, (int)rows, dim, 64, s))) return rc;
            if ((rc = layernorm(hbuf, SA_LW_EMB_LN_W, SA_LW_EMB_LN_B, x, nullptr, rows, h * wd, dim, 1e-5f, s))) return rc;
        }
        for (int st = 0; st < c.n_stages; ++st) {
            const int sb = stage_base[st], nh = c.heads[st], nkv = c.kv_heads[st], ws = c.window;
            if (dim / nh != 32) return SA_ERR_UNSUPPORTED;
            const int rpi = h * wd;
            {
                const long n4 = rows * (dim / 4);
                hipLaunchKernelGGL(lay::add_rows_kernel<T>, dim3((unsigned)cdivl(n4, 256)), dim3(256), 0, s, x, W(sb + SA_LS_SINCOS), rows, rpi, dim);
            }
            const int qkv_n = (nh + 2 * nkv) * 32;
            for (int bi = 0; bi < c.depths[st]; ++bi) {
                const int wb = sb + SA_LS_COUNT + bi * SA_LB_COUNT;
                const bool shifted = (bi % 2 == 1) && std::min(h, wd) > ws;
                const int* pm = perm[2 * st + (shifted ? 1 : 0)];
                const int hp = grid_hp[st], wp = grid_wp[st], rpw = hp * wp;                  // the grid in whole windows
                const long rows_w = (long)B * rpw;
                const int pi = 2 * st + (shifted ? 1 : 0);
                if (pad_count[pi]) {
                    const long n4 = (long)B * pad_count[pi] * (dim / 4);
                    hipLaunchKernelGGL(lay::zero_rows_kernel<T>, dim3((unsigned)cdivl(n4, 256)), dim3(256), 0, s, hbuf, pad_rows[pi], pad_count[pi], B, rpw, dim);
                }
                if ((rc = layernorm(x, wb + SA_LB_LN1_W, wb + SA_LB_LN1_B, hbuf, pm, rows, rpi, dim, c.enc_eps, s, rpw))) return rc;
                if ((rc = gemm<EPI_BIAS>(hbuf, dim, W(wb + SA_LB_QKV_W), dim, qkv, qkv_n, W(wb + SA_LB_QKV_B), nullptr, 0, (int)rows_w, qkv_n, dim, s)))
                    return rc;
                const int nwx = wp / ws, nwy = hp / ws;
                if constexpr (std::is_same<T, bf16_t>::value)
                    hipLaunchKernelGGL(lay::swin_window_attn_mfma_kernel, dim3((unsigned)(rows_w / 64), nh), dim3(128), 0, s, qkv,
                                       reinterpret_cast<const float*>(w[wb + SA_LB_RELBIAS]), att, nh, nkv, nwx, nwy, shifted ? ws / 2 : 0, ws);
                else
                    hipLaunchKernelGGL(lay::swin_window_attn_kernel<T>, dim3((unsigned)(rows_w / 64), nh), dim3(256), 0, s, qkv,
                                       reinterpret_cast<const float*>(w[wb + SA_LB_RELBIAS]), att, nh, nkv, nwx, nwy, shifted ? ws / 2 : 0, ws);
                if ((rc = gemm<EPI_BIAS>(att, dim, W(wb + SA_LB_PROJ_W), dim, hbuf, dim, W(wb + SA_LB_PROJ_B), nullptr, 0, (int)rows_w, dim, dim, s)))
                    return rc;
                {
                    const long n4 = rows * (dim / 4);
                    hipLaunchKernelGGL(lay::gather_add_kernel<T>, dim3((unsigned)cdivl(n4, 256)), dim3(256), 0, s, x, hbuf, pm, rows, rpi, dim, rpw);
                }
                if ((rc = layernorm(x, wb + SA_LB_LN2_W, wb + SA_LB_LN2_B, hbuf, nullptr, rows, rpi, dim, c.enc_eps, s))) return rc;
                if ((rc = gemm<EPI_GELU>(hbuf, dim, W(wb + SA_LB_FC1_W), dim, mlp, 4 * dim, W(wb + SA_LB_FC1_B), nullptr, 0, (int)rows, 4 * dim, dim, s)))
                    return rc;
                if ((rc = gemm<EPI_RESIDUAL>(mlp, 4 * dim, W(wb + SA_LB_FC2_W), 4 * dim, x, dim, W(wb + SA_LB_FC2_B), x, dim, (int)rows, dim, 4 * dim,
                                             s))) return rc;
            }
            if (st + 1 < c.n_stages) {
                const long orows = rows / 4;
                hipLaunchKernelGGL(lay::merge_ln_kernel<T>, dim3((unsigned)cdivl(orows, 4)), dim3(256), 0, s, x, W(sb + SA_LS_MERGE_NORM_W),
                                   W(sb + SA_LS_MERGE_NORM_B), mlp, B, h, wd, dim, 1e-5f);
                if ((rc = gemm<EPI_BIAS>(mlp, 4 * dim, W(sb + SA_LS_MERGE_RED_W), 4 * dim, x, 2 * dim, nullptr, nullptr, 0, (int)orows, 2 * dim,
                                         4 * dim, s))) return rc;
                rows = orows; h /= 2; wd /= 2; dim *= 2;
            }
        }
        {
            const long n4 = rows * (dim / 4);
            hipLaunchKernelGGL(lay::add_rows_kernel<T>, dim3((unsigned)cdivl(n4, 256)), dim3(256), 0, s, x, W(SA_LW_POS_EMB), rows, h * wd, dim);
        }
        enc_rows_final = (int)rows;
        batch_encoded = batch_active = B;
        fed_ready = false;
        SA_HIP(hipMemcpyAsync(cross_map_dev, slots_dev, (size_t)B * sizeof(int), hipMemcpyDeviceToDevice, s));     // identity
        // cross-attention keys / values of every decoder layer (adetr/decoder.py:167-173: projected once, then cached)
        const int kv2 = 2 * kvd();
        for (int l = 0; l < c.dec_layers; ++l) {
            const int lb = dec_base + l * SA_LD_COUNT;
            T* dst = ckv + (size_t)l * c.max_batch * Lk * kv2;
            if ((rc = gemm<EPI_BIAS>(x, dim, W(lb + SA_LD_CKV_W), dim, dst, kv2, nullptr, nullptr, 0, (int)rows, kv2, dim, s))) return rc;
            if constexpr (std::is_same<T, bf16_t>::value)
                hipLaunchKernelGGL(lay::transpose_cross_v_kernel<T>, dim3(64, B), dim3(256), 0, s, dst, cvT + (size_t)l * c.max_batch * kvd() * Lkp, Lk, Lkp,
                                   c.dec_kv_heads, hd());
        }
        return (int)hipGetLastError();
    }

    int encoder_states(void* out, int B, hipStream_t s) override {
        if (B != batch_encoded) return SA_ERR_STATE;
        const size_t He = (size_t)c.embed_dim << (c.n_stages - 1);
        SA_HIP(hipMemcpyAsync(out, x, (size_t)enc_rows_final * He * sizeof(T), hipMemcpyDeviceToDevice, s));
        return SA_OK;
    }

    // Decoder rows != encoded images: row i cross-attends image src[i] (table recognition's cell pass). Synchronous, tiny.
    int select(const int32_t* src, int n) override {
        if (n <= 0 || n > c.max_batch || batch_encoded <= 0) return SA_ERR_ARG;
        for (int i = 0; i < n; ++i)
            if (src[i] < 0 || src[i] >= batch_encoded) return SA_ERR_ARG;
        SA_HIP(hipDeviceSynchronize());
        SA_HIP(hipMemcpy(cross_map_dev, src, (size_t)n * sizeof(int), hipMemcpyHostToDevice));
        batch_active = n;
        fed_ready = false;
        return SA_OK;
    }

    // A decoder prompt of Tn tokens per row in ONE pass (positions 0 .. Tn - 1; the decode steps continue at Tn): GEMMs over B * Tn rows,
    // cross attention per (row, token), causal self-attention inside each row's prompt with the K / V rows written to the cache. What
    // the reference's first decoder call does with prefill = True (surya/table_rec/__init__.py:60-68). Activations live in the
    // encoder's idle MLP / qkv workspaces. Returns the heads' outputs of every row's LAST prompt token.
    int prefill(const int32_t* boxes, int B, int Tn, float* cls, float* box, hipStream_t s) override {
        if (B != batch_active) return SA_ERR_STATE;
        if (Tn < 1 || Tn > MAX_PROMPT || Tn > c.max_boxes) return SA_ERR_ARG;
        fed_ready = false;
        const int Hd = c.dec_hidden, I = c.dec_inter, nq = c.dec_heads, nkv = c.dec_kv_heads, d = hd(), kv = kvd();
        const int qkv_d = Hd + 2 * kv, rows = B * Tn, G = nq / nkv;
        if (G > 8 || (d != 64 && d != 32)) return SA_ERR_UNSUPPORTED;
        size_t off = 0;
        auto take = [&](size_t bytes) { size_t o = off; off = lalign(off + bytes); return o; };
        const size_t o_x = take((size_t)rows * Hd * sizeof(T)), o_h = take((size_t)rows * Hd * sizeof(T)), o_q = take((size_t)rows * Hd * sizeof(T));
        const size_t o_qkv = take((size_t)rows * qkv_d * sizeof(T)), o_at = take((size_t)rows * Hd * sizeof(T)), o_rs = take((size_t)rows * Hd * sizeof(T));
        const size_t o_ml = take((size_t)rows * I * sizeof(T));
        if (off > enc_mlp_bytes) return SA_ERR_ARG;                   // callers fall back to one decode step per prompt token
        char* wb = reinterpret_cast<char*>(mlp);
        T *px = (T*)(wb + o_x), *ph = (T*)(wb + o_h), *pq = (T*)(wb + o_q), *pqkv = (T*)(wb + o_qkv), *pat = (T*)(wb + o_at), *prs = (T*)(wb + o_rs),
          *pml = (T*)(wb + o_ml);
        off = 0;
        const size_t o_sc = take((size_t)rows * nq * cross_ranges * (d + 2) * sizeof(float)), o_mp = take((size_t)rows * sizeof(int)),
                     o_bx = take((size_t)rows * tokw() * sizeof(int));
        if (off > enc_qkv_bytes) return SA_ERR_ARG;
        char* qb = reinterpret_cast<char*>(qkv);
        float* pscr = (float*)(qb + o_sc);
        int *pmap = (int*)(qb + o_mp), *pbox = (int*)(qb + o_bx);
        int rc;
        int* hb = reinterpret_cast<int*>(pinned);
        memcpy(hb, boxes, (size_t)rows * tokw() * sizeof(int));
        SA_HIP(hipMemcpyAsync(pbox, hb, (size_t)rows * tokw() * sizeof(int), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(lay::expand_map_kernel, dim3(cdiv(rows, 256)), dim3(256), 0, s, cross_map_dev, pmap, B, Tn);
        if (c.family == SA_FAMILY_TABLE)
            hipLaunchKernelGGL(lay::table_embed_kernel<T>, dim3(rows), dim3(256), 0, s, pbox, tabs_dev, px, Hd, c.box_embed, c.bbox_size, c.vocab,
                               c.category_count, c.merge_count);
        else
            hipLaunchKernelGGL(lay::box_embed_kernel<T>, dim3(rows), dim3(256), 0, s, pbox, tabs_dev, px, Hd, c.bbox_size, c.vocab, c.label_count);
        const float scale = 1.0f / sqrtf((float)d);
        const size_t layer_kv = (size_t)c.max_batch * nkv * c.max_boxes * d;
        auto norm = [&](const T* in, int wi, T* outp) {
            hipLaunchKernelGGL(lay::adetr_rmsnorm_kernel<T>, dim3(cdiv(rows, 4)), dim3(256), 0, s, in, W(wi), outp, rows, Hd, c.rms_eps);
        };
        for (int l = 0; l < c.dec_layers; ++l) {
            const int lb = dec_base + l * SA_LD_COUNT;
            norm(px, lb + SA_LD_CNORM, ph);
            if ((rc = gemm<EPI_BIAS>(ph, Hd, W(lb + SA_LD_CQ_W), Hd, pq, Hd, nullptr, nullptr, 0, rows, Hd, Hd, s))) return rc;
            {
                const T* kvp = ckv + (size_t)l * c.max_batch * Lk * 2 * kv;
                const size_t lds = ((size_t)G * cross_chunk + (size_t)G * d + 1024) * sizeof(float);
                dim3 grid(rows, nkv, cross_ranges);
                const int mblocks = cdiv(rows * nq * (d / 4), 256);
                if constexpr (std::is_same<T, bf16_t>::value) {
                    const T* vtp = cvT + (size_t)l * c.max_batch * kv * Lkp;
                    if (d == 64) hipLaunchKernelGGL((lay::cross_attn_mfma_kernel<64>), dim3(rows, nkv), dim3(512), 0, s, reinterpret_cast<const float*>(pq), 0, rows,
                                                    kvp, vtp, pat, pmap, nq, nkv, Lk, Lkp, scale);
                    else hipLaunchKernelGGL((lay::cross_attn_mfma_kernel<32>), dim3(rows, nkv), dim3(512), 0, s, reinterpret_cast<const float*>(pq), 0, rows,
                                            kvp, vtp, pat, pmap, nq, nkv, Lk, Lkp, scale);
                } else if (d == 64) {
                    hipLaunchKernelGGL((lay::cross_attn_split_kernel<T, 64>), grid, dim3(256), lds, s, reinterpret_cast<const float*>(pq), 0, rows, kvp, pscr,
                                       pmap, nq, nkv, Lk, cross_chunk, scale);
                    hipLaunchKernelGGL((lay::cross_attn_merge_kernel<T, 64>), dim3(mblocks), dim3(256), 0, s, pscr, pat, rows * nq, cross_ranges);
                } else {
                    hipLaunchKernelGGL((lay::cross_attn_split_kernel<T, 32>), grid, dim3(256), lds, s, reinterpret_cast<const float*>(pq), 0, rows, kvp, pscr,
                                       pmap, nq, nkv, Lk, cross_chunk, scale);
                    hipLaunchKernelGGL((lay::cross_attn_merge_kernel<T, 32>), dim3(mblocks), dim3(256), 0, s, pscr, pat, rows * nq, cross_ranges);
                }
            }
            if ((rc = gemm<EPI_RESIDUAL>(pat, Hd, W(lb + SA_LD_CO_W), Hd, prs, Hd, W(lb + SA_LD_CO_B), px, Hd, rows, Hd, Hd, s))) return rc;
            norm(prs, lb + SA_LD_TNORM, ph);
            if ((rc = gemm<EPI_BIAS>(ph, Hd, W(lb + SA_LD_QKV_W), Hd, pqkv, qkv_d, nullptr, nullptr, 0, rows, qkv_d, Hd, s))) return rc;
            {
                T* kc = kcache + (size_t)l * layer_kv;
                T* vc = vcache + (size_t)l * layer_kv;
                const size_t lds = (size_t)2 * Tn * d * sizeof(float);
                if (d == 64) hipLaunchKernelGGL((lay::adetr_prefill_attn_kernel<T, 64>), dim3(B, nkv), dim3(256), lds, s, pqkv, pat, kc, vc, rope_cs, Tn, nq, nkv,
                                                c.max_boxes, scale);
                else hipLaunchKernelGGL((lay::adetr_prefill_attn_kernel<T, 32>), dim3(B, nkv), dim3(256), lds, s, pqkv, pat, kc, vc, rope_cs, Tn, nq, nkv,
                                        c.max_boxes, scale);
            }
            // layout: + RAW layer input (double residual flow); table_rec: + the cross-attention output
            if ((rc = gemm<EPI_RESIDUAL>(pat, Hd, W(lb + SA_LD_TO_W), Hd, prs, Hd, W(lb + SA_LD_TO_B), c.family == SA_FAMILY_TABLE ? prs : px, Hd, rows, Hd,
                                         Hd, s))) return rc;
            norm(prs, lb + SA_LD_MNORM, ph);
            if ((rc = gemm<EPI_GEGLU>(ph, Hd, W(lb + SA_LD_GU_W), Hd, pml, I, nullptr, nullptr, 0, rows, 2 * I, Hd, s))) return rc;
            if ((rc = gemm<EPI_RESIDUAL>(pml, I, W(lb + SA_LD_DOWN_W), I, px, Hd, nullptr, prs, Hd, rows, Hd, I, s))) return rc;
        }
        hipLaunchKernelGGL((lay::layout_heads_kernel<T, false>), dim3(B), dim3(256), (size_t)Hd * 4, s, px + (size_t)(Tn - 1) * Hd, W(SA_LW_DEC_FNORM),
                           W(SA_LW_DEC_LN_W), W(SA_LW_DEC_LN_B), W(SA_LW_DEC_LM_W), W(SA_LW_DEC_BB_W), W(SA_LW_DEC_BB_B), cls_dev, box_dev, Hd,
                           c.label_count, c.rms_eps, c.ln_eps, (long)Tn * Hd);
        if ((rc = (int)hipGetLastError())) return rc;
        float* hc = reinterpret_cast<float*>(pinned + 256 + (size_t)c.max_batch * MAX_PROMPT * 10 * sizeof(int));
        float* hbx = hc + (size_t)c.max_batch * c.label_count;
        SA_HIP(hipMemcpyAsync(hc, cls_dev, (size_t)B * c.label_count * sizeof(float), hipMemcpyDeviceToHost, s));
        SA_HIP(hipMemcpyAsync(hbx, box_dev, (size_t)B * 6 * sizeof(float), hipMemcpyDeviceToHost, s));
        SA_HIP(hipStreamSynchronize(s));
        memcpy(cls, hc, (size_t)B * c.label_count * sizeof(float));
        memcpy(box, hbx, (size_t)B * 6 * sizeof(float));
        return SA_OK;
    }

    // Host-fed start of a decode step: the token rows go to the device, are embedded, and the first layer's cross_pre_norm is applied
    // (dx = embedding, dh = its norm). The fused heads kernel (lay::layout_heads_kernel<T, true>) leaves the same state for the
    // step after it, so only the FIRST step of a device-fed run comes through here.
    int start_step(const int32_t* boxes, int B, int pos, hipStream_t s) {
        const int Hd = c.dec_hidden;
        int* hb = reinterpret_cast<int*>(pinned);
        memcpy(hb, boxes, (size_t)B * tokw() * sizeof(int));
        SA_HIP(hipMemcpyAsync(boxes_dev, hb, (size_t)B * tokw() * sizeof(int), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(fill_int_kernel, dim3(cdiv(B, 64)), dim3(64), 0, s, len_dev, pos, B);
        if (c.family == SA_FAMILY_TABLE)
            hipLaunchKernelGGL(lay::table_embed_kernel<T>, dim3(B), dim3(256), 0, s, boxes_dev, tabs_dev, dx, Hd, c.box_embed, c.bbox_size, c.vocab,
                               c.category_count, c.merge_count);
        else
            hipLaunchKernelGGL(lay::box_embed_kernel<T>, dim3(B), dim3(256), 0, s, boxes_dev, tabs_dev, dx, Hd, c.bbox_size, c.vocab, c.label_count);
        hipLaunchKernelGGL(lay::adetr_rmsnorm_kernel<T>, dim3(cdiv(B, 4)), dim3(256), 0, s, dx, W(dec_base + SA_LD_CNORM), dh, B, Hd, c.rms_eps);
        return (int)hipGetLastError();
    }

    // The decoder layers of one step for B rows: dx = the token embeddings, dh = cross_pre_norm(dx) of layer 0, len_dev = the cache
    // position of every row on entry; dx = the final layer's output on exit.
    int decode_layers(int B, hipStream_t s) {
        const int Hd = c.dec_hidden, I = c.dec_inter, nq = c.dec_heads, nkv = c.dec_kv_heads, d = hd(), kv = kvd();
        const int qkv_d = Hd + 2 * kv;
        int rc;
        const float scale = 1.0f / sqrtf((float)d);
        const size_t layer_kv = (size_t)c.max_batch * nkv * c.max_boxes * d;
        // Every M = B projection runs split-K (64 x 64 tiles over ~128-256 workgroups instead of 32) and hands its slabs to the
        // consumer: the cross-attention kernel sums the q slabs itself, the decode-attention kernel the qkv slabs, and
        // splitk_residual_adetr_norm_kernel folds reduce + bias + residual + the NEXT norm into one launch (r03: 1476 -> 680 us
        // per step of 32 pages, profiles/r03_g_layout_kernel_stats.md; the unsplit 64 x 32 launches ran 14 us each on 32 workgroups).
        auto splitk = [&](const T* A, int lda, const T* Wt, int N, int K, int& S) -> int {
            GemmArgs<T, T> a{A, lda, Wt, K, nullptr, 0, nullptr, nullptr, 0, B, N, K, 1, part};
            const int r = launch_gemm_splitk<T>(a, s);
            S = a.splitk;
            return r;
        };
        auto reduce_norm = [&](int S, const T* res, const T* bias, T* xo, const T* nw) {
            const int threads = std::min(1024, ((Hd / 4 + 63) / 64) * 64);
            hipLaunchKernelGGL(lay::splitk_residual_adetr_norm_kernel<T>, dim3(B), dim3(threads), 0, s, part, S, B, res, bias, xo, nw, dh, Hd,
                               c.rms_eps);
        };
        if (Hd % 4 || Hd > 4096) return SA_ERR_UNSUPPORTED;
        for (int l = 0; l < c.dec_layers; ++l) {
            const int lb = dec_base + l * SA_LD_COUNT;
            int S = 1;
            // cross attention (double residual flow, adetr/decoder.py:430-457): cross = o(attn(norm(x))) + x; dh = norm(x) on entry
            if ((rc = splitk(dh, Hd, W(lb + SA_LD_CQ_W), Hd, Hd, S))) return rc;
            {
                const T* kvp = ckv + (size_t)l * c.max_batch * Lk * 2 * kv;
                const int G = nq / nkv;
                const size_t lds = ((size_t)G * cross_chunk + (size_t)G * d + 1024) * sizeof(float);
                dim3 grid(B, nkv, cross_ranges);
                if (G > 8 || G * (d / 4) > 256) return SA_ERR_UNSUPPORTED;
                const int mblocks = cdiv(B * nq * (d / 4), 256);
                bool done = false;
                if constexpr (std::is_same<T, bf16_t>::value) {          // matrix-core kernel on the transposed values (layout_kernels.h)
                    const T* vtp = cvT + (size_t)l * c.max_batch * kv * Lkp;
                    done = true;
                    if (d == 64) hipLaunchKernelGGL((lay::cross_attn_mfma_kernel<64>), dim3(B, nkv), dim3(512), 0, s, part, S, B, kvp, vtp, dattn, cross_map_dev,
                                                    nq, nkv, Lk, Lkp, scale);
                    else if (d == 32) hipLaunchKernelGGL((lay::cross_attn_mfma_kernel<32>), dim3(B, nkv), dim3(512), 0, s, part, S, B, kvp, vtp, dattn,
                                                         cross_map_dev, nq, nkv, Lk, Lkp, scale);
                    else done = false;
                }
                if (done) {
                } else if (d == 64) {
                    hipLaunchKernelGGL((lay::cross_attn_split_kernel<T, 64>), grid, dim3(256), lds, s, part, S, B, kvp, cross_scratch, cross_map_dev, nq, nkv, Lk,
                                       cross_chunk, scale);
                    hipLaunchKernelGGL((lay::cross_attn_merge_kernel<T, 64>), dim3(mblocks), dim3(256), 0, s, cross_scratch, dattn, B * nq, cross_ranges);
                } else if (d == 32) {
                    hipLaunchKernelGGL((lay::cross_attn_split_kernel<T, 32>), grid, dim3(256), lds, s, part, S, B, kvp, cross_scratch, cross_map_dev, nq, nkv, Lk,
                                       cross_chunk, scale);
                    hipLaunchKernelGGL((lay::cross_attn_merge_kernel<T, 32>), dim3(mblocks), dim3(256), 0, s, cross_scratch, dattn, B * nq, cross_ranges);
                } else return SA_ERR_UNSUPPORTED;
            }
            if ((rc = splitk(dattn, Hd, W(lb + SA_LD_CO_W), Hd, Hd, S))) return rc;
            reduce_norm(S, dx, W(lb + SA_LD_CO_B), dres, W(lb + SA_LD_TNORM));              // dres = cross, dh = temporal_pre_norm(cross)
            // self attention on norm(cross); residual = o(attn) + RAW layer input
            {
                GemmArgs<T, T> a{dh, Hd, W(lb + SA_LD_QKV_W), Hd, nullptr, 0, nullptr, nullptr, 0, B, qkv_d, Hd, 1, part};
                if ((rc = launch_gemm_splitk<T>(a, s))) return rc;
                const int S = a.splitk, G = nq / nkv;
                dim3 grid(B, nkv), block(256);
                T* kc = kcache + (size_t)l * layer_kv;
                T* vc = vcache + (size_t)l * layer_kv;
#define SA_LAY_DEC(KERN, LDS, ...)                                                                                                  \
    {                                                                                                                               \
        auto kern = KERN;                                                                                                           \
        static AttrOnce attr;                                                                                                       \
        attr.ensure(kern, LDS);                                                                                                     \
        hipLaunchKernelGGL(kern, grid, block, LDS, s, part, S, W(SA_LW_DEC_ZERO_BIAS), dattn, kc, vc, slots_dev, len_dev, rope_cs, nq, \
                           nkv, c.max_boxes, scale, ##__VA_ARGS__);                                                                \
    }
                bool done = false;
                if constexpr (std::is_same<T, bf16_t>::value) {
                    done = true;
                    // round 4: the thread-local-prologue kernel of the recogniser's decode step (decode_attn.h, fourth version; dattn = 3 keeps the third)
                    if (tuning().dattn == 3) {
                        if (d == 64 && G <= 8) SA_LAY_DEC((decode_attn_flash_kernel<64, 8>), (decode_attn_flash_lds<64, 8>()), (uint8_t*)nullptr, (uint8_t*)nullptr, 0)
                        else if (d == 32 && G <= 8) SA_LAY_DEC((decode_attn_flash_kernel<32, 8>), (decode_attn_flash_lds<32, 8>()), (uint8_t*)nullptr, (uint8_t*)nullptr, 0)
                        else done = false;
                    } else if (d == 64 && G <= 8) SA_LAY_DEC((decode_attn_flash2_kernel<64, 8, false>), (decode_attn_flash2_lds<64, 8, false>()), (uint8_t*)nullptr, (uint8_t*)nullptr, 0)
                    else if (d == 32 && G <= 8) SA_LAY_DEC((decode_attn_flash2_kernel<32, 8, false>), (decode_attn_flash2_lds<32, 8, false>()), (uint8_t*)nullptr, (uint8_t*)nullptr, 0)
                    else done = false;
                }
                if (!done) {
                    if (d == 64 && G <= 8) SA_LAY_DEC((decode_attn_mfma_kernel<T, 64, 8>), (decode_attn_mfma_lds<T, 64, 8>()))
                    else if (d == 32 && G <= 8) SA_LAY_DEC((decode_attn_mfma_kernel<T, 32, 8>), (decode_attn_mfma_lds<T, 32, 8>()))
                    else return SA_ERR_UNSUPPORTED;
                }
#undef SA_LAY_DEC
            }
            if ((rc = splitk(dattn, Hd, W(lb + SA_LD_TO_W), Hd, Hd, S))) return rc;
            // layout: + RAW layer input (double residual flow); table_rec: + the cross-attention output (adetr/decoder.py:395-417)
            reduce_norm(S, c.family == SA_FAMILY_TABLE ? dres : dx, W(lb + SA_LD_TO_B), dres, W(lb + SA_LD_MNORM));   // dh = channel_pre_norm(residual)
            // MLP: x = down(gelu_tanh(gate(n)) * up(n)) + residual; dh <- the next layer's cross_pre_norm(x)
            if ((rc = gemm<EPI_GEGLU>(dh, Hd, W(lb + SA_LD_GU_W), Hd, dmlp, I, nullptr, nullptr, 0, B, 2 * I, Hd, s))) return rc;
            if ((rc = splitk(dmlp, I, W(lb + SA_LD_DOWN_W), Hd, I, S))) return rc;
            reduce_norm(S, dres, nullptr, dx, l + 1 < c.dec_layers ? W(lb + SA_LD_COUNT + SA_LD_CNORM) : nullptr);
        }
        return (int)hipGetLastError();
    }

    int decode_step(const int32_t* boxes, int B, int pos, float* cls, float* box, hipStream_t s) override {
        if (B != batch_active) return SA_ERR_STATE;
        if (pos < 0 || pos >= c.max_boxes) return SA_ERR_ARG;
        const int Hd = c.dec_hidden;
        int rc;
        fed_ready = false;
        if ((rc = start_step(boxes, B, pos, s))) return rc;
        if ((rc = decode_layers(B, s))) return rc;
        hipLaunchKernelGGL((lay::layout_heads_kernel<T, false>), dim3(B), dim3(256), (size_t)Hd * 4, s, dx, W(SA_LW_DEC_FNORM), W(SA_LW_DEC_LN_W),
                           W(SA_LW_DEC_LN_B), W(SA_LW_DEC_LM_W), W(SA_LW_DEC_BB_W), W(SA_LW_DEC_BB_B), cls_dev, box_dev, Hd, c.label_count,
                           c.rms_eps, c.ln_eps, (long)Hd);
        if ((rc = (int)hipGetLastError())) return rc;
        float* hc = reinterpret_cast<float*>(pinned + 256 + (size_t)c.max_batch * MAX_PROMPT * 10 * sizeof(int));
        float* hbx = hc + (size_t)c.max_batch * c.label_count;
        SA_HIP(hipMemcpyAsync(hc, cls_dev, (size_t)B * c.label_count * sizeof(float), hipMemcpyDeviceToHost, s));
        SA_HIP(hipMemcpyAsync(hbx, box_dev, (size_t)B * 6 * sizeof(float), hipMemcpyDeviceToHost, s));
        SA_HIP(hipStreamSynchronize(s));
        memcpy(cls, hc, (size_t)B * c.label_count * sizeof(float));
        memcpy(box, hbx, (size_t)B * 6 * sizeof(float));
        return SA_OK;
    }

    // ------------------------------------------------------------------------------------------------ device-fed decode steps (round 4)
    // n_steps decode steps whose fed-back tokens never leave the device (lay::layout_heads_kernel<T, true>), recorded in ring `ring`
    // ([step][row] class logits / boxes / fed tokens); one D2H copy + event at the end, no host synchronisation: the caller enqueues
    // the next run before it waits for this one. With surya_set_tuning("graph", 1) a run is replayed as a hipGraph once its (rows,
    // steps) shape has been seen twice (the cache position lives in len_dev, so it is not part of the shape); measured slower, off by default.
    static constexpr int RING_STEPS = 16;
    float *cls_ring = nullptr, *box_ring = nullptr;
    int *tok_ring = nullptr, *page_sizes_dev = nullptr;
    char* ring_host = nullptr;                   // pinned: [2] x (cls | box | tok) of RING_STEPS steps
    size_t ring_cls = 0, ring_box = 0, ring_tok = 0;     // elements per ring
    hipEvent_t ev_ring[2] = {nullptr, nullptr}, gev_in = nullptr, gev_out = nullptr;
    hipStream_t gstream = nullptr;
    std::map<long, hipGraphExec_t> graphs;
    std::set<long> seen_keys;
    bool use_graph = true, fed_ready = false;
    int fed_rows = 0, fed_pos = 0;
    surya_layout_feedback fbcfg{};
    bool have_sizes = false, fb_set = false;
    int graph_epoch = 0;
    void drop_graphs() {
        for (auto& kv : graphs) (void)hipGraphExecDestroy(kv.second);
        graphs.clear(); seen_keys.clear();
    }

    int init_rings() {
        const size_t B = c.max_batch;
        ring_cls = (size_t)RING_STEPS * B * c.label_count; ring_box = (size_t)RING_STEPS * B * 6; ring_tok = (size_t)RING_STEPS * B * 10;
        SA_HIP(hipMalloc((void**)&cls_ring, 2 * ring_cls * sizeof(float)));
        SA_HIP(hipMalloc((void**)&box_ring, 2 * ring_box * sizeof(float)));
        SA_HIP(hipMalloc((void**)&tok_ring, 2 * ring_tok * sizeof(int)));
        SA_HIP(hipMalloc((void**)&page_sizes_dev, B * 2 * sizeof(int)));
        SA_HIP(hipHostMalloc((void**)&ring_host, 2 * (ring_cls + ring_box + ring_tok) * 4 + B * 2 * sizeof(int), hipHostMallocDefault));
        for (auto& e : ev_ring) SA_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        SA_HIP(hipEventCreateWithFlags(&gev_in, hipEventDisableTiming));
        SA_HIP(hipEventCreateWithFlags(&gev_out, hipEventDisableTiming));
        SA_HIP(hipStreamCreateWithFlags(&gstream, hipStreamNonBlocking));
        return SA_OK;
    }
    void free_rings() {
        for (auto& kv : graphs) (void)hipGraphExecDestroy(kv.second);
        graphs.clear();
        if (cls_ring) (void)hipFree(cls_ring);
        if (box_ring) (void)hipFree(box_ring);
        if (tok_ring) (void)hipFree(tok_ring);
        if (page_sizes_dev) (void)hipFree(page_sizes_dev);
        if (ring_host) (void)hipHostFree(ring_host);
        for (auto& e : ev_ring) if (e) (void)hipEventDestroy(e);
        if (gev_in) (void)hipEventDestroy(gev_in);
        if (gev_out) (void)hipEventDestroy(gev_out);
        if (gstream) (void)hipStreamDestroy(gstream);
    }

    int set_feedback(const surya_layout_feedback* fb, int B, hipStream_t s) override {
        if (!fb) return SA_ERR_ARG;
        if (B <= 0 || B > c.max_batch) return SA_ERR_ARG;
        if (c.label_count + 6 > 64) return SA_ERR_UNSUPPORTED;
        if (c.family == SA_FAMILY_TABLE && fb->head_widths[0] + fb->head_widths[1] + fb->head_widths[2] + fb->head_widths[3] != c.label_count) return SA_ERR_SHAPE;
        if (c.family == SA_FAMILY_TABLE && (fb->head_widths[2] != 1 || fb->head_widths[0] < 1 || fb->head_widths[1] < 1 || fb->head_widths[3] < 1)) return SA_ERR_SHAPE;
        const bool changed = memcmp(&fbcfg, fb, sizeof(int32_t) * 7) != 0 || have_sizes != (fb->page_sizes != nullptr);
        fbcfg = *fb;
        have_sizes = fb->page_sizes != nullptr;
        fbcfg.page_sizes = nullptr;
        if (changed) drop_graphs();                              // captured runs hold the old rule's constants
        fb_set = true;
        if (have_sizes) {
            int* hs = reinterpret_cast<int*>(ring_host + 2 * (ring_cls + ring_box + ring_tok) * 4);
            SA_HIP(hipStreamSynchronize(s));                     // an earlier upload from this staging area may still be in flight
            memcpy(hs, fb->page_sizes, (size_t)B * 2 * sizeof(int));
            SA_HIP(hipMemcpyAsync(page_sizes_dev, hs, (size_t)B * 2 * sizeof(int), hipMemcpyHostToDevice, s));
        }
        return SA_OK;
    }

    int fed_steps_eager(int B, int n_steps, int ring, hipStream_t s) {
        const int Hd = c.dec_hidden;
        int rc;
        lay::LayoutFeedback fb;
        fb.boxes = boxes_dev; fb.len = len_dev; fb.page_sizes = have_sizes ? page_sizes_dev : nullptr;
        fb.family = c.family; fb.tokw = tokw(); fb.bbox_size = c.bbox_size; fb.vocab = c.vocab; fb.skew_scaler = fbcfg.skew_scaler;
        fb.relabel_a = fbcfg.relabel_ids[0]; fb.relabel_b = fbcfg.relabel_ids[1];
        fb.wcat = fbcfg.head_widths[0]; fb.wmer = fbcfg.head_widths[1]; fb.whdr = fbcfg.head_widths[3];
        fb.box_embed = c.box_embed; fb.category_count = c.category_count; fb.merge_count = c.merge_count; fb.embed_labels = c.label_count;
        for (int k = 0; k < n_steps; ++k) {
            if ((rc = decode_layers(B, s))) return rc;
            const size_t slot = (size_t)ring * RING_STEPS + k;
            fb.tok_ring = tok_ring + slot * c.max_batch * 10;
            hipLaunchKernelGGL((lay::layout_heads_kernel<T, true>), dim3(B), dim3(256), (size_t)Hd * 4, s, dx, W(SA_LW_DEC_FNORM), W(SA_LW_DEC_LN_W),
                               W(SA_LW_DEC_LN_B), W(SA_LW_DEC_LM_W), W(SA_LW_DEC_BB_W), W(SA_LW_DEC_BB_B),
                               cls_ring + slot * c.max_batch * c.label_count, box_ring + slot * c.max_batch * 6, Hd, c.label_count, c.rms_eps,
                               c.ln_eps, (long)Hd, fb, tabs_dev, dx, W(dec_base + SA_LD_CNORM), dh);
            if ((rc = (int)hipGetLastError())) return rc;
        }
        return SA_OK;
    }

    int decode_steps(const int32_t* boxes, int B, int pos0, int n_steps, int ring, hipStream_t s) override {
        if (B != batch_active) return SA_ERR_STATE;
        if (n_steps < 1 || n_steps > RING_STEPS || ring < 0 || ring > 1 || pos0 < 0 || pos0 + n_steps > c.max_boxes) return SA_ERR_ARG;
        if (!fb_set) return SA_ERR_STATE;                       // surya_layout_set_feedback first
        if (graph_epoch != tuning_epoch()) { drop_graphs(); graph_epoch = tuning_epoch(); }      // split-K shapes follow the tuning knobs
        int rc;
        if (boxes) {
            if ((rc = start_step(boxes, B, pos0, s))) return rc;
        } else if (!fed_ready || fed_rows != B || fed_pos != pos0) {
            return SA_ERR_STATE;                                 // nothing on the device to continue from at this position
        }
        fed_ready = false;
        const long key = (long)B * 64 + n_steps + (long)ring * (1L << 40);
        bool replayed = false;
        // hipGraph replay is opt-in (surya_set_tuning("graph", 1)), as for the recogniser: a run is ~1400 kernel nodes and the replay measured
        // SLOWER than plain launches enqueued ahead of the device (gpurun r04e / r04f, 32 pages: 628 vs 552 us per step; host-fed loop 570-587)
        if (use_graph && tuning().graph == 1 && !gemm_profiler().enabled) {
            auto it = graphs.find(key);
            if (it == graphs.end() && seen_keys.count(key)) {
                hipGraph_t g = nullptr;
                SA_HIP(hipStreamBeginCapture(gstream, hipStreamCaptureModeThreadLocal));
                rc = fed_steps_eager(B, n_steps, ring, gstream);
                hipError_t e = hipStreamEndCapture(gstream, &g);
                hipGraphExec_t ex = nullptr;
                if (!rc && e == hipSuccess && g) e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
                if (g) (void)hipGraphDestroy(g);
                if (rc || e != hipSuccess || !ex) { (void)hipGetLastError(); use_graph = false; }
                else it = graphs.emplace(key, ex).first;
            }
            seen_keys.insert(key);                               // the first run of a shape is eager: one-time attribute calls happen there
            if (it != graphs.end()) {
                SA_HIP(hipEventRecord(gev_in, s));
                SA_HIP(hipStreamWaitEvent(gstream, gev_in, 0));
                SA_HIP(hipGraphLaunch(it->second, gstream));
                SA_HIP(hipEventRecord(gev_out, gstream));
                SA_HIP(hipStreamWaitEvent(s, gev_out, 0));
                replayed = true;
            }
        }
        if (!replayed && (rc = fed_steps_eager(B, n_steps, ring, s))) return rc;
        const size_t mb = c.max_batch, off = (size_t)ring * RING_STEPS;
        float* hc = reinterpret_cast<float*>(ring_host) + (size_t)ring * (ring_cls + ring_box + ring_tok);
        float* hb = hc + ring_cls;
        int* ht = reinterpret_cast<int*>(hb + ring_box);
        SA_HIP(hipMemcpyAsync(hc, cls_ring + off * mb * c.label_count, (size_t)n_steps * mb * c.label_count * sizeof(float), hipMemcpyDeviceToHost, s));
        SA_HIP(hipMemcpyAsync(hb, box_ring + off * mb * 6, (size_t)n_steps * mb * 6 * sizeof(float), hipMemcpyDeviceToHost, s));
        SA_HIP(hipMemcpyAsync(ht, tok_ring + off * mb * 10, (size_t)n_steps * mb * 10 * sizeof(int), hipMemcpyDeviceToHost, s));
        SA_HIP(hipEventRecord(ev_ring[ring], s));
        fed_ready = true; fed_rows = B; fed_pos = pos0 + n_steps;
        return SA_OK;
    }

    int wait_steps(int ring, int B, int n_steps, float* cls, float* box, int32_t* tok) override {
        if (ring < 0 || ring > 1 || n_steps < 1 || n_steps > RING_STEPS || B < 1 || B > c.max_batch) return SA_ERR_ARG;
        SA_HIP(hipEventSynchronize(ev_ring[ring]));
        const size_t mb = c.max_batch;
        const float* hc = reinterpret_cast<const float*>(ring_host) + (size_t)ring * (ring_cls + ring_box + ring_tok);
        const float* hb = hc + ring_cls;
        const int* ht = reinterpret_cast<const int*>(hb + ring_box);
        const int tw = tokw();
        for (int k = 0; k < n_steps; ++k) {                      // device rings are [step][max_batch]; the caller's arrays [step][B]
            memcpy(cls + (size_t)k * B * c.label_count, hc + (size_t)k * mb * c.label_count, (size_t)B * c.label_count * sizeof(float));
            memcpy(box + (size_t)k * B * 6, hb + (size_t)k * mb * 6, (size_t)B * 6 * sizeof(float));
            memcpy(tok + (size_t)k * B * tw, ht + (size_t)k * mb * 10, (size_t)B * tw * sizeof(int));          // rows packed at the token width
        }
        return SA_OK;
    }
};

}  // namespace sa

using namespace sa;
struct surya_layout { std::unique_ptr<LayoutBase> impl; };

extern "C" {

int surya_layout_create(const surya_layout_config* cfg, const void* const* weights, int n_weights, surya_layout** out) {
    if (!cfg || !weights || !out) return SA_ERR_ARG;
    if (cfg->n_stages < 1 || cfg->n_stages > 8 || cfg->patch * cfg->patch * 3 > 64 || cfg->embed_dim % 64 || cfg->window != 8) return SA_ERR_UNSUPPORTED;
    if (cfg->img_h % cfg->patch || cfg->img_w % cfg->patch || cfg->dec_hidden % 64 || cfg->dec_inter % 64 || cfg->dec_heads % cfg->dec_kv_heads)
        return SA_ERR_SHAPE;
    if (cfg->max_batch <= 0 || cfg->max_boxes <= 0 || cfg->label_count <= 0) return SA_ERR_ARG;
    if (cfg->vocab <= cfg->bbox_size) return SA_ERR_SHAPE;       // box_embed / table_embed clamp corners to [0, bbox_size] and index [vocab]-row tables
    if (cfg->family != SA_FAMILY_LAYOUT && cfg->family != SA_FAMILY_TABLE) return SA_ERR_ARG;
    if (cfg->family == SA_FAMILY_TABLE &&
        (cfg->box_embed <= 0 || cfg->box_embed >= cfg->dec_hidden || cfg->category_count <= 0 || cfg->merge_count <= 0)) return SA_ERR_ARG;
    for (int i = 0; i < n_weights; ++i)
        if (!weights[i]) return SA_ERR_ARG;
    auto* h = new surya_layout();
    int rc;
    if (cfg->dtype == SA_DTYPE_F32) {
        auto m = std::make_unique<LayoutModel<float>>();
        rc = m->init(*cfg, weights, n_weights);
        h->impl = std::move(m);
    } else if (cfg->dtype == SA_DTYPE_BF16) {
        auto m = std::make_unique<LayoutModel<bf16_t>>();
        rc = m->init(*cfg, weights, n_weights);
        h->impl = std::move(m);
    } else {
        rc = SA_ERR_UNSUPPORTED;
    }
    if (rc) { delete h; return rc; }
    *out = h;
    return SA_OK;
}

int surya_layout_destroy(surya_layout* h) {
    if (!h) return SA_ERR_ARG;
    (void)hipDeviceSynchronize();
    delete h;
    return SA_OK;
}

int surya_layout_encode(surya_layout* h, const float* pixel_values, int batch, void* stream) {
    if (!h || !pixel_values) return SA_ERR_ARG;
    return h->impl->encode(pixel_values, batch, (hipStream_t)stream);
}

int surya_layout_decode_step(surya_layout* h, const int32_t* boxes, int batch, int position, float* class_logits, float* bbox, void* stream) {
    if (!h || !boxes || !class_logits || !bbox) return SA_ERR_ARG;
    return h->impl->decode_step(boxes, batch, position, class_logits, bbox, (hipStream_t)stream);
}

int surya_layout_prefill(surya_layout* h, const int32_t* boxes, int batch, int n_tokens, float* class_logits, float* bbox, void* stream) {
    if (!h || !boxes || !class_logits || !bbox) return SA_ERR_ARG;
    return h->impl->prefill(boxes, batch, n_tokens, class_logits, bbox, (hipStream_t)stream);
}

int surya_layout_set_feedback(surya_layout* h, const surya_layout_feedback* fb, int batch, void* stream) {
    if (!h || !fb) return SA_ERR_ARG;
    return h->impl->set_feedback(fb, batch, (hipStream_t)stream);
}

int surya_layout_decode_steps(surya_layout* h, const int32_t* boxes, int batch, int position, int n_steps, int ring, void* stream) {
    if (!h) return SA_ERR_ARG;
    return h->impl->decode_steps(boxes, batch, position, n_steps, ring, (hipStream_t)stream);
}

int surya_layout_wait_steps(surya_layout* h, int ring, int batch, int n_steps, float* class_logits, float* bbox, int32_t* fed_tokens) {
    if (!h || !class_logits || !bbox || !fed_tokens) return SA_ERR_ARG;
    return h->impl->wait_steps(ring, batch, n_steps, class_logits, bbox, fed_tokens);
}

int surya_layout_select(surya_layout* h, const int32_t* src_index, int n) {
    if (!h || !src_index) return SA_ERR_ARG;
    return h->impl->select(src_index, n);
}

int surya_layout_encoder_states(surya_layout* h, void* out, int batch, void* stream) {
    if (!h || !out) return SA_ERR_ARG;
    return h->impl->encoder_states(out, batch, (hipStream_t)stream);
}

}  // extern "C"
