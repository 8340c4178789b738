/*
 * surya_amd.h -- C ABI of libsurya_amd.so, the MI355X (gfx950) implementation of surya's batched model-inference
 * hot path. Plain pointers and sizes only; no torch types. All device pointers are HIP device memory owned by the
 * caller (PyTorch-ROCm caching allocator on the Python side); the library owns only what it allocates in
 * *_create(): KV-cache slots, activation workspaces and small pinned staging buffers, freed by *_destroy().
 *
 * Nothing like this exists in the reference (it has no native code at all): each entry point
 * cites the reference Python interface it replaces. The reference-side binding a maintainer would add is shown in
 * INTEGRATION.md.
 *
 * Conventions
 *   - every function returns int: 0 = ok, <0 = SA_ERR_* (caller error), >0 = hipError_t passthrough.
 *     Nothing throws across the ABI.
 *   - all work is enqueued on the hipStream_t passed in (as void*); functions do not synchronise unless
 *     documented ("sync"). A handle is re-entrant but not thread-safe: one host thread per handle, like the
 *     reference's single-threaded device path (surya/settings.py:179-183).
 *   - dtype: 0 = fp32 ("reference mode": exact-f32 MFMA, used for bit-exact token tests), 1 = bf16.
 */
#ifndef SURYA_AMD_H
#define SURYA_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SA_OK 0
#define SA_ERR_ARG (-1)
#define SA_ERR_SHAPE (-2)
#define SA_ERR_UNSUPPORTED (-3)
#define SA_ERR_STATE (-4)
#define SA_ERR_NOMEM (-5)

#define SA_DTYPE_F32 0
#define SA_DTYPE_BF16 1

/* Library / build info: returns a static string "surya_amd <version> gfx950". */
const char* surya_amd_version(void);

/* ------------------------------------------------------------------------------------------------------------
 * Recognition model: vision encoder + decoder + heads.
 * Replaces SuryaModel.forward (surya/common/surya/__init__.py:274-338), process_outputs / decode / prefill
 * (surya/recognition/__init__.py:294-471) and ContinuousBatchingCache (surya/recognition/cache.py:7-109).
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct surya_rec_config {
    /* vision encoder (surya/common/surya/encoder/config.py:18-53) */
    int32_t enc_depth, enc_hidden, enc_inter, enc_inter_pad, enc_heads;
    int32_t patch_dim, patch_dim_pad;    /* 588 and its K padding (multiple of 64) */
    int32_t merge;                       /* spatial_merge_size */
    int32_t window_tokens;               /* window_size / merge / patch_size (merged tokens per window side) */
    int32_t enc_out_hidden;
    uint32_t fullatt_mask;               /* bit i set: block i attends over the whole image */
    float enc_eps;
    /* decoder (surya/common/surya/decoder/config.py:28-85) */
    int32_t vocab, dec_hidden, dec_inter, dec_layers, dec_heads, dec_kv_heads, dec_head_dim;
    float dec_eps;
    /* model (surya/common/surya/config.py:12-71) */
    int32_t bbox_size, embed_multiplier;
    int32_t image_token_id, pad_token_id, eos_token_id;
    /* capacities */
    int32_t max_slots;            /* continuous-batching slots (RecognitionPredictor batch size) */
    int32_t max_kv_len;           /* prompt + generated tokens per slot */
    int32_t max_patches;          /* encoder chunk capacity in patches (reference: encoder_chunk_size) */
    int32_t max_prefill_tokens;   /* packed prompt tokens per prefill call */
    int32_t dtype;
} surya_rec_config;

/* Weight table: device pointers in the compute dtype (inv_freq tables are fp32), kernel layout:
 *   Linear weights [out, in_padded] row-major; gate|up fused and row-interleaved (g0,u0,g1,u1,...), q|k|v fused.
 * Index = SA_RW_* for globals, SA_RW_ENC(l, k), SA_RW_DEC(l, k) for layers. The Python loader
 * (surya_amd/recognition/weights.py) builds it from the reference's state-dict names. */
enum {
    SA_RW_PATCH = 0, SA_RW_MERGER_LN, SA_RW_FC1_W, SA_RW_FC1_B, SA_RW_FC2_W, SA_RW_FC2_B, SA_RW_IMG_H, SA_RW_IMG_W,
    SA_RW_DEC_NORM, SA_RW_TOK_EMBED, SA_RW_LM_W, SA_RW_LM_B, SA_RW_BBOX_W, SA_RW_BBOX_B, SA_RW_ENC_INVFREQ,
    SA_RW_DEC_INVFREQ, SA_RW_GLOBALS
};
enum { SA_RE_NORM1 = 0, SA_RE_QKV_W, SA_RE_QKV_B, SA_RE_PROJ_W, SA_RE_PROJ_B, SA_RE_NORM2, SA_RE_GU_W, SA_RE_GU_B,
       SA_RE_DOWN_W, SA_RE_DOWN_B, SA_RE_COUNT };
enum { SA_RD_LN1 = 0, SA_RD_QKV_W, SA_RD_QKV_B, SA_RD_O_W, SA_RD_LN2, SA_RD_GU_W, SA_RD_DOWN_W, SA_RD_COUNT };
#define SA_RW_ENC(l, k) (SA_RW_GLOBALS + (l) * SA_RE_COUNT + (k))
#define SA_RW_DEC(cfg_enc_depth, l, k) (SA_RW_GLOBALS + (cfg_enc_depth) * SA_RE_COUNT + (l) * SA_RD_COUNT + (k))
#define SA_RW_TOTAL(enc_depth, dec_layers) (SA_RW_GLOBALS + (enc_depth) * SA_RE_COUNT + (dec_layers) * SA_RD_COUNT)

typedef struct surya_rec surya_rec;

/* Bytes of device memory *_create will allocate for this config (KV cache + workspaces). */
size_t surya_rec_workspace_bytes(const surya_rec_config* cfg);

/* Create a model instance. `weights` has SA_RW_TOTAL entries; the caller keeps the tensors alive. */
int surya_rec_create(const surya_rec_config* cfg, const void* const* weights, int n_weights, surya_rec** out);
int surya_rec_destroy(surya_rec* h);

/* Host-only planning of the vision encoder's index math for one packed batch of images: window order
 * (get_window_index, encoder/__init__.py:552-597), rotary position ids (rot_pos_emb, :523-550), window / image
 * segment boundaries (:615-646). Runs without a GPU; exposed for tests.
 *   grid_hw      [n_images*2]  patch-grid (h, w) per image (both even)
 *   src_row      [P]           window-ordered row r reads original tile row src_row[r]
 *   pos_hw       [P*2]         (h, w) patch coordinates in window order
 *   cu_window    [*n_windows+1] patch offsets of the (non-empty) windows, capacity P/4 + n_images + 1
 *   merged_src   [P/merge^2]   window-ordered merged token g came from original merged token merged_src[g]
 */
int surya_rec_plan_encoder(const surya_rec_config* cfg, const int32_t* grid_hw, int n_images, int32_t* src_row,
                           int32_t* pos_hw, int32_t* cu_window, int32_t* n_windows, int32_t* merged_src);

/* Prefill n sequences into KV slots (replaces RecognitionPredictor.prefill's model call + cache merge,
 * recognition/__init__.py:354-471 and cache.py:57-105).
 *   tiles        device fp32 [P, patch_dim], merge-block-major rows as produced by the processor
 *                (processor/__init__.py:214-228), images concatenated in sequence order
 *   grid_hw      host [n_images*2]
 *   input_ids    host, packed prompt tokens of all sequences (no padding)
 *   seq_offsets  host [n_seqs+1]
 *   slot_ids     host [n_seqs] destination slots (distinct, < max_slots)
 * Image features are scattered to the positions where input_ids == image_token_id, in order
 * (common/surya/__init__.py:214-225). Afterwards each slot holds its first generated token as next input and
 * outputs(step 0) hold token / score / bbox for the slots given. Enqueue only. */
int surya_rec_prefill(surya_rec* h, const float* tiles, const int32_t* grid_hw, int n_images, const int32_t* input_ids,
                      const int32_t* seq_offsets, const int32_t* slot_ids, int n_seqs, void* stream);

/* Look-ahead encoding (no counterpart in the reference, which encodes inside the prefill call, common/surya/__init__.py:
 * 130-195): run the vision encoder for the images of the NEXT prompts on the library's own low-priority stream (after
 * the work already queued on `stream`, which produced `tiles`), concurrently with decode steps on `stream`. The merged
 * image embeddings stay inside the handle; a later surya_rec_prefill with tiles == NULL consumes them front to back
 * (its images must be the next ones in the order given here; grid_hw is still required). At most one look-ahead batch is
 * outstanding: SA_ERR_STATE while the previous one has unconsumed images, SA_ERR_SHAPE above max_prefill_tokens image
 * tokens. n_images == 0 (tiles / grid_hw may be NULL) discards whatever is unconsumed: what a caller does before it starts a
 * new batch of lines if its previous loop may have ended early. Enqueue only. */
int surya_rec_encode_ahead(surya_rec* h, const float* tiles, const int32_t* grid_hw, int n_images, void* stream);

/* Set the list of slots that take part in decode steps (host bookkeeping of batch_prompt_mapping,
 * recognition/__init__.py:125-136). */
int surya_rec_set_active(surya_rec* h, const int32_t* slots, int n_active, void* stream);

/* Run `n_steps` greedy decode steps for the active slots without host round trips (replaces
 * RecognitionPredictor.decode + process_outputs, recognition/__init__.py:294-352). Step s writes
 * outputs(step s). n_steps <= SA_MAX_STEPS. Enqueue only. */
#define SA_MAX_STEPS 16
int surya_rec_decode(surya_rec* h, int n_steps, void* stream);

/* Sync the stream and copy the outputs of steps [0, n_steps) to host arrays indexed [step][slot]:
 *   tokens int32 [n_steps*max_slots], scores fp32 [n_steps*max_slots], bboxes int32 [n_steps*max_slots*6]. */
int surya_rec_read_outputs(surya_rec* h, int n_steps, int32_t* tokens, float* scores, int32_t* bboxes, void* stream);

/* Pipelined pair (the reference blocks on .cpu() after every step, recognition/__init__.py:545-576; here the host may run
 * one call behind the device): decode_async = surya_rec_decode with its outputs in ring half `ring` (0 or 1,
 * n_steps <= SA_MAX_STEPS / 2) plus an async copy of that half to pinned host memory and an event; wait_outputs blocks on
 * that event only (NOT on the stream, so a later decode_async may already be queued) and returns the arrays laid out
 * like read_outputs. A slot whose line finished inside call n still steps through call n + 1; its KV length is clamped
 * to max_kv_len - 1 on the device and the caller ignores its outputs. */
int surya_rec_decode_async(surya_rec* h, int n_steps, int ring, void* stream);
int surya_rec_wait_outputs(surya_rec* h, int n_steps, int ring, int32_t* tokens, float* scores, int32_t* bboxes);

/* Line-crop pre-processing on the device: page pixels -> image_tiles for n_lines lines in one call.
 * Replaces, per line, slice_bboxes_from_image / slice_and_pad_poly (surya/input/processing.py:35-101), scale_to_fit
 * (surya/common/surya/processor/__init__.py:141-178, LANCZOS4) and _process_and_tile (:185-230: CUBIC round-up to multiples of
 * patch * merge, x / 255 in fp64, (x - mean) / std in fp32, merge-block-major patch rows).
 * pages: device uint8, every page HWC at its descriptor's byte offset, `pixel_stride` bytes per pixel: 3 (RGB) or 4 (RGBX, the
 * layout PIL keeps in memory -- the host then uploads page memory as is instead of repacking it, ~2 ms per 1024^2 page). lines: device array of n_lines descriptors
 *   { int64 page_off; int32 page_w, page_h; int32 x0, y0, cw, ch (crop rectangle, inside the page, >= 1 px);
 *     int32 has_poly; float poly[8] (4 vertices relative to the crop origin; pixels outside read as pad_value);
 *     int32 mid_w, mid_h (size after scale_to_fit); int32 out_w, out_h (multiples of patch * merge);
 *     int64 mask_off (bytes into mask_arena, cw * ch per polygon line); int64 mid_off (floats into mid_arena, 3 mid_w mid_h per
 *     line whose mid size differs from its crop size); int64 tile_row (first row of the line in tiles) }   -- 112 bytes, C layout.
 * tiles: device fp32 [sum (out_h / patch) (out_w / patch)][3 patch^2]. The host computes only these integers
 * (surya_amd/recognition/preprocess_gpu.py). any_poly: some line has a polygon; max_stage1_width: largest mid_w among the lines whose
 * mid size differs from their crop size (0 = no line needs the Lanczos stage). Enqueue only; every buffer is caller-owned. */
int surya_rec_preprocess(const uint8_t* pages, const void* lines, int n_lines, uint8_t* mask_arena, float* mid_arena, float* tiles,
                         int patch_size, int merge_size, float pad_value, const float* mean, const float* std, int any_poly,
                         int max_stage1_width, int pixel_stride, void* stream);

/* Test hooks (tolerance tests of intermediate tensors):
 *   encode_only: run the vision encoder + 2-D position embedding, write [P/merge^2, dec_hidden] features in
 *                ORIGINAL token order (= get_image_embeddings, common/surya/__init__.py:130-195) to `out`
 *                (device, compute dtype).
 *   copy_last_logits: fp32 logits [rows, vocab] of the last prefill / decode step into `dst` (device, capacity
 *                max_rows rows); row r = r-th sequence of the prefill / r-th active slot. The product path keeps logits
 *                in LDS (greedy reduction in the lm_head epilogue), so this recomputes them on `stream` from the
 *                final-norm rows of that step. */
int surya_rec_encode_only(surya_rec* h, const float* tiles, const int32_t* grid_hw, int n_images, void* out, void* stream);
int surya_rec_copy_last_logits(surya_rec* h, float* dst, int max_rows, int* rows, void* stream);
/* Force the token fed to the next decode step (teacher forcing in parity tests). */
int surya_rec_set_next_tokens(surya_rec* h, const int32_t* slots, const int32_t* tokens, int n, void* stream);

/* MXFP8 weights for the decode steps (BASELINE.json configs[4]: the fp8 MFMA weight path of the texify configuration; the
 * reference has no fp8 mode -- its decode loop, surya/recognition/__init__.py:473-607, runs the checkpoint dtype). bf16
 * models only. Each projection is given as OCP e4m3 bytes in the SAME kernel layout as its bf16 weight (qkv fused, gate|up
 * row-interleaved, [out][in]) plus one E8M0 scale byte per 32 consecutive input features, stored K-TILE-MAJOR:
 * [in / 128][out][4] bytes (the 4 block scales of a row inside one 128-wide K-tile are one dword, and the dwords of
 * consecutive rows are contiguous -- what the GEMM's scale fetch reads in one piece). After this call
 * surya_rec_decode / _decode_async multiply MXFP8 activations (quantised inside the producing kernels) by these weights with
 * v_mfma_scale_f32_32x32x64_f8f6f4; prefill keeps the bf16 weights. table == NULL returns to the bf16 decode path.
 * The caller keeps the tensors alive. Needs dec_hidden, heads * head_dim and intermediate sizes that are multiples of 128. */
enum { SA_MX_QKV_W = 0, SA_MX_QKV_S, SA_MX_O_W, SA_MX_O_S, SA_MX_GU_W, SA_MX_GU_S, SA_MX_DOWN_W, SA_MX_DOWN_S, SA_MX_COUNT };
enum { SA_MX_LM_W = 0, SA_MX_LM_S, SA_MX_GLOBALS };      /* after the per-layer entries */
#define SA_MX_TOTAL(dec_layers) ((dec_layers) * SA_MX_COUNT + SA_MX_GLOBALS)
int surya_rec_set_mx_weights(surya_rec* h, const void* const* table, int n);

/* FP8 (OCP e4m3) KV cache for the decode steps -- the second half of configs[4]'s fp8 path: at the texify horizon (KV 202 -> 970)
 * the decode-attention kernel is the largest of the step and moves its rows at HBM speed, so the lever is the bytes. Format "KV8":
 * per (slot, kv head, token) one power-of-two scale (the smallest with absmax / scale <= 448) + round-to-nearest-even e4m3
 * elements; K rows [slot][kv_head][max_kv_len][head_dim] bytes, V transposed per 128-token tile [slot][kv_head][T8 / 128][head_dim][128] bytes, scales fp32
 * [slot][kv_head][T8], T8 = max_kv_len rounded up to 256; the arrays live inside the handle. Prefill still attends over the bf16
 * cache and quantises the prompt's rows; decode steps read and append fp8 only (decode_attn_kv8.h). bf16 models, head_dim in
 * {32, 64, 128}. No reference counterpart (surya/recognition/__init__.py:379-395 offers HQQ 8-bit only): the format is pinned by
 * oracle/mx_oracle.py::kv8_quantize and tests/test_gpu_kv8.py. Call while no line is in flight. */
int surya_rec_set_kv_fp8(surya_rec* h, int on);

/* ------------------------------------------------------------------------------------------------------------
 * Op-level entry points (unit tests of the kernels through the same library; row-major, compute dtype).
 * ---------------------------------------------------------------------------------------------------------- */
/* C[M,N] = X[M,K] W[N,K]^T + bias, epilogue: 0 none/bias, 1 +residual R, 2 gelu, 3 swiglu (W rows interleaved,
 * C is [M,N/2]), 4 hardswish, 5 relu. out_f32 != 0: C (and R) are fp32 regardless of dtype. */
int surya_op_gemm(int dtype, int out_f32, int epi, const void* X, long ldx, const void* W, long ldw, void* C, long ldc,
                  const void* bias, const void* R, long ldr, int M, int N, int K, void* stream);
int surya_op_rmsnorm(int dtype, const void* x, long ldx, const void* w, void* y, long ldy, int rows, int C, float eps,
                     void* stream);
/* Attention kernels by themselves (tests against fp32 PyTorch SDPA; synchronous -- the call returns after the kernel ran).
 * surya_op_attn: segment attention = the vision encoder's window / whole-image attention (non-causal varlen,
 * surya/common/surya/encoder/__init__.py:238-261) and the decoder prefill's causal GQA (decoder/__init__.py:101-128).
 * dtype bf16 runs attn_mfma_kernel<head_dim>, fp32 runs attn_valu_kernel (reference mode). Segment s has seg_len[s] queries
 * and keys; its first query / key / value / output row starts at element offset q_off / k_off / v_off / o_off[s] (host
 * arrays) of q / k / v / out (device), rows `*_row` elements apart, heads `*_head` elements apart; query head h reads kv
 * head h / group. head_dim in {32, 64, 80, 128}.
 * surya_op_decode_attn: one decode step's attention launch exactly as RecModel::decode_layer issues it: row r = active slot
 * active_slots[r] with row_len[r] cached tokens; q|k|v of the new token = sum of n_slabs fp32 split-K slabs
 * qkv_part[slab][rows][(heads + 2 kv_heads) * head_dim] + qkv_bias, rounded to the storage dtype; RoPE from the (cos, sin) table
 * rope_cs[max_kv_len][head_dim / 2][2]; k, v appended to the caches [slot][kv_head][max_kv_len][head_dim] at row_len[r];
 * out[r][heads * head_dim] = attention over row_len[r] + 1 keys (decoder/__init__.py:193-234). bf16 runs
 * decode_attn_flash_kernel, fp32 decode_attn_mfma_kernel. All pointers device. Enqueue only. */
int surya_op_attn(int dtype, int head_dim, const void* q, const void* k, const void* v, void* out, const int32_t* seg_len,
                  const int64_t* q_off, const int64_t* k_off, const int64_t* v_off, const int64_t* o_off, int n_seg, int heads, int group,
                  int causal, float scale, long q_row, long q_head, long k_row, long k_head, long o_row, long o_head, void* stream);
int surya_op_decode_attn(int dtype, int head_dim, const float* qkv_part, int n_slabs, const void* qkv_bias, void* out, void* kcache,
                         void* vcache, const int32_t* active_slots, const int32_t* row_len, const float* rope_cs, int rows, int heads,
                         int kv_heads, int max_kv_len, float scale, void* stream);
/* The KV8 kernels by themselves (tests): surya_op_kv8_quant_rows quantises rows (tok_slot[i], tok_pos[i]) of bf16 caches
 * [slot][kv_head][max_kv_len][head_dim] into the KV8 arrays (layout above); surya_op_decode_attn_kv8 = surya_op_decode_attn on those
 * arrays (bf16 model path only), appending the new token's quantised k / v. All pointers device; enqueue only. */
int surya_op_kv8_quant_rows(int head_dim, const void* kcache, const void* vcache, const int32_t* tok_slot, const int32_t* tok_pos, int n_tokens,
                            void* k8, void* v8t, float* kscale, float* vscale, int kv_heads, int max_kv_len, void* stream);
int surya_op_decode_attn_kv8(int head_dim, const float* qkv_part, int n_slabs, const void* qkv_bias, void* out, void* k8, void* v8t,
                             float* kscale, float* vscale, const int32_t* active_slots, const int32_t* row_len, const float* rope_cs, int rows,
                             int heads, int kv_heads, int max_kv_len, float scale, void* stream);

/* MXFP8 ops (csrc/gemm_mx.h). quantize: fp32 rows [rows][K], K % 128 == 0 -> e4m3 [rows][K] + e8m0 scales K-tile-major
 * [K / 128][rows][4], with the rule every producer kernel uses (block scale = smallest power of two that keeps absmax <=
 * 448, round to nearest even). gemm_mx: C[M,N] fp32 = X W^T from MXFP8 operands (scales K-tile-major with M resp. N rows),
 * M <= 256. mode 0: one pass; mode 1: split-K, `C` receives the slabs [*splitk][M][N] (capacity 8 slabs) and the caller sums
 * them; mode 2: SwiGLU epilogue -> MXFP8 [M][N/2] in q_out, scales [N / 256][M][4] in sq_out (N % 256 == 0). */
int surya_op_mx_quantize(const float* x, int rows, int K, uint8_t* q, uint8_t* scales, void* stream);
int surya_op_gemm_mx(int mode, const uint8_t* X, const uint8_t* SX, const uint8_t* W, const uint8_t* SW, int M, int N, int K,
                     float* C, int* splitk, uint8_t* q_out, uint8_t* sq_out, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Detection model: EfficientViT-L backbone + SegFormer-style decode head + sigmoid + x4 bilinear upsample.
 * Replaces EfficientViTForSemanticSegmentation.forward (surya/detection/model/encoderdecoder.py:734-753) and the
 * interpolate + .float() of DetectionPredictor.batch_detection (surya/detection/__init__.py:119-132).
 * The network is described as a list of ops over NHWC activation buffers (built by surya_amd/detection/plan.py from
 * the reference's state-dict; BatchNorm folded, 3x3 weights as [Cout][ky][kx][Cin] with K padded to x64).
 * ---------------------------------------------------------------------------------------------------------- */
enum { SA_DET_INPUT = 0,       /* fp32 NCHW pixels -> NHWC, channels padded to `cout`                      */
       SA_DET_CONV,            /* dense KxK conv as implicit GEMM: bias, act, optional residual `res`        */
       SA_DET_DWCONV,          /* depthwise KxK: weights [K*K][C], bias, act                                 */
       SA_DET_GROUPED1X1,      /* grouped 1x1, p0 = channels per group (in == out)                           */
       SA_DET_LITEMLA,         /* ReLU linear attention; in0 = qkv conv, in1 = aggregated qkv; p0 = head dim */
       SA_DET_UPCAT,           /* bilinear resize of in0 into channels [p0, p0+cin) of `out` (cout wide)     */
       SA_DET_CLASSIFY,        /* 1x1 conv to `cout` labels + sigmoid -> fp32 planes                         */
       SA_DET_UPSAMPLE_OUT,    /* fp32 planes -> output size, bilinear                                       */
       /* Decode head in its folded form (round 4; surya/detection/model/encoderdecoder.py:699-722 is linear up to the ReLU, and a
        * 1x1 convolution commutes with a bilinear resize): linear_fuse(cat(up(linear_c_s(x_s)))) = sum_s up(A_s x_s) + c with
        * A_s = (bn_scale * W_fuse[:, s]) W_c_s built at load. The 4 x decoder_layer_hidden concat and the K = 4 x 128 GEMM over it
        * never exist: every stage runs ONE 1x1 conv to decoder_hidden channels at its own resolution, and the sum + ReLU + classifier
        * + sigmoid is one pass over the full-resolution stage. */
       SA_DET_UPSUM_SRC,       /* declares in0 ([hin, win, cin] per image) as a low-resolution addend of the next UPSUM_CLASSIFY (<= 3) */
       SA_DET_UPSUM_CLASSIFY };/* planes = sigmoid(classifier(relu(in0 + sum of the bilinearly resized addends))): in0 [hin, win, cin], `cout` labels */
enum { SA_ACT_NONE = 0, SA_ACT_HSWISH = 1, SA_ACT_RELU = 2 };

typedef struct surya_det_op {
    int32_t type;
    int32_t in0, in1, out, res;      /* buffer ids (-1 = none) */
    int32_t cin, cout, k, stride, act;
    int32_t hin, win, hout, wout;
    int32_t w_idx, b_idx;            /* weight-table indices (-1 = none) */
    int32_t p0, p1;                  /* CONV: pad, Kpad; see the enum for others */
} surya_det_op;

typedef struct surya_det_config {
    int32_t n_ops, max_batch, height, width, num_labels, dtype;
} surya_det_config;

typedef struct surya_det surya_det;

/* buf_elems[i] = elements PER IMAGE of activation buffer i (the library allocates max_batch times that). */
int surya_det_create(const surya_det_config* cfg, const surya_det_op* ops, const void* const* weights, int n_weights,
                     const size_t* buf_elems, int n_bufs, surya_det** out);
int surya_det_destroy(surya_det* h);
/* pixel_values: device fp32 [batch, 3, H, W] (rescaled + ImageNet-normalised, surya/detection/processor.py:126-146).
 * heatmaps: device fp32 [batch, labels, H, W] (may be NULL); lowres: device fp32 [batch, labels, H/4, W/4] (may be NULL)
 * = the model's own output before the predictor-side upsample. Enqueue only. */
int surya_det_forward(surya_det* h, const float* pixel_values, int batch, float* heatmaps, float* lowres, void* stream);
/* Measurement support (tools/det_op_times.py, bench.py's detection buckets): the same forward with a hipEvent in front of every op of the
 * list; op_ms[i] (host, n_op_ms >= surya_det_op_count) = milliseconds of op i, 0 for an op folded into a fused form (sa::Tuning det_fuse).
 * Synchronises the stream. Not for timed regions. */
int surya_det_op_count(surya_det* h);
int surya_det_forward_timed(surya_det* h, const float* pixel_values, int batch, float* heatmaps, float* lowres, void* stream, float* op_ms,
                            int n_op_ms);
/* Same forward from the resized pages themselves: device uint8 [batch, H, W, pixel_stride], pixel_stride 3 (RGB) or 4 (RGBX =
 * PIL's in-memory layout, uploaded without repacking; the fourth byte is ignored). The rescale
 * (x * 1/255 in fp32) and normalisation ((x - mean) / std) of SegformerImageProcessor._preprocess
 * (surya/detection/processor.py:126-146) run inside the first layout kernel: bit-identical pixel_values, a quarter of the
 * PCIe bytes, no per-pixel host work. */
int surya_det_forward_u8(surya_det* h, const uint8_t* pixels_nhwc, int pixel_stride, const float* mean, const float* std, int batch,
                         float* heatmaps, float* lowres, void* stream);

/* One Pillow `Image.resize(size, LANCZOS)` of an 8-bit RGB page on the device (detection side). Replaces the
 * two host resizes of DetectionPredictor.prepare_image (surya/detection/__init__.py:50-57: img.thumbnail(size, LANCZOS) then
 * img.resize(size, LANCZOS) = two calls here; the sizes and the 22-bit fixed-point coefficient tables of Pillow's resampler come
 * from surya_amd/common/pil_resample.py, which restates Pillow's precompute_coeffs / normalize_coeffs_8bpc). Bit-identical to
 * Pillow: int32 accumulation from 2^21, arithmetic shift by 22, clip to 0..255, horizontal pass first into `tmp`.
 *   src / dst   device uint8 [h][w][pix], pix = 3 (RGB) or 4 (RGBX, fourth byte ignored / written 0)
 *   bounds_*    device int32 [out][2] = (first source index, tap count); taps_* device int32 [out][ksize_*]
 *               (x tables may be NULL when the width does not change, y tables when the height does not)
 *   tmp         device scratch of src_h * dst_w * 4 bytes, needed when both axes change. Enqueue only. */
int surya_resample_lanczos_u8(const uint8_t* src, int src_w, int src_h, int src_pix, uint8_t* dst, int dst_w, int dst_h, int dst_pix,
                              const int32_t* bounds_x, const int32_t* taps_x, int ksize_x, const int32_t* bounds_y,
                              const int32_t* taps_y, int ksize_y, uint8_t* tmp, void* stream);

/* Heat map -> text boxes on the device. Replaces detect_boxes (surya/detection/heatmap.py:27-107:
 * get_dynamic_thresholds :14-24, cv2.connectedComponentsWithStats, per-component cv2.dilate + cv2.minAreaRect + cv2.boxPoints,
 * corner order, confidence = component max / page max) for `batch` pages at once; what is left for the host is
 * get_and_clean_boxes' rescale / fit_to_bounds / clean_boxes on a few hundred 4-point boxes (heatmap.py:127-136).
 * heat: device fp32; the text heat map of page b starts at heat + b * page_stride (floats) and is [height][width]
 * (width % 4 == 0; values must be non-negative, as sigmoid outputs are). boxes: device fp32 [batch][max_boxes][4][2] (x, y),
 * clockwise from the corner with the smallest x + y, in heat-map pixels; conf: device fp32 [batch][max_boxes]; count: device
 * int32 [batch] = boxes of the page in raster order of their first pixel (the order cv2 labels them), or -1 if the page has
 * more than max_boxes components (nothing else is written for that page). workspace: device, caller-owned,
 * >= surya_det_boxes_workspace_bytes(...). Enqueue only. */
size_t surya_det_boxes_workspace_bytes(int batch, int height, int width, int max_boxes);
int surya_det_boxes(const float* heat, long page_stride, int batch, int height, int width, float text_threshold, float low_text,
                    int max_boxes, float* boxes, float* conf, int32_t* count, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Layout model family: Donut-Swin window-attention encoder + ADETR decoder with cross- and self-attention.
 * Replaces DonutSwinLayoutModel.forward (surya/layout/model/encoder.py:36-80, surya/common/donut/encoder.py) and
 * SuryaLayoutDecoder.forward (surya/layout/model/decoder.py:96-131, surya/common/adetr/decoder.py) as
 * LayoutPredictor.batch_layout_detection calls them (surya/layout/__init__.py:95-131): one `encode` per image batch, then one
 * `decode_step` per box until every image emitted its end token (the per-step host round trip is the reference's own).
 * Window 8 (64 tokens), head dim 32 in the encoder; image sides a multiple of patch * 2^(stages-1) (the reference's per-stage sin-cos
 * tables ask for the same), every stage grid at least one window; grids that are not whole windows are zero-padded inside every block
 * as DonutSwinLayer.maybe_pad does (surya/common/donut/encoder.py:588-596, 668).
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct surya_layout_config {
    int32_t img_h, img_w, patch, embed_dim, n_stages;
    int32_t depths[8], heads[8], kv_heads[8];
    int32_t window;
    float enc_eps;                    /* layer_norm_eps of the Swin blocks */
    int32_t encoder_length;           /* rows of the learned position_embeddings */
    int32_t dec_layers, dec_hidden, dec_inter, dec_heads, dec_kv_heads;
    int32_t vocab, label_count, bbox_size;
    float rms_eps, ln_eps;
    int32_t max_batch, max_boxes, dtype;
    /* Model family on the same encoder / decoder stack:
     *   SA_FAMILY_LAYOUT (0): surya/layout -- 7-number tokens, BboxEmbedding (15 tables of dec_hidden), double residual flow
     *     (layout/model/config.py:221), heads = lm_head [label_count] + sigmoid(bbox_head + bias);
     *   SA_FAMILY_TABLE (1): surya/table_rec -- 10-number tokens (bbox 6, category, merges, colspan, is_header), LabelEmbedding =
     *     concat(14 box tables of box_embed, category + merge + colspan tables of dec_hidden - box_embed)
     *     (table_rec/model/decoder.py:12-73), the plain residual flow (adetr/decoder.py:395-417), heads = the four non-bbox
     *     box_property_heads stacked [category | merges | colspan | is_header] (label_count rows in all, no bias) + sigmoid(bbox head). */
    int32_t family, box_embed, category_count, merge_count;
} surya_layout_config;
enum { SA_FAMILY_LAYOUT = 0, SA_FAMILY_TABLE = 1 };

/* Weight table (device pointers, compute dtype unless noted; Linear weights [out, in]):
 *   globals SA_LW_*; PATCH_W is [embed_dim][64] = the Conv2d weight flattened (c, ky, kx) and zero padded; DEC_INVFREQ fp32 [hd / 2];
 *   DEC_ZERO_BIAS = zeros [(heads + 2 kv_heads) * hd] (the ADETR attention has no qkv bias; the fused decode-attention kernel takes one);
 *   EMB_TABLES = 17 slots: w, h, cx, cy, xskew, yskew, x1, y1, x2, y2, x3, y3, x4, y4 ([vocab][hidden], table family [vocab][box_embed]), then
 *     layout: label ([label_count][hidden]) + 2 unused; table: category ([category_count][P]), merge ([merge_count][P]), colspan ([vocab][P]),
 *     P = dec_hidden - box_embed;
 *   per stage SA_LS_* (SINCOS = the stage's 2-D sin-cos table [tokens][dim] built as the reference builds it; MERGE_* of the last stage
 *   are ignored), then per block SA_LB_* (QKV fused q | k | v rows; RELBIAS fp32 [heads][64][64] = relative_position_bias_table gathered
 *   through relative_position_index, rounded to the compute dtype first); then per decoder layer SA_LD_* (CKV_W = cross k | v rows fused,
 *   QKV_W = self q | k | v fused, GU_W = gate / up rows interleaved g0, u0, g1, u1 ...). */
enum { SA_LW_PATCH_W = 0, SA_LW_PATCH_B, SA_LW_EMB_LN_W, SA_LW_EMB_LN_B, SA_LW_POS_EMB, SA_LW_DEC_FNORM, SA_LW_DEC_LN_W, SA_LW_DEC_LN_B,
       SA_LW_DEC_LM_W, SA_LW_DEC_BB_W, SA_LW_DEC_BB_B, SA_LW_DEC_INVFREQ, SA_LW_DEC_ZERO_BIAS, SA_LW_EMB_TABLES, SA_LW_GLOBALS = SA_LW_EMB_TABLES + 17 };
enum { SA_LS_SINCOS = 0, SA_LS_MERGE_NORM_W, SA_LS_MERGE_NORM_B, SA_LS_MERGE_RED_W, SA_LS_COUNT };
enum { SA_LB_LN1_W = 0, SA_LB_LN1_B, SA_LB_QKV_W, SA_LB_QKV_B, SA_LB_RELBIAS, SA_LB_PROJ_W, SA_LB_PROJ_B, SA_LB_LN2_W, SA_LB_LN2_B, SA_LB_FC1_W,
       SA_LB_FC1_B, SA_LB_FC2_W, SA_LB_FC2_B, SA_LB_COUNT };
enum { SA_LD_CNORM = 0, SA_LD_CQ_W, SA_LD_CKV_W, SA_LD_CO_W, SA_LD_CO_B, SA_LD_TNORM, SA_LD_QKV_W, SA_LD_TO_W, SA_LD_TO_B, SA_LD_MNORM, SA_LD_GU_W,
       SA_LD_DOWN_W, SA_LD_COUNT };

typedef struct surya_layout surya_layout;
int surya_layout_create(const surya_layout_config* cfg, const void* const* weights, int n_weights, surya_layout** out);
int surya_layout_destroy(surya_layout* h);
/* pixel_values: device fp32 [batch, 3, img_h, img_w] (rescaled + normalised by the processor). Runs the encoder, keeps its output
 * inside the handle and projects every decoder layer's cross-attention keys / values; resets the self-attention caches. Enqueue only. */
int surya_layout_encode(surya_layout* h, const float* pixel_values, int batch, void* stream);
/* One decoder token per image at cache position `position` (0 = the start token): boxes host int32 [batch][7] = (cx, cy, w, h, xskew,
 * yskew, label) as the reference feeds them back (table family: [batch][10] = bbox 6, category, merges, colspan, is_header -- a prompt of
 * T tokens is T calls, which is what a causal prefill computes); class_logits host fp32 [batch][label_count], bbox host fp32 [batch][6] (after the
 * sigmoid). Synchronises the stream (the reference moves both to the host after every step as well). */
int surya_layout_decode_step(surya_layout* h, const int32_t* boxes, int batch, int position, float* class_logits, float* bbox, void* stream);
/* A decoder PROMPT of n_tokens (<= 64) tokens per row in one pass -- the reference's first decoder call, prefill = True
 * (surya/table_rec/__init__.py:60-68; the layout prompt is a single token) -- at positions 0 .. n_tokens - 1; boxes host int32
 * [batch][n_tokens][token width]; outputs as surya_layout_decode_step, for every row's LAST prompt token; the following decode steps
 * continue at position n_tokens. Equivalent to n_tokens decode steps (a causal prefill is the same arithmetic); SA_ERR_ARG when the prompt
 * does not fit the borrowed encoder workspaces -- callers then take the step-by-step route. Synchronises the stream. */
int surya_layout_prefill(surya_layout* h, const int32_t* boxes, int batch, int n_tokens, float* class_logits, float* bbox, void* stream);
/* Device-fed decode steps (round 4). The reference's box loop moves every step's outputs to the host, forms the next token there and sends
 * it back (surya/layout/__init__.py:110-177; surya/table_rec/__init__.py:57-129). Here the heads kernel forms that token itself --
 *   layout: (trunc(bbox * bbox_size) x 6, argmax class), a PageHeader / PageFooter whose polygon (surya/layout/util.py:4-40, float64)
 *           lies in the middle of its page takes the next-best class (__init__.py:158-169);
 *   table:  (trunc(clamp(bbox * bbox_size, 0, bbox_size)) x 6, argmax category, argmax merges, round(max(colspan, 1)), argmax is_header)
 *           = LabelShaper.dict_to_labels of the step's predictions (table_rec/shaper.py:12-51) --
 * embeds it and runs on; the host reads n_steps steps of records at a time and re-derives the tokens as a check.
 *   surya_layout_set_feedback: the constants of the rule (per model) and, for the layout family, the (width, height) of every row's page
 *     slice (host int32 [batch][2]; NULL = rule off). Call after surya_layout_encode / surya_layout_select, before the first run.
 *   surya_layout_decode_steps: enqueue n_steps (<= 16) steps at cache positions position .. position + n_steps - 1 into record ring
 *     `ring` (0 / 1). boxes = host int32 [batch][token width] feeds the first step from the host; NULL continues from the token the
 *     previous run left on the device (its position must be this run's `position`, else SA_ERR_STATE). Enqueue only.
 *   surya_layout_wait_steps: wait for ring `ring` and copy its records: class_logits fp32 [n_steps][batch][label_count], bbox fp32
 *     [n_steps][batch][6], fed_tokens int32 [n_steps][batch][token width] (the token each step fed to the one after it). */
typedef struct surya_layout_feedback {
    int32_t skew_scaler;              /* layout: decoder.skew_scaler (512) */
    int32_t relabel_ids[2];           /* layout: class ids (with the special-token offset) of PageHeader / PageFooter; -1 = none */
    int32_t head_widths[4];           /* table: rows of the category | merges | colspan | is_header heads (their sum = label_count) */
    const int32_t* page_sizes;        /* layout: host [batch][2] = (width, height) of each slice, NULL = no header / footer rule */
} surya_layout_feedback;
int surya_layout_set_feedback(surya_layout* h, const surya_layout_feedback* fb, int batch, void* stream);
int surya_layout_decode_steps(surya_layout* h, const int32_t* boxes, int batch, int position, int n_steps, int ring, void* stream);
int surya_layout_wait_steps(surya_layout* h, int ring, int batch, int n_steps, float* class_logits, float* bbox, int32_t* fed_tokens);
/* Re-batch the decoder after surya_layout_encode: the following decode steps run n rows (n <= max_batch), row i cross-attending the
 * encoder states of image src_index[i] (host array, values < the encoded batch). Table recognition decodes the cells of every detected
 * ROW against its table image (surya/table_rec/__init__.py:196-230: row_encoder_hidden_states = stacked copies); here the copies are an
 * index. Synchronous. */
int surya_layout_select(surya_layout* h, const int32_t* src_index, int n);
/* Test hook: encoder output [batch * tokens, hidden] of the last encode (device, compute dtype). */
int surya_layout_encoder_states(surya_layout* h, void* out, int batch, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Measurement support (bench.py `roofline`): when enabled every GEMM launch is bracketed by hipEvents on its own
 * stream. surya_prof_read syncs the device and returns, per bucket (0: 128x128 GEMM, 1: tall 256-row GEMM tiles, 2: smaller GEMM tiles,
 * 3: implicit-GEMM convolutions),
 * the number of launches, the summed event time (ms) and the summed ALGORITHMIC flops / bytes
 * (2MNK; X + W + C (+R) once each). Arrays need >= 4 entries. Not for use inside timed regions.
 * ---------------------------------------------------------------------------------------------------------- */
int surya_prof_enable(int on);
/* Launch-policy knob for A/B measurements inside one process (tools/microbench/decode_sweep.py; keys = the fields of
 * sa::Tuning in csrc/common.h: "graph", "split_target", "bigtile", "persist", "lmhead", "dattn", ...). Process-wide; results never depend on it
 * beyond fp rounding order. Returns SA_ERR_ARG for an unknown key. */
int surya_set_tuning(const char* key, int value);
int surya_prof_read(int max_cfg, int* launches, double* ms, double* flops, double* bytes);
/* surya_prof_read + slab_bytes[]: the fp32 partial slabs split-K launches wrote (a by-product of the kernel's own decomposition,
 * NOT part of `bytes`: `bytes` counts the result once in the storage type, as an unsplit GEMM would write it). slab_bytes may be NULL. */
int surya_prof_read2(int max_cfg, int* launches, double* ms, double* flops, double* bytes, double* slab_bytes);
/* Median event-pair time (ms) around an empty kernel on `stream`: the fixed cost inside every per-launch figure above. */
int surya_prof_event_overhead(void* stream, double* ms);

#ifdef __cplusplus
}
#endif
#endif /* SURYA_AMD_H */
