"""CPU oracle for the recognition hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module. The product
(surya_amd/) never does; it fails loudly when the HIP extension is missing.

A plain-PyTorch fp32 restatement of what the reference (VikParuchuri/surya @ v0.14.6) computes for
RecognitionPredictor's model path. Every function cites the reference file:line it follows. Parity of this
restatement is PINNED against the real reference modules run in the build container through
oracle/ref_shim (see oracle/make_golden.py -> tests/golden/rec_*.pt and tests/test_oracle_vs_reference.py).
Real-weight behaviour (the reference's own "Hello World" tests) is unpinned: no checkpoints offline.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from surya_amd.config import RecConfig, EncoderConfig, DecoderConfig

SD = Dict[str, torch.Tensor]

# "eager" = the numerically explicit definition (encoder/__init__.py:202-263, decoder/__init__.py:101-128) used by every
# parity test; "sdpa" = the reference's default CPU implementation (encoder :266-411; decoder through
# ALL_ATTENTION_FUNCTIONS["sdpa"], decoder/__init__.py:220-222), <= 5e-7 apart in fp32 and what a user of the
# reference actually runs: bench.py's cpu_baseline times this one.
ATTN_IMPL = "eager"
# Emulation of the MXFP8 decode path (csrc/gemm_mx.h): when True, OracleRecModel.decode() quantise-dequantises both operands
# of the decoder projections and lm_head (oracle/mx_oracle.py; weights from their bf16 rounding, as the product does). There is
# no reference counterpart -- this pins the product's fp8 arithmetic to the published MX format, not to surya.
MX_DECODE = False
# Emulate the FP8 KV cache of the decode steps (csrc/decode_attn_kv8.h): every cached K / V row (one per token and kv head) passes
# through mx_oracle.kv8_quantize / kv8_dequantize before the decode steps' attention; prefill attends over unquantised rows.
KV8_DECODE = False
_MX_W = {}


def _lin(x, sd, key, bias_key=None, mx=False):
    w = sd[key]
    b = sd[bias_key] if bias_key else None
    if mx:
        from . import mx_oracle
        ck = (id(sd), key)
        if ck not in _MX_W:
            _MX_W[ck] = mx_oracle.fake_quant(w.bfloat16().float())
        return F.linear(mx_oracle.fake_quant(x), _MX_W[ck].to(x.dtype), b)
    return F.linear(x, w, b)


# ------------------------------------------------------------------------------------------ shared pieces
def rms_norm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    """Qwen2RMSNorm (encoder/__init__.py:99-104, decoder/__init__.py:250-255): fp32 math, cast back, THEN *weight."""
    dt = x.dtype
    xf = x.to(torch.float32)
    var = xf.pow(2).mean(-1, keepdim=True)
    xf = xf * torch.rsqrt(var + eps)
    return w * xf.to(dt)


def rotate_half(x: torch.Tensor) -> torch.Tensor:
    """encoder/__init__.py:181-185, decoder/__init__.py:53-57."""
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


# ------------------------------------------------------------------------------------------ vision encoder
def vision_pos_ids(grid_thw: Sequence[Tuple[int, int, int]], merge: int) -> torch.Tensor:
    """(h, w) patch coordinates in merge-block-major row order (encoder/__init__.py:523-546)."""
    out = []
    for t, h, w in grid_thw:
        hp = torch.arange(h).unsqueeze(1).expand(-1, w)
        hp = hp.reshape(h // merge, merge, w // merge, merge).permute(0, 2, 1, 3).flatten()
        wp = torch.arange(w).unsqueeze(0).expand(h, -1)
        wp = wp.reshape(h // merge, merge, w // merge, merge).permute(0, 2, 1, 3).flatten()
        out.append(torch.stack([hp, wp], dim=-1).repeat(t, 1))
    return torch.cat(out, dim=0)


def vision_rotary(grid_thw, head_dim: int, merge: int, theta: float = 10000.0) -> torch.Tensor:
    """rot_pos_emb (encoder/__init__.py:523-550) with the table of Qwen2_5_VisionRotaryEmbedding (:76-87):
    dim = head_dim // 2, inv_freq over arange(0, dim, 2); result [P, head_dim // 2] = [freq(h) | freq(w)]."""
    dim = head_dim // 2
    inv_freq = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float) / dim))
    pos = vision_pos_ids(grid_thw, merge)
    max_grid = max(max(h, w) for _, h, w in grid_thw)
    seq = torch.arange(max_grid, dtype=inv_freq.dtype)
    freqs = torch.outer(seq, inv_freq)
    return freqs[pos].flatten(1)


def window_index(grid_thw, window_size: int, merge: int, patch: int):
    """get_window_index (encoder/__init__.py:552-597). Returns (window_index over merged tokens,
    cu_window_seqlens in PATCH units, consecutive duplicates removed as at :620)."""
    index_list: List[torch.Tensor] = []
    cu = [0]
    base = 0
    vw = window_size // merge // patch
    unit = merge * merge
    for t, h, w in grid_thw:
        lh, lw = h // merge, w // merge
        index = torch.arange(t * lh * lw).reshape(t, lh, lw)
        pad_h = vw - lh % vw
        pad_w = vw - lw % vw
        nh, nw = (lh + pad_h) // vw, (lw + pad_w) // vw
        padded = F.pad(index, (0, pad_w, 0, pad_h), "constant", -100)
        padded = padded.reshape(t, nh, vw, nw, vw).permute(0, 1, 3, 2, 4).reshape(t, nh * nw, vw, vw)
        seqlens = (padded != -100).sum([2, 3]).reshape(-1)
        padded = padded.reshape(-1)
        index_list.append(padded[padded != -100] + base)
        cu.extend((seqlens.cumsum(0) * unit + cu[-1]).tolist())
        base += t * lh * lw
    cu_t = torch.unique_consecutive(torch.tensor(cu, dtype=torch.int32))
    return torch.cat(index_list, dim=0), cu_t


def _segment_attention(q, k, v, cu: Sequence[int], scale: float) -> torch.Tensor:
    """Block-diagonal non-causal attention, the semantics of the eager path (encoder/__init__.py:238-261):
    softmax in fp32 over each [cu[i-1], cu[i]) segment. q,k,v: [P, heads, d]."""
    out = torch.empty_like(q)
    for i in range(1, len(cu)):
        a, b = int(cu[i - 1]), int(cu[i])
        qs, ks, vs = (x[a:b].transpose(0, 1) for x in (q, k, v))          # [heads, L, d]
        if ATTN_IMPL == "sdpa":
            out[a:b] = F.scaled_dot_product_attention(qs[None], ks[None], vs[None], scale=scale)[0].transpose(0, 1)
            continue
        w = torch.matmul(qs, ks.transpose(1, 2)) * scale
        w = F.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
        out[a:b] = torch.matmul(w, vs).transpose(0, 1)
    return out


def encoder_forward(sd: SD, e: EncoderConfig, tiles: torch.Tensor, grid_thw, prefix="vision_encoder.",
                    taps: Optional[dict] = None) -> torch.Tensor:
    """Qwen2_5_VisionTransformerPretrainedModel.forward (encoder/__init__.py:599-672).
    tiles [P, C*T*ps*ps] -> [P / merge^2, out_hidden_size] in ORIGINAL (un-windowed) merged-token order."""
    grid_thw = [tuple(int(x) for x in g) for g in grid_thw]
    He, nh, hd = e.hidden_size, e.num_heads, e.head_dim
    unit = e.spatial_merge_size ** 2
    dt = sd[prefix + "patch_embed.proj.weight"].dtype
    # patch embed: Conv3d with kernel == stride (encoder/__init__.py:53-73) == one GEMM
    x = tiles.to(dt) @ sd[prefix + "patch_embed.proj.weight"].reshape(He, -1).t()
    rot = vision_rotary(grid_thw, hd, e.spatial_merge_size)
    widx, cu_win = window_index(grid_thw, e.window_size, e.spatial_merge_size, e.patch_size)
    P = x.shape[0]
    x = x.reshape(P // unit, unit, -1)[widx].reshape(P, -1)               # :622-627
    rot = rot.reshape(P // unit, unit, -1)[widx].reshape(P, -1)           # :628-632
    emb = torch.cat((rot, rot), dim=-1)                                   # :633
    cos, sin = emb.cos().unsqueeze(-2).float(), emb.sin().unsqueeze(-2).float()
    sizes = torch.tensor([h * w for t, h, w in grid_thw for _ in range(t)])
    cu_full = F.pad(sizes.cumsum(0), (1, 0)).tolist()                     # :636-646
    cu_win = cu_win.tolist()
    scale = 1.0 / math.sqrt(hd)
    for li in range(e.depth):
        p = f"{prefix}blocks.{li}."
        cu = cu_full if li in e.fullatt_block_indexes else cu_win        # :649-652
        h = rms_norm(x, sd[p + "norm1.weight"], e.rms_norm_eps)
        qkv = F.linear(h, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"])
        q, k, v = qkv.reshape(P, 3, nh, hd).permute(1, 0, 2, 3).unbind(0)  # :218-223
        qf, kf = q.float(), k.float()                                     # apply_rotary_pos_emb_vision :188-199
        q = (qf * cos + rotate_half(qf) * sin).to(q.dtype)
        k = (kf * cos + rotate_half(kf) * sin).to(k.dtype)
        a = _segment_attention(q, k, v, cu, scale).reshape(P, -1)
        x = x + F.linear(a, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
        h = rms_norm(x, sd[p + "norm2.weight"], e.rms_norm_eps)
        g = F.linear(h, sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.gate_proj.bias"])
        u = F.linear(h, sd[p + "mlp.up_proj.weight"], sd[p + "mlp.up_proj.bias"])
        x = x + F.linear(F.silu(g) * u, sd[p + "mlp.down_proj.weight"], sd[p + "mlp.down_proj.bias"])
        if taps is not None:
            taps[f"enc_block{li}"] = x.clone()
    # merger (encoder/__init__.py:110-123): RMSNorm per patch, group 4 patches, Linear+GELU(erf)+Linear
    h = rms_norm(x, sd[prefix + "merger.ln_q.weight"], 1e-6).view(-1, He * unit)
    h = F.gelu(F.linear(h, sd[prefix + "merger.mlp.0.weight"], sd[prefix + "merger.mlp.0.bias"]))
    h = F.linear(h, sd[prefix + "merger.mlp.2.weight"], sd[prefix + "merger.mlp.2.bias"])
    return h[torch.argsort(widx)]                                         # :669-670


def learned_2d_embeddings(sd: SD, cfg: RecConfig, grid_thw) -> torch.Tensor:
    """get_2d_learned_embeddings (common/surya/__init__.py:233-272): float division then truncation."""
    out = []
    m = cfg.encoder.spatial_merge_size
    mult = cfg.image_embed_encoding_multiplier
    for t, gh, gw in grid_thw:
        lh, lw = int(gh) // m, int(gw) // m
        ih = (torch.arange(lh) / max(1, lh - 1) * mult).to(torch.long)
        iw = (torch.arange(lw) / max(1, lw - 1) * mult).to(torch.long)
        full = sd["img_h_embed.weight"][ih][:, None] + sd["img_w_embed.weight"][iw][None, :]
        out.append(full.flatten(0, 1))
    return torch.cat(out, dim=0)


def image_embeddings(sd: SD, cfg: RecConfig, tiles, grid_thw, taps=None) -> torch.Tensor:
    """get_image_embeddings (common/surya/__init__.py:130-195). Encoder chunking (:137-170) is numerically
    transparent (per-image attention) so the oracle encodes the whole packed batch at once."""
    emb = encoder_forward(sd, cfg.encoder, tiles, grid_thw, taps=taps)
    return emb + learned_2d_embeddings(sd, cfg, grid_thw)


# ------------------------------------------------------------------------------------------------- decoder
def decoder_rope(position_ids: torch.Tensor, d: DecoderConfig, dtype) -> Tuple[torch.Tensor, torch.Tensor]:
    """Qwen2RotaryEmbedding.forward (decoder/__init__.py:346-361): fp32 freqs, cos/sin cast to model dtype."""
    inv_freq = 1.0 / (d.rope_theta ** (torch.arange(0, d.head_dim, 2, dtype=torch.int64).float() / d.head_dim))
    freqs = (inv_freq[None, :, None].expand(position_ids.shape[0], -1, 1) @
             position_ids[:, None, :].float()).transpose(1, 2)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def causal_mask_4d(attention_mask: torch.Tensor, q_len: int, past_len: int, dtype) -> torch.Tensor:
    """_prepare_4d_causal_attention_mask_with_cache_position (decoder/__init__.py:554-631) for a 2-D mask
    [B, past_len + q_len]: additive finfo.min where (key index > cache_position) or key is padding."""
    B, T = attention_mask.shape
    mn = torch.finfo(dtype).min
    cache_pos = torch.arange(past_len, past_len + q_len)
    m = torch.full((q_len, T), mn, dtype=dtype)
    m = m * (torch.arange(T) > cache_pos.reshape(-1, 1))
    m = m[None, None].expand(B, 1, -1, -1).clone()
    pad = (m + attention_mask[:, None, None, :].to(dtype)) == 0
    return m.masked_fill(pad, mn)


class OracleKV:
    """DynamicCache semantics (decoder/__init__.py:190-195): per layer concat along time."""

    def __init__(self, n_layers: int):
        self.k: List[Optional[torch.Tensor]] = [None] * n_layers
        self.v: List[Optional[torch.Tensor]] = [None] * n_layers

    def length(self) -> int:
        return 0 if self.k[0] is None else self.k[0].shape[2]

    def update(self, li, k, v):
        if self.k[li] is None:
            self.k[li], self.v[li] = k, v
        else:
            self.k[li] = torch.cat([self.k[li], k], dim=2)
            self.v[li] = torch.cat([self.v[li], v], dim=2)
        return self.k[li], self.v[li]


def decoder_forward(sd: SD, d: DecoderConfig, x: torch.Tensor, attention_mask, position_ids, cache: OracleKV,
                    taps: Optional[dict] = None, mx: bool = False, kv8: bool = False) -> torch.Tensor:
    """SuryaDecoderModel.forward (decoder/__init__.py:417-490) with the eager attention of :101-128."""
    B, S, H = x.shape
    nq, nkv, hd = d.num_attention_heads, d.num_key_value_heads, d.head_dim
    past = cache.length()
    mask = causal_mask_4d(attention_mask, S, past, x.dtype)
    cos, sin = decoder_rope(position_ids, d, x.dtype)
    cos, sin = cos.unsqueeze(1), sin.unsqueeze(1)
    scaling = hd ** -0.5
    for li in range(d.num_hidden_layers):
        p = f"decoder.layers.{li}."
        h = rms_norm(x, sd[p + "input_layernorm.weight"], d.rms_norm_eps)
        q = _lin(h, sd, p + "self_attn.q_proj.weight", p + "self_attn.q_proj.bias", mx).view(B, S, nq, hd).transpose(1, 2)
        k = _lin(h, sd, p + "self_attn.k_proj.weight", p + "self_attn.k_proj.bias", mx).view(B, S, nkv, hd).transpose(1, 2)
        v = _lin(h, sd, p + "self_attn.v_proj.weight", p + "self_attn.v_proj.bias", mx).view(B, S, nkv, hd).transpose(1, 2)
        q = q * cos + rotate_half(q) * sin                                # apply_rotary_pos_emb :60-84
        k = k * cos + rotate_half(k) * sin
        k, v = cache.update(li, k, v)
        if kv8:
            from oracle import mx_oracle as _mo
            k = torch.from_numpy(_mo.kv8_dequantize(*_mo.kv8_quantize(k.float().numpy()))).to(k.dtype)
            v = torch.from_numpy(_mo.kv8_dequantize(*_mo.kv8_quantize(v.float().numpy()))).to(v.dtype)
        g = nq // nkv
        kk = k[:, :, None].expand(B, nkv, g, -1, hd).reshape(B, nq, -1, hd)   # repeat_kv :87-98
        vv = v[:, :, None].expand(B, nkv, g, -1, hd).reshape(B, nq, -1, hd)
        if ATTN_IMPL == "sdpa":
            a = F.scaled_dot_product_attention(q, kk, vv, attn_mask=mask[:, :, :, : kk.shape[-2]], scale=scaling)
            a = a.transpose(1, 2).reshape(B, S, -1)
        else:
            w = torch.matmul(q, kk.transpose(2, 3)) * scaling + mask[:, :, :, : kk.shape[-2]]
            w = F.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
            a = torch.matmul(w, vv).transpose(1, 2).reshape(B, S, -1)
        x = x + _lin(a, sd, p + "self_attn.o_proj.weight", None, mx)
        h = rms_norm(x, sd[p + "post_attention_layernorm.weight"], d.rms_norm_eps)
        h = F.silu(_lin(h, sd, p + "mlp.gate_proj.weight", None, mx)) * _lin(h, sd, p + "mlp.up_proj.weight", None, mx)
        x = x + _lin(h, sd, p + "mlp.down_proj.weight", None, mx)
        if taps is not None:
            taps[f"dec_layer{li}"] = x.clone()
    return rms_norm(x, sd["decoder.norm.weight"], d.rms_norm_eps)


# ------------------------------------------------------------------------------------------------- model
IMAGE_TOKEN_ID_DEFAULT = 3


class OracleRecModel:
    """SuryaModel.forward (common/surya/__init__.py:274-338) as two entry points sharing one KV cache."""

    def __init__(self, cfg: RecConfig, sd: SD, image_token_id: int):
        self.cfg, self.sd, self.image_token_id = cfg, sd, image_token_id
        self.cache: Optional[OracleKV] = None

    def embed(self, input_ids, tiles, grid_thw, taps=None):
        """embed_ids_boxes_images (:197-231): token embedding, then masked_scatter of image features."""
        x = self.sd["embedder.token_embed.weight"][input_ids]
        if tiles is not None:
            feats = image_embeddings(self.sd, self.cfg, tiles, grid_thw, taps=taps)
            if taps is not None:
                taps["image_embeddings"] = feats.clone()
            m = (input_ids == self.image_token_id).unsqueeze(-1).expand_as(x)
            x = x.masked_scatter(m, feats.to(x.dtype))
        return x

    def heads(self, hidden_last: torch.Tensor, mx: bool = False):
        """:323-330 with logits_to_keep=1."""
        bbox = torch.sigmoid(F.linear(hidden_last, self.sd["bbox_head.weight"], self.sd["bbox_head.bias"]))
        lm = _lin(hidden_last, self.sd, "lm_head.weight", "lm_head.bias", mx)
        return lm, bbox

    @torch.inference_mode()
    def prefill(self, input_ids, tiles, grid_thw, attention_mask, position_ids, taps=None):
        self.cache = OracleKV(self.cfg.decoder.num_hidden_layers)
        x = self.embed(input_ids, tiles, grid_thw, taps)
        h = decoder_forward(self.sd, self.cfg.decoder, x, attention_mask, position_ids, self.cache, taps)
        return self.heads(h[:, -1:, :])

    @torch.inference_mode()
    def decode(self, input_ids, attention_mask, position_ids):
        x = self.sd["embedder.token_embed.weight"][input_ids]
        h = decoder_forward(self.sd, self.cfg.decoder, x, attention_mask, position_ids, self.cache, mx=MX_DECODE, kv8=KV8_DECODE)
        return self.heads(h[:, -1:, :], mx=MX_DECODE)


def process_outputs(lm_logits, bbox_logits, eos_id: int, pad_id: int, bbox_size: int):
    """RecognitionPredictor.process_outputs (recognition/__init__.py:294-324)."""
    logits = lm_logits[:, -1:, :].clone().float()
    bl = bbox_logits[:, -1:, :].clone().float()
    preds = torch.argmax(logits, dim=-1)
    done = ((preds == eos_id) | (preds == pad_id)).squeeze(-1)
    next_ids = torch.where(done.unsqueeze(1), torch.tensor(pad_id), preds).to(torch.long)
    scores = torch.max(F.softmax(logits[:, -1], dim=-1), dim=-1).values
    scores = scores.masked_fill(done, 0).unsqueeze(1)
    boxes = (bl * bbox_size).to(torch.long)
    process_outputs.last_raw_boxes = bl * bbox_size      # un-truncated values, for boundary-aware test asserts
    return next_ids, preds, boxes, done, scores


def detect_repeat_token(tokens: List[int], max_repeats: int = 40) -> bool:
    """recognition/util.py:59-69."""
    if len(tokens) < max_repeats:
        return False
    last_n = tokens[-max_repeats:]
    u = len(set(last_n))
    if u > 5:
        return False
    return last_n[-u:] == last_n[-u * 2: -u]


@torch.inference_mode()
def generate(model: OracleRecModel, input_ids, tiles, grid_thw, attention_mask, position_ids, max_tokens: int,
             eos_id: int, pad_id: int, nop_id: int, record_logits: bool = False):
    """Static-batch greedy loop with the reference's stop rules (recognition/__init__.py:539-595).
    Per-line token streams do not depend on batch composition (SURVEY 8(d) probe), so this reproduces what the
    reference's continuous-batching loop emits for each line."""
    B = input_ids.shape[0]
    tokens: List[List[int]] = [[] for _ in range(B)]
    scores: List[List[float]] = [[] for _ in range(B)]
    boxes: List[List[List[int]]] = [[] for _ in range(B)]
    raw_boxes: List[List[List[float]]] = [[] for _ in range(B)]
    logits_log = [] if record_logits else None
    active = [True] * B
    lm, bb = model.prefill(input_ids, tiles, grid_thw, attention_mask, position_ids)
    nxt, preds, bx, done, sc = process_outputs(lm, bb, eos_id, pad_id, model.cfg.bbox_size)
    if record_logits:
        logits_log.append(lm[:, -1].clone())
    for b in range(B):
        t = int(preds[b, 0])
        tokens[b].append(t); scores[b].append(float(sc[b, 0])); boxes[b].append(bx[b, 0].tolist())
        raw_boxes[b].append(process_outputs.last_raw_boxes[b, 0].tolist())
        if t in (eos_id, nop_id):                                          # prefill stop rule :559-563
            active[b] = False
    attention_mask = F.pad(attention_mask, (0, 1), value=1)
    position_ids = position_ids[:, -1:] + 1
    while any(active):
        lm, bb = model.decode(nxt, attention_mask, position_ids)
        nxt, preds, bx, done, sc = process_outputs(lm, bb, eos_id, pad_id, model.cfg.bbox_size)
        if record_logits:
            logits_log.append(lm[:, -1].clone())
        for b in range(B):
            if not active[b]:
                continue
            t = int(preds[b, 0])
            tokens[b].append(t); scores[b].append(float(sc[b, 0])); boxes[b].append(bx[b, 0].tolist())
            raw_boxes[b].append(process_outputs.last_raw_boxes[b, 0].tolist())
            rep = len(tokens[b]) >= max_tokens or detect_repeat_token(tokens[b])   # :583-595
            if t in (eos_id, pad_id) or rep:
                active[b] = False
        attention_mask = F.pad(attention_mask, (0, 1), value=1)
        position_ids = position_ids[:, -1:] + 1
    generate.last_raw_boxes = raw_boxes
    return tokens, boxes, scores, logits_log


@torch.inference_mode()
def teacher_forced_logits(model: OracleRecModel, input_ids, tiles, grid_thw, attention_mask, position_ids,
                          forced: List[List[int]], pad_id: int):
    """Logits of every step when the given token streams are fed back (lines shorter than the longest are fed pad).
    Used to compare reduced-precision runs position by position without error feedback through argmax."""
    out = []
    lm, _ = model.prefill(input_ids, tiles, grid_thw, attention_mask, position_ids)
    out.append(lm[:, -1].float().clone())
    steps = max(len(f) for f in forced)
    for s in range(steps - 1):
        nxt = torch.tensor([[f[s] if s < len(f) else pad_id] for f in forced], dtype=torch.long)
        attention_mask = F.pad(attention_mask, (0, 1), value=1)
        position_ids = position_ids[:, -1:] + 1
        lm, _ = model.decode(nxt, attention_mask, position_ids)
        out.append(lm[:, -1].float().clone())
    return out
