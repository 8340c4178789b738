"""Repack a SuryaModel state dict (reference parameter names) into the kernel layout of libsurya_amd.so.

One-time, at load (the reference does the equivalent `.to(device)` in RecognitionModelLoader.model,
surya/recognition/loader.py:25-58). Layout decisions (see include/surya_amd.h, SA_RW_*):
  * Linear weights stay [out, in] (K-contiguous "NT" GEMM operands); K padded with zeros to a multiple of 64
    (patch embed 588 -> 640, encoder down_proj 3420 -> 3456);
  * gate_proj / up_proj fused and row-INTERLEAVED (g0, u0, g1, u1, ...) so the SwiGLU epilogue finds a
    (gate, up) pair inside one lane's 4 output columns;
  * q/k/v projections of the decoder fused into one [ (nq + 2 nkv) d, H ] matrix;
  * RoPE inverse-frequency tables are computed here in fp32 exactly as the reference does
    (encoder/__init__.py:79, decoder rope init 1/theta^(2i/d)).
"""
from __future__ import annotations

from typing import Dict, List

import torch

from ..config import RecConfig
from .. import _lib as L


def pad64(n: int) -> int:
    return (n + 63) // 64 * 64


def _pad_k(w: torch.Tensor, k: int) -> torch.Tensor:
    if w.shape[1] == k:
        return w
    out = w.new_zeros((w.shape[0], k))
    out[:, : w.shape[1]] = w
    return out


def _interleave(g: torch.Tensor, u: torch.Tensor, rows_pad: int) -> torch.Tensor:
    """[I, ...] x2 -> [2*rows_pad, ...] with rows (g0, u0, g1, u1, ...), zero padded."""
    shape = (2 * rows_pad,) + tuple(g.shape[1:])
    out = g.new_zeros(shape)
    out[0: 2 * g.shape[0]: 2] = g
    out[1: 2 * u.shape[0]: 2] = u
    return out


def _pair_interleave_qk(t: torch.Tensor, heads: int, d: int) -> torch.Tensor:
    """Vision qkv projection rows [q | k | v], each [heads, d]: inside every q and k head move the rotate_half partners
    (j, j + d/2) next to each other (new row 2j = old j, 2j + 1 = old j + d/2) so the GEMM's EPI_ROPE epilogue finds a
    complete pair inside one lane's four consecutive output columns. q.k dot products are invariant to this common
    permutation of the head dim; v rows stay as they are."""
    hd = heads * d
    perm = torch.stack([torch.arange(d // 2), torch.arange(d // 2) + d // 2], 1).reshape(-1)        # [0, d/2, 1, d/2+1, ...]
    idx = torch.arange(3 * hd).reshape(3, heads, d)
    idx[:2] = idx[:2][..., perm]
    return t[idx.reshape(-1)]


def repack_rec_weights(cfg: RecConfig, sd: Dict[str, torch.Tensor], dtype: torch.dtype, device) -> List[torch.Tensor]:
    e, d = cfg.encoder, cfg.decoder
    He = e.hidden_size
    Ip, Kp = pad64(e.intermediate_size), pad64(e.patch_dim)
    Idp = pad64(d.intermediate_size)
    n = L.RW_GLOBALS + e.depth * L.RE_COUNT + d.num_hidden_layers * L.RD_COUNT
    out: List[torch.Tensor] = [None] * n

    def put(idx, t, dt=None):
        out[idx] = t.to(device=device, dtype=dt or dtype).contiguous()

    f = lambda k: sd[k].float()
    put(L.RW_PATCH, _pad_k(f("vision_encoder.patch_embed.proj.weight").reshape(He, -1), Kp))
    put(L.RW_MERGER_LN, f("vision_encoder.merger.ln_q.weight"))
    put(L.RW_FC1_W, f("vision_encoder.merger.mlp.0.weight"))
    put(L.RW_FC1_B, f("vision_encoder.merger.mlp.0.bias"))
    put(L.RW_FC2_W, f("vision_encoder.merger.mlp.2.weight"))
    put(L.RW_FC2_B, f("vision_encoder.merger.mlp.2.bias"))
    put(L.RW_IMG_H, f("img_h_embed.weight"))
    put(L.RW_IMG_W, f("img_w_embed.weight"))
    put(L.RW_DEC_NORM, f("decoder.norm.weight"))
    put(L.RW_TOK_EMBED, f("embedder.token_embed.weight"))
    # lm_head is tied to the token embedding in real checkpoints (common/surya/__init__.py:111-116)
    put(L.RW_LM_W, f("lm_head.weight") if "lm_head.weight" in sd else f("embedder.token_embed.weight"))
    put(L.RW_LM_B, f("lm_head.bias") if "lm_head.bias" in sd else torch.zeros(d.vocab_size))
    put(L.RW_BBOX_W, f("bbox_head.weight"))
    put(L.RW_BBOX_B, f("bbox_head.bias"))
    dim = e.head_dim // 2
    put(L.RW_ENC_INVFREQ, 1.0 / (10000.0 ** (torch.arange(0, dim, 2, dtype=torch.float) / dim)), torch.float32)
    put(L.RW_DEC_INVFREQ,
        1.0 / (d.rope_theta ** (torch.arange(0, d.head_dim, 2, dtype=torch.int64).float() / d.head_dim)), torch.float32)
    for l in range(e.depth):
        p = f"vision_encoder.blocks.{l}."
        b = L.RW_GLOBALS + l * L.RE_COUNT
        put(b + L.RE_NORM1, f(p + "norm1.weight"))
        put(b + L.RE_QKV_W, _pair_interleave_qk(f(p + "attn.qkv.weight"), e.num_heads, e.head_dim))
        put(b + L.RE_QKV_B, _pair_interleave_qk(f(p + "attn.qkv.bias"), e.num_heads, e.head_dim))
        put(b + L.RE_PROJ_W, f(p + "attn.proj.weight"))
        put(b + L.RE_PROJ_B, f(p + "attn.proj.bias"))
        put(b + L.RE_NORM2, f(p + "norm2.weight"))
        put(b + L.RE_GU_W, _interleave(f(p + "mlp.gate_proj.weight"), f(p + "mlp.up_proj.weight"), Ip))
        put(b + L.RE_GU_B, _interleave(f(p + "mlp.gate_proj.bias"), f(p + "mlp.up_proj.bias"), Ip))
        put(b + L.RE_DOWN_W, _pad_k(f(p + "mlp.down_proj.weight"), Ip))
        put(b + L.RE_DOWN_B, f(p + "mlp.down_proj.bias"))
    base = L.RW_GLOBALS + e.depth * L.RE_COUNT
    for l in range(d.num_hidden_layers):
        p = f"decoder.layers.{l}."
        b = base + l * L.RD_COUNT
        put(b + L.RD_LN1, f(p + "input_layernorm.weight"))
        put(b + L.RD_QKV_W, torch.cat([f(p + "self_attn.q_proj.weight"), f(p + "self_attn.k_proj.weight"),
                                       f(p + "self_attn.v_proj.weight")], 0))
        put(b + L.RD_QKV_B, torch.cat([f(p + "self_attn.q_proj.bias"), f(p + "self_attn.k_proj.bias"),
                                       f(p + "self_attn.v_proj.bias")], 0))
        put(b + L.RD_O_W, f(p + "self_attn.o_proj.weight"))
        put(b + L.RD_LN2, f(p + "post_attention_layernorm.weight"))
        put(b + L.RD_GU_W, _interleave(f(p + "mlp.gate_proj.weight"), f(p + "mlp.up_proj.weight"), Idp))
        put(b + L.RD_DOWN_W, _pad_k(f(p + "mlp.down_proj.weight"), Idp))
    return out
