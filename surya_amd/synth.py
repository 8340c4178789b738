"""Seeded synthetic weights and inputs (no checkpoints or datasets exist offline).

State-dict keys are the reference's own parameter names (``SuryaModel.state_dict()`` /
``EfficientViTForSemanticSegmentation.state_dict()``) so a real safetensors checkpoint can be dropped in.

Init recipes are the probe-verified ones from SURVEY.md 8(d): HF-style N(0, 0.02) makes the tied-embedding
recogniser repeat its last token and the detector emit a constant 0.5 map, so neither would exercise parity.
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import numpy as np
import torch

from .config import RecConfig, DetConfig


def _normal(gen, shape, std):
    return torch.randn(shape, generator=gen, dtype=torch.float32) * std


def make_rec_weights(cfg: RecConfig, seed: int = 0, recipe: str = "default") -> Dict[str, torch.Tensor]:
    """fp32 CPU state dict for SuryaModel (surya/common/surya/__init__.py:71-109 lists the sub-modules).
    recipe "conditioned": see make_rec_weights_conditioned."""
    if recipe == "conditioned":
        return make_rec_weights_conditioned(cfg, seed)
    assert recipe == "default", recipe
    g = torch.Generator().manual_seed(seed)
    e, d = cfg.encoder, cfg.decoder
    sd: Dict[str, torch.Tensor] = {}
    He, Hd = e.hidden_size, d.hidden_size
    se, sdd = 1.7 / math.sqrt(He), 1.7 / math.sqrt(Hd)
    sd["vision_encoder.patch_embed.proj.weight"] = _normal(
        g, (He, e.in_channels, e.temporal_patch_size, e.patch_size, e.patch_size), 1.7 / math.sqrt(e.patch_dim))
    for i in range(e.depth):
        p = f"vision_encoder.blocks.{i}."
        sd[p + "norm1.weight"] = torch.ones(He)
        sd[p + "norm2.weight"] = torch.ones(He)
        sd[p + "attn.qkv.weight"] = _normal(g, (3 * He, He), se)
        sd[p + "attn.qkv.bias"] = _normal(g, (3 * He,), 0.02)
        sd[p + "attn.proj.weight"] = _normal(g, (He, He), se)
        sd[p + "attn.proj.bias"] = _normal(g, (He,), 0.02)
        for nm, shp in (("gate_proj", (e.intermediate_size, He)), ("up_proj", (e.intermediate_size, He)),
                        ("down_proj", (He, e.intermediate_size))):
            sd[p + f"mlp.{nm}.weight"] = _normal(g, shp, 1.7 / math.sqrt(shp[1]))
            sd[p + f"mlp.{nm}.bias"] = _normal(g, (shp[0],), 0.02)
    m = e.spatial_merge_size ** 2
    sd["vision_encoder.merger.ln_q.weight"] = torch.ones(He)
    sd["vision_encoder.merger.mlp.0.weight"] = _normal(g, (He * m, He * m), 1.7 / math.sqrt(He * m))
    sd["vision_encoder.merger.mlp.0.bias"] = _normal(g, (He * m,), 0.02)
    sd["vision_encoder.merger.mlp.2.weight"] = _normal(g, (e.out_hidden_size, He * m), 1.7 / math.sqrt(He * m))
    sd["vision_encoder.merger.mlp.2.bias"] = _normal(g, (e.out_hidden_size,), 0.02)
    qd, kvd = d.num_attention_heads * d.head_dim, d.num_key_value_heads * d.head_dim
    for i in range(d.num_hidden_layers):
        p = f"decoder.layers.{i}."
        sd[p + "self_attn.q_proj.weight"] = _normal(g, (qd, Hd), sdd)
        sd[p + "self_attn.q_proj.bias"] = _normal(g, (qd,), 0.02)
        sd[p + "self_attn.k_proj.weight"] = _normal(g, (kvd, Hd), sdd)
        sd[p + "self_attn.k_proj.bias"] = _normal(g, (kvd,), 0.02)
        sd[p + "self_attn.v_proj.weight"] = _normal(g, (kvd, Hd), sdd)
        sd[p + "self_attn.v_proj.bias"] = _normal(g, (kvd,), 0.02)
        sd[p + "self_attn.o_proj.weight"] = _normal(g, (Hd, qd), 1.7 / math.sqrt(qd))
        sd[p + "mlp.gate_proj.weight"] = _normal(g, (d.intermediate_size, Hd), sdd)
        sd[p + "mlp.up_proj.weight"] = _normal(g, (d.intermediate_size, Hd), sdd)
        sd[p + "mlp.down_proj.weight"] = _normal(g, (Hd, d.intermediate_size), 1.7 / math.sqrt(d.intermediate_size))
        sd[p + "input_layernorm.weight"] = torch.ones(Hd)
        sd[p + "post_attention_layernorm.weight"] = torch.ones(Hd)
    sd["decoder.norm.weight"] = torch.ones(Hd)
    sd["embedder.token_embed.weight"] = _normal(g, (d.vocab_size, Hd), 0.02)
    sd["img_w_embed.weight"] = _normal(g, (cfg.image_embed_encoding_size, Hd), 0.02)
    sd["img_h_embed.weight"] = _normal(g, (cfg.image_embed_encoding_size, Hd), 0.02)
    sd["bbox_head.weight"] = _normal(g, (6, Hd), sdd)
    sd["bbox_head.bias"] = _normal(g, (6,), 0.02)
    # untied, separately seeded head: with a tied random embedding the model only repeats its input
    sd["lm_head.weight"] = _normal(g, (d.vocab_size, Hd), 0.2)
    sd["lm_head.bias"] = _normal(g, (d.vocab_size,), 0.6)
    return sd


def make_rec_weights_conditioned(cfg: RecConfig, seed: int = 0, embed_rms: float = 2.0, q_gain: float = 0.35, beta: float = 10.0, r_gain: float = 8.0,
                                 top_logit: float = 16.0) -> Dict[str, torch.Tensor]:
    """A second seeded weight set on which reduced precision is a small perturbation, for parity tests of the bf16 path that can fail.

    The default recipe (gain 1.7 in every linear layer, residual branches as loud as the stream they add to, attention logits of
    std ~3) amplifies every rounding: the REFERENCE's own bf16 run sits 12-21 % of max|logit| away from its fp32 run and no argmax
    survives a tolerance built on that. Here
      * every residual-branch output projection (encoder attn.proj / mlp.down_proj, decoder o_proj / down_proj) is scaled by
        1 / sqrt(2 L) (L = depth of its stack), the query projections by `q_gain` (attention logits of std ~1 instead of ~3) and the
        token embedding has rms `embed_rms`, so the residual stream is a sum of comparable, non-amplifying contributions;
      * the (untied) lm_head is built so that a greedy step is CONFIDENT and still depends on what the network computed: with e_t the
        unit direction of the current token's embedding and r one fixed random unit direction,
            row[pi_plus(t)] += a e_t + b r,      row[pi_minus(t)] += a e_t - b r            (b = beta a)
        (pi_plus / pi_minus: seeded maps of the tokens that can occur into the even / odd half of the printable UTF-16 range), every
        other row is near zero. The current token's two rows win by a e_t . h; which of the two wins is the sign of r . h -- a
        function of the image, the attention and every layer. Top-2 margin = min(2 b |r . h|, ~a e_t . h): large except where
        r . h ~ 0.
    Nothing maps to </S> or <PAD>, so every line runs to its token limit."""
    sd = make_rec_weights(cfg, seed + 1000)
    e, d = cfg.encoder, cfg.decoder
    g = torch.Generator().manual_seed(seed + 2000)
    se, sdec = 1.0 / math.sqrt(2 * e.depth), 1.0 / math.sqrt(2 * d.num_hidden_layers)
    for i in range(e.depth):
        p = f"vision_encoder.blocks.{i}."
        for nm in ("attn.proj", "mlp.down_proj"):
            sd[p + nm + ".weight"] *= se
            sd[p + nm + ".bias"] *= se
        sd[p + "attn.qkv.weight"][: e.hidden_size] *= q_gain
        sd[p + "attn.qkv.bias"][: e.hidden_size] *= q_gain
    for i in range(d.num_hidden_layers):
        p = f"decoder.layers.{i}."
        sd[p + "self_attn.o_proj.weight"] *= sdec
        sd[p + "mlp.down_proj.weight"] *= sdec
        sd[p + "self_attn.q_proj.weight"] *= q_gain
        sd[p + "self_attn.q_proj.bias"] *= q_gain
    V, Hd = d.vocab_size, d.hidden_size
    r = _normal(g, (Hd,), 1.0)
    r = r / r.norm()
    for i in range(d.num_hidden_layers):     # the MLPs write nothing along r either: only attention (what a step READS from the image
        w = sd[f"decoder.layers.{i}.mlp.down_proj.weight"]          # and the earlier tokens) decides the sign of r . h
        w -= r[:, None] * (r @ w)[None, :]
        w = sd[f"decoder.layers.{i}.self_attn.o_proj.weight"]       # ... and attention writes along r with gain r_gain, so that
        w += (r_gain - 1.0) * r[:, None] * (r @ w)[None, :]         # r . h is O(1), far above the rounding noise of the stream
    emb = _normal(g, (V, Hd), embed_rms)
    emb -= (emb @ r)[:, None] * r            # embeddings carry nothing along r: r . h comes from the attention / MLP branches only
    sd["embedder.token_embed.weight"] = emb
    ehat = emb / emb.norm(dim=1, keepdim=True)
    off = cfg.special_token_offset
    lo, hi = off + 0x20, min(off + 0xD800, V)                   # printable BMP units below the surrogates
    if hi - lo < 64:                                            # tiny vocabularies (REC-TINY): whatever UTF-16 units exist
        lo, hi = off, V
    live = torch.arange(lo, hi)
    plus, minus = live[0::2], live[1::2]
    # tokens that can be the CURRENT token of a step: the live ones and the tags a prompt can end with
    dom = torch.cat([torch.arange(cfg.qwen_offset, off), live])
    pi_p = plus[torch.randperm(len(dom), generator=g) % len(plus)]
    pi_m = minus[torch.randperm(len(dom), generator=g) % len(minus)]
    # the final-norm hidden state has unit rms; its projection on e_t was measured at ~0.6 sqrt(Hd) for embed_rms = 2 (REC-FULL)
    a = top_logit / (0.6 * math.sqrt(Hd))
    W = _normal(g, (V, Hd), 0.003)
    W.index_add_(0, pi_p, a * ehat[dom])
    W.index_add_(0, pi_m, a * ehat[dom])
    W[plus] += beta * a * r
    W[minus] -= beta * a * r
    sd["lm_head.weight"] = W
    sd["lm_head.bias"] = _normal(g, (V,), 0.05)
    make_rec_weights_conditioned.aux = {"r": r, "pi_plus": pi_p, "pi_minus": pi_m, "domain": dom, "a": a}    # diagnostics only
    return sd


# ----------------------------------------------------------------------------------------------- detector
def det_param_specs(cfg: DetConfig) -> List[Tuple[str, str, tuple]]:
    """(name, kind, shape) for every parameter/buffer of EfficientViTForSemanticSegmentation in module order
    (surya/detection/model/encoderdecoder.py:484-630 backbone, :673-722 head). kind in {conv, bias, bn}."""
    specs: List[Tuple[str, str, tuple]] = []
    w = cfg.widths

    def conv(prefix, cout, cin_per_group, k, bias, bn):
        specs.append((prefix + ".conv.weight", "conv", (cout, cin_per_group, k, k)))
        if bias:
            specs.append((prefix + ".conv.bias", "bias", (cout,)))
        if bn:
            specs.append((prefix + ".norm", "bn", (cout,)))

    conv("vit.stem.in_conv", w[0], cfg.num_channels, 3, False, True)
    for r in range(cfg.depths[0]):
        conv(f"vit.stem.res{r}.main.conv1", w[0], w[0], 3, False, True)
        conv(f"vit.stem.res{r}.main.conv2", w[0], w[0], 3, False, True)
    cin = w[0]
    for si, (cout, depth) in enumerate(zip(w[1:], cfg.depths[1:])):
        vit_stage, fewer = si >= 3, si >= 2
        blk = f"vit.stages.{si}.blocks.0.main"
        exp = 24 if vit_stage else 16
        mid = round(cin * exp)
        if fewer:  # MBConv, bias+no-norm on first two convs
            conv(blk + ".inverted_conv", mid, cin, 1, True, False)
            conv(blk + ".depth_conv", mid, 1, 3, True, False)
            conv(blk + ".point_conv", cout, mid, 1, False, True)
        else:      # FusedMBConv
            conv(blk + ".spatial_conv", mid, cin, 3, False, True)
            conv(blk + ".point_conv", cout, mid, 1, False, True)
        for bi in range(1, depth + 1):
            if vit_stage:
                ctx = f"vit.stages.{si}.blocks.{bi}.context_module.main"
                td = cout  # heads*dim == in_channels (heads_ratio 1)
                heads = cout // cfg.head_dim
                conv(ctx + ".qkv", 3 * td, cout, 1, False, False)
                specs.append((ctx + ".aggreg.0.0.weight", "conv", (3 * td, 1, 5, 5)))
                specs.append((ctx + ".aggreg.0.1.weight", "conv", (3 * td, cfg.head_dim, 1, 1)))
                conv(ctx + ".proj", cout, 2 * td, 1, False, True)
                loc = f"vit.stages.{si}.blocks.{bi}.local_module.main"
                mid = cout * 6
                conv(loc + ".inverted_conv", mid, cout, 1, True, False)
                conv(loc + ".depth_conv", mid, 1, 3, True, False)
                conv(loc + ".point_conv", cout, mid, 1, False, True)
            else:
                blk = f"vit.stages.{si}.blocks.{bi}.main"
                mid = cout * 4
                if fewer:
                    conv(blk + ".inverted_conv", mid, cout, 1, True, False)
                    conv(blk + ".depth_conv", mid, 1, 3, True, False)
                    conv(blk + ".point_conv", cout, mid, 1, False, True)
                else:
                    conv(blk + ".spatial_conv", mid, cout, 3, False, True)
                    conv(blk + ".point_conv", cout, mid, 1, False, True)
        cin = cout
    for i, width in enumerate(w[1:]):
        specs.append((f"decode_head.linear_c.{i}.proj.weight", "conv", (cfg.decoder_layer_hidden_size, width)))
        specs.append((f"decode_head.linear_c.{i}.proj.bias", "bias", (cfg.decoder_layer_hidden_size,)))
    nst = len(w) - 1
    specs.append(("decode_head.linear_fuse.weight", "conv",
                  (cfg.decoder_hidden_size, cfg.decoder_layer_hidden_size * nst, 1, 1)))
    specs.append(("decode_head.batch_norm", "bn", (cfg.decoder_hidden_size,)))
    specs.append(("decode_head.classifier.weight", "conv", (cfg.num_labels, cfg.decoder_hidden_size, 1, 1)))
    specs.append(("decode_head.classifier.bias", "bias", (cfg.num_labels,)))
    return specs


def make_det_weights(cfg: DetConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    for name, kind, shape in det_param_specs(cfg):
        if kind == "conv":
            fan_in = int(np.prod(shape[1:]))
            sd[name] = _normal(g, shape, 1.0 / math.sqrt(fan_in))
        elif kind == "bias":
            sd[name] = _normal(g, shape, 0.1)
        else:  # bn: gamma U(0.8,1.2), beta N(0,0.1), running mean N(0,0.1), running var U(0.5,1.5)
            sd[name + ".weight"] = 0.8 + 0.4 * torch.rand(shape, generator=g)
            sd[name + ".bias"] = _normal(g, shape, 0.1)
            sd[name + ".running_mean"] = _normal(g, shape, 0.1)
            sd[name + ".running_var"] = 0.5 + torch.rand(shape, generator=g)
    return sd


# ------------------------------------------------------------------------------------------------- inputs
def make_line_crops(n: int, height: int = 64, width_range=(128, 512), seed: int = 1234, fixed_width=None):
    """uint8 RGB line crops: white background with dark random-width bars."""
    rng = np.random.default_rng(seed)
    crops = []
    for _ in range(n):
        w = int(fixed_width) if fixed_width else int(rng.integers(width_range[0], width_range[1] + 1))
        img = np.full((height, w, 3), 255, np.uint8)
        x = int(rng.integers(2, 8))
        while x < w - 4:
            bw = int(rng.integers(2, 12))
            y0 = int(rng.integers(4, height // 3))
            y1 = int(rng.integers(2 * height // 3, height - 4))
            img[y0:y1, x:min(w, x + bw)] = rng.integers(0, 96, size=3, dtype=np.uint8)
            x += bw + int(rng.integers(2, 10))
        crops.append(img)
    return crops


def make_pages(n: int, size: int = 1024, seed: int = 1234):
    """uint8 RGB pages: white with dark text-like rectangles in rows."""
    rng = np.random.default_rng(seed)
    pages = []
    for _ in range(n):
        img = np.full((size, size, 3), 255, np.uint8)
        y = int(rng.integers(20, 60))
        while y < size - 40:
            lh = int(rng.integers(14, 28))
            x = int(rng.integers(30, 90))
            xe = size - int(rng.integers(30, 300))
            while x < xe:
                ww = int(rng.integers(10, 70))
                img[y:y + lh, x:min(xe, x + ww)] = rng.integers(0, 80, size=3, dtype=np.uint8)
                x += ww + int(rng.integers(6, 16))
            y += lh + int(rng.integers(12, 36))
        pages.append(img)
    return pages


def make_pages_with_lines(n: int, size: int = 1024, seed: int = 1234):
    """make_pages plus, per page, the bounding boxes [x0, y0, x1, y1] of the text rows that were drawn (4 px margin):
    the line geometry a trained detector would return for these pages (a randomly initialised one finds ~1 box per page)."""
    rng = np.random.default_rng(seed)
    pages, boxes = [], []
    for _ in range(n):
        img = np.full((size, size, 3), 255, np.uint8)
        rows = []
        y = int(rng.integers(20, 60))
        while y < size - 40:
            lh = int(rng.integers(14, 28))
            x = int(rng.integers(30, 90))
            xs = x
            xe = size - int(rng.integers(30, 300))
            last = x
            while x < xe:
                ww = int(rng.integers(10, 70))
                img[y:y + lh, x:min(xe, x + ww)] = rng.integers(0, 80, size=3, dtype=np.uint8)
                last = min(xe, x + ww)
                x += ww + int(rng.integers(6, 16))
            rows.append([max(xs - 4, 0), max(y - 4, 0), min(last + 4, size), min(y + lh + 4, size)])
            y += lh + int(rng.integers(12, 36))
        pages.append(img)
        boxes.append(rows)
    return pages, boxes


def text_like_map(h: int, w: int, n_lines: int, seed: int = 0) -> np.ndarray:
    """A text heat map as a trained detector draws it: `n_lines` soft line-shaped blobs (height 8-20 px, length 6-60 % of the page
    width, a few slightly rotated) laid out in rows over a low noise floor, values in (0, 1). float32 [h, w].
    Used to time / test surya_det_boxes on realistic component counts (a randomly initialised detector yields one page-sized blob)."""
    rng = np.random.default_rng(seed)
    m = rng.random((h, w), dtype=np.float32) * 0.1
    per_row = max(1, int(np.ceil(n_lines / max(1, (h - 40) // 30))))
    n_rows = int(np.ceil(n_lines / per_row))
    pitch = (h - 40) / n_rows
    done = 0
    for r in range(n_rows):
        cy = 20 + (r + 0.5) * pitch
        x = rng.uniform(10, 40)
        for _ in range(per_row):
            if done == n_lines:
                break
            half_len = rng.uniform(0.03, 0.3) * w / per_row
            half_h = rng.uniform(4, min(10, pitch / 2 - 3))
            cx = x + half_len
            if cx + half_len > w - 8:
                half_len = max(6.0, (w - 8 - x) / 2)
                cx = x + half_len
            th = rng.choice([0.0, 0.0, 0.0, rng.uniform(-0.03, 0.03)])
            y0, y1 = int(max(0, cy - half_h - half_len * abs(th) - 14)), int(min(h, cy + half_h + half_len * abs(th) + 14))
            x0, x1 = int(max(0, cx - half_len - 14)), int(min(w, cx + half_len + 14))
            yy, xx = np.mgrid[y0:y1, x0:x1].astype(np.float32)
            u = (xx - cx) * np.cos(th) + (yy - cy) * np.sin(th)
            v = -(xx - cx) * np.sin(th) + (yy - cy) * np.cos(th)
            blob = np.exp(-np.maximum(np.abs(u) / half_len, np.abs(v) / half_h) ** 4) * rng.uniform(0.7, 0.98)
            m[y0:y1, x0:x1] = np.maximum(m[y0:y1, x0:x1], blob.astype(np.float32))
            x = cx + half_len + rng.uniform(12, 40)
            done += 1
    return np.ascontiguousarray(np.clip(m, 0.001, 0.999).astype(np.float32))


# ------------------------------------------------------------------------------------------------- layout model
def make_layout_weights(cfg, seed: int = 0) -> Dict[str, torch.Tensor]:
    """fp32 CPU state dict with the reference's parameter names: `encoder.*` = DonutSwinLayoutModel (surya/layout/model/encoder.py,
    surya/common/donut/encoder.py), `decoder.*` = SuryaLayoutDecoder (surya/layout/model/decoder.py, surya/common/adetr/decoder.py).
    Gains are chosen so activations stay O(1) through the stacks and the class logits are spread (a random tied-scale head would
    emit one label forever)."""
    g = torch.Generator().manual_seed(seed)
    e, d = cfg.encoder, cfg.decoder
    sd: Dict[str, torch.Tensor] = {}
    E = e.embed_dim
    sd["encoder.embeddings.patch_embeddings.projection.weight"] = _normal(g, (E, e.num_channels, e.patch_size, e.patch_size),
                                                                            1.0 / math.sqrt(e.num_channels * e.patch_size ** 2))
    sd["encoder.embeddings.patch_embeddings.projection.bias"] = _normal(g, (E,), 0.02)
    sd["encoder.embeddings.norm.weight"] = 1.0 + _normal(g, (E,), 0.05)
    sd["encoder.embeddings.norm.bias"] = _normal(g, (E,), 0.05)
    ws = e.window_size
    for si, depth in enumerate(e.depths):
        dim, nh, nkv = E * 2 ** si, e.num_heads[si], e.num_kv_heads[si]
        hd = dim // nh
        for bi in range(depth):
            p = f"encoder.encoder.layers.{si}.blocks.{bi}."
            sd[p + "layernorm_before.weight"] = 1.0 + _normal(g, (dim,), 0.05)
            sd[p + "layernorm_before.bias"] = _normal(g, (dim,), 0.05)
            sd[p + "attention.self.relative_position_bias_table"] = _normal(g, ((2 * ws - 1) ** 2, nh), 0.5)
            sd[p + "attention.self.query.weight"] = _normal(g, (dim, dim), 1.2 / math.sqrt(dim))
            sd[p + "attention.self.query.bias"] = _normal(g, (dim,), 0.05)
            sd[p + "attention.self.key.weight"] = _normal(g, (nkv * hd, dim), 1.2 / math.sqrt(dim))
            sd[p + "attention.self.key.bias"] = _normal(g, (nkv * hd,), 0.05)
            sd[p + "attention.self.value.weight"] = _normal(g, (nkv * hd, dim), 1.0 / math.sqrt(dim))
            sd[p + "attention.self.value.bias"] = _normal(g, (nkv * hd,), 0.05)
            sd[p + "attention.output.dense.weight"] = _normal(g, (dim, dim), 0.5 / math.sqrt(dim))
            sd[p + "attention.output.dense.bias"] = _normal(g, (dim,), 0.02)
            sd[p + "layernorm_after.weight"] = 1.0 + _normal(g, (dim,), 0.05)
            sd[p + "layernorm_after.bias"] = _normal(g, (dim,), 0.05)
            inter = int(e.mlp_ratio * dim)
            sd[p + "intermediate.dense.weight"] = _normal(g, (inter, dim), 1.0 / math.sqrt(dim))
            sd[p + "intermediate.dense.bias"] = _normal(g, (inter,), 0.05)
            sd[p + "output.dense.weight"] = _normal(g, (dim, inter), 0.5 / math.sqrt(inter))
            sd[p + "output.dense.bias"] = _normal(g, (dim,), 0.02)
        if si < len(e.depths) - 1:
            p = f"encoder.encoder.layers.{si}.downsample."
            sd[p + "reduction.weight"] = _normal(g, (2 * dim, 4 * dim), 1.0 / math.sqrt(4 * dim))
            sd[p + "norm.weight"] = 1.0 + _normal(g, (4 * dim,), 0.05)
            sd[p + "norm.bias"] = _normal(g, (4 * dim,), 0.05)
    sd["encoder.position_embeddings"] = _normal(g, (1, e.encoder_length, e.hidden_size), 0.3)
    H, I, qd, kvd = d.hidden_size, d.intermediate_size, d.num_attention_heads * d.head_dim, d.num_key_value_heads * d.head_dim
    table = hasattr(d, "box_embed_size")                      # table_rec.config.TableDecoderConfig: concat(box, property) embedding
    BE = d.box_embed_size if table else H
    for nm in ("w", "h", "cx", "cy", "xskew", "yskew", "x1", "y1", "x2", "y2", "x3", "y3", "x4", "y4"):
        sd[f"decoder.model.embed_tokens.{nm}_embed.weight"] = _normal(g, (d.vocab_size, BE), 0.3)
    if table:
        P = d.property_embed_size
        sd["decoder.model.embed_tokens.category_embed.weight"] = _normal(g, (d.category_count, P), 0.5)
        sd["decoder.model.embed_tokens.merge_embed.weight"] = _normal(g, (d.merge_count, P), 0.5)
        sd["decoder.model.embed_tokens.colspan_embed.weight"] = _normal(g, (d.vocab_size, P), 0.5)
    else:
        sd["decoder.model.embed_tokens.label_embed.weight"] = _normal(g, (d.label_count, H), 0.5)
    for li in range(d.num_hidden_layers):
        p = f"decoder.model.layers.{li}."
        for nm in ("cross_pre_norm", "temporal_pre_norm", "channel_pre_norm"):
            sd[p + nm + ".weight"] = _normal(g, (H,), 0.1)                      # the norm multiplies by (1 + weight)
        for blk, kin in (("temporal_block", H), ("cross_attn_block", d.encoder_hidden_size)):
            sd[p + blk + ".q_proj.weight"] = _normal(g, (qd, H), 1.2 / math.sqrt(H))
            sd[p + blk + ".k_proj.weight"] = _normal(g, (kvd, kin), 1.2 / math.sqrt(kin))
            sd[p + blk + ".v_proj.weight"] = _normal(g, (kvd, kin), 1.0 / math.sqrt(kin))
            sd[p + blk + ".o_proj.weight"] = _normal(g, (H, qd), 0.7 / math.sqrt(qd))
            sd[p + blk + ".o_proj.bias"] = _normal(g, (H,), 0.02)
        sd[p + "mlp_block.gate_proj.weight"] = _normal(g, (I, H), 1.0 / math.sqrt(H))
        sd[p + "mlp_block.up_proj.weight"] = _normal(g, (I, H), 1.0 / math.sqrt(H))
        sd[p + "mlp_block.down_proj.weight"] = _normal(g, (H, I), 0.7 / math.sqrt(I))
    sd["decoder.model.final_norm.weight"] = _normal(g, (H,), 0.1)
    sd["decoder.pre_output_norm.weight"] = 1.0 + _normal(g, (H,), 0.05)
    sd["decoder.pre_output_norm.bias"] = _normal(g, (H,), 0.05)
    if table:
        # surya/table_rec/model/decoder.py:88-93: one bias-free Linear per box property
        for k, rows in d.head_widths():
            gain = {"bbox": 1.0, "colspan": 2.0}.get(k, 1.5)
            sd[f"decoder.box_property_heads.{k}.weight"] = _normal(g, (rows, H), gain / math.sqrt(H))
        # nothing ends a table early by itself: keep </S> / <PAD> unlikely as a category so a fixed number of boxes is decoded in tests
        sd["decoder.box_property_heads.category.weight"][: d.special_token_count] *= 0.05
        return sd
    sd["decoder.lm_head.weight"] = _normal(g, (d.label_count, H), 1.5 / math.sqrt(H))
    # nothing ends a page early by itself: keep </S> / <PAD> / pause unlikely so a fixed number of boxes is decoded in tests
    sd["decoder.lm_head.weight"][: d.special_token_count] *= 0.05
    sd["decoder.bbox_head.weight"] = _normal(g, (6, H), 1.0 / math.sqrt(H))
    sd["decoder.bbox_head.bias"] = _normal(g, (6,), 0.1)
    return sd


def make_table_weights(cfg, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Table-recognition twin of make_layout_weights: `encoder.*` = table_rec's DonutSwinModel (the same Swin stack,
    surya/table_rec/model/encoder.py), `decoder.*` = SuryaTableRecDecoder (surya/table_rec/model/decoder.py)."""
    return make_layout_weights(cfg, seed)
