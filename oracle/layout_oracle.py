"""CPU oracle for the layout model family -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's checker legs may import this module. A plain-PyTorch restatement of what the
reference (VikParuchuri/surya @ v0.14.6) computes in LayoutPredictor's model path: the Donut-Swin encoder
(surya/common/donut/encoder.py, surya/layout/model/encoder.py) and the ADETR decoder with cross- and self-attention
(surya/common/adetr/decoder.py, surya/layout/model/decoder.py), plus the greedy box loop of surya/layout/__init__.py:95-190.
Pinned against the real reference modules through oracle/ref_shim (oracle/make_golden_layout.py -> tests/golden/layout_*.pt and
the live test in tests/test_oracle_vs_reference.py).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------------------------------------------- encoder
def relative_position_index(ws: int) -> torch.Tensor:
    """donut/encoder.py:348-359."""
    coords = torch.stack(torch.meshgrid([torch.arange(ws), torch.arange(ws)], indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def sincos_2d(width: int, height: int, dim: int, temperature: float = 10000.0) -> torch.Tensor:
    """DonutSwinStage.build_2d_sincos_position_embedding (donut/encoder.py:735-757), called as (input_resolution[1],
    input_resolution[0]): meshgrid(indexing="ij") over (w, h), i.e. row n of the table belongs to (w = n // H, h = n % H) while token
    n of the stage is (h = n // W, w = n % W) -- the reference's own pairing, kept as is."""
    gw, gh = torch.meshgrid(torch.arange(int(width), dtype=torch.float32), torch.arange(int(height), dtype=torch.float32), indexing="ij")
    pos_dim = dim // 4
    omega = 1.0 / (temperature ** (torch.arange(pos_dim, dtype=torch.float32) / pos_dim))
    ow, oh = gw.flatten()[..., None] @ omega[None], gh.flatten()[..., None] @ omega[None]
    return torch.cat([ow.sin(), ow.cos(), oh.sin(), oh.cos()], dim=1)


def window_partition(x, ws):
    B, H, W, C = x.shape
    return x.view(B, H // ws, ws, W // ws, ws, C).permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws, ws, C)


def window_reverse(w, ws, H, W):
    C = w.shape[-1]
    return w.view(-1, H // ws, W // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, H, W, C)


def shift_mask(Hp: int, Wp: int, ws: int, shift: int, dtype) -> Optional[torch.Tensor]:
    """DonutSwinLayer.get_attn_mask (donut/encoder.py:560-586): -100 between tokens of different cyclic-shift regions."""
    if shift <= 0:
        return None
    img = torch.zeros((1, Hp, Wp, 1), dtype=dtype)
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[:, hs, wsl, :] = cnt
            cnt += 1
    mw = window_partition(img, ws).view(-1, ws * ws)
    am = mw.unsqueeze(1) - mw.unsqueeze(2)
    return am.masked_fill(am != 0, -100.0).masked_fill(am == 0, 0.0)


def swin_layer(sd: SD, p: str, x: torch.Tensor, hw: Tuple[int, int], nh: int, nkv: int, ws_cfg: int, shift_cfg: int, eps: float):
    """DonutSwinLayer.forward (donut/encoder.py:598-686) with DonutSwinSelfAttention (:387-444)."""
    H, W = hw
    B, _, C = x.shape
    ws, shift = (min(hw), 0) if min(hw) <= ws_cfg else (ws_cfg, shift_cfg)        # set_shift_and_window_size :551-559
    shortcut = x
    h = F.layer_norm(x, (C,), sd[p + "layernorm_before.weight"], sd[p + "layernorm_before.bias"], eps).view(B, H, W, C)
    pr, pb = (ws - W % ws) % ws, (ws - H % ws) % ws
    h = F.pad(h, (0, 0, 0, pr, 0, pb))
    Hp, Wp = H + pb, W + pr
    if shift > 0:
        h = torch.roll(h, shifts=(-shift, -shift), dims=(1, 2))
    hwin = window_partition(h, ws).view(-1, ws * ws, C)
    mask = shift_mask(Hp, Wp, ws, shift, x.dtype)
    hd = C // nh
    nW, N = hwin.shape[0], ws * ws
    q = F.linear(hwin, sd[p + "attention.self.query.weight"], sd[p + "attention.self.query.bias"]).view(nW, N, nh, hd).permute(0, 2, 1, 3)
    k = F.linear(hwin, sd[p + "attention.self.key.weight"], sd[p + "attention.self.key.bias"]).view(nW, N, nkv, hd)
    v = F.linear(hwin, sd[p + "attention.self.value.weight"], sd[p + "attention.self.value.bias"]).view(nW, N, nkv, hd)
    k = k.repeat(1, 1, nh // nkv, 1).permute(0, 2, 1, 3)                          # query head h reads kv head h % nkv (:379-385)
    v = v.repeat(1, 1, nh // nkv, 1).permute(0, 2, 1, 3)
    ws_tab = ws_cfg                                                                  # the bias table is built for the CONFIG's window
    bias = sd[p + "attention.self.relative_position_bias_table"][relative_position_index(ws_tab).view(-1)]
    bias = bias.view(ws_tab * ws_tab, ws_tab * ws_tab, -1).permute(2, 0, 1).contiguous().unsqueeze(0)
    if ws != ws_tab:
        raise NotImplementedError("input resolution below the window size (the reference's bias table would not fit either)")
    am = bias if mask is None else mask.repeat(nW // mask.shape[0], 1, 1).unsqueeze(1) + bias
    a = F.scaled_dot_product_attention(q, k, v, attn_mask=am.to(q.dtype), scale=hd ** -0.5)
    a = a.transpose(1, 2).reshape(nW, N, C)
    a = F.linear(a, sd[p + "attention.output.dense.weight"], sd[p + "attention.output.dense.bias"])
    a = window_reverse(a.view(-1, ws, ws, C), ws, Hp, Wp)
    if shift > 0:
        a = torch.roll(a, shifts=(shift, shift), dims=(1, 2))
    a = a[:, :H, :W, :].contiguous().view(B, H * W, C)
    x = shortcut + a
    y = F.layer_norm(x, (C,), sd[p + "layernorm_after.weight"], sd[p + "layernorm_after.bias"], eps)
    y = F.gelu(F.linear(y, sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"]))
    return x + F.linear(y, sd[p + "output.dense.weight"], sd[p + "output.dense.bias"])


def patch_merging(sd: SD, p: str, x: torch.Tensor, hw: Tuple[int, int]):
    """DonutSwinPatchMerging.forward (donut/encoder.py:289-319); LayerNorm of 4C with the default eps 1e-5."""
    H, W = hw
    B, _, C = x.shape
    x = x.view(B, H, W, C)
    x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
    x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1).view(B, -1, 4 * C)
    x = F.layer_norm(x, (4 * C,), sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-5)
    return F.linear(x, sd[p + "reduction.weight"])


def encoder_forward(sd: SD, e, pixel_values: torch.Tensor, taps: Optional[dict] = None) -> torch.Tensor:
    """DonutSwinLayoutModel.forward (layout/model/encoder.py:36-80): patch embedding (conv 4x4 / 4) + LayerNorm, stages with the 2-D
    sin-cos table added at their entry, patch merging between stages, learned position_embeddings added to the output."""
    P = e.patch_size
    x = pixel_values
    _, _, Hh, Ww = x.shape
    x = F.pad(x, (0, (P - Ww % P) % P, 0, (P - Hh % P) % P))
    x = F.conv2d(x, sd["encoder.embeddings.patch_embeddings.projection.weight"], sd["encoder.embeddings.patch_embeddings.projection.bias"],
                 stride=P)
    hw = (x.shape[2], x.shape[3])
    x = x.flatten(2).transpose(1, 2)
    x = F.layer_norm(x, (e.embed_dim,), sd["encoder.embeddings.norm.weight"], sd["encoder.embeddings.norm.bias"], 1e-5)
    if taps is not None:
        taps["embeddings"] = x.clone()
    grid = e.grid
    for si, depth in enumerate(e.depths):
        dim = e.embed_dim * 2 ** si
        if e.use_positional_embeddings:
            res = (grid[0] // 2 ** si, grid[1] // 2 ** si)                            # the stage's CONFIGURED resolution (:823-826)
            x = x + sincos_2d(res[1], res[0], dim).to(x.dtype)[None]
        for bi in range(depth):
            x = swin_layer(sd, f"encoder.encoder.layers.{si}.blocks.{bi}.", x, hw, e.num_heads[si], e.num_kv_heads[si], e.window_size,
                           0 if bi % 2 == 0 else e.window_size // 2, e.layer_norm_eps)
        if taps is not None:
            taps[f"stage{si}"] = x.clone()
        if si < len(e.depths) - 1:
            x = patch_merging(sd, f"encoder.encoder.layers.{si}.downsample.", x, hw)
            hw = ((hw[0] + 1) // 2, (hw[1] + 1) // 2)
    return x + sd["encoder.position_embeddings"][:, : x.shape[1], :].to(x.dtype)


# --------------------------------------------------------------------------------------------------------------- decoder
def adetr_rms_norm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    """SuryaADETRDecoderRMSNorm (adetr/decoder.py:23-47): variance clamped at eps (not added), (1 + weight), clamp to the dtype's range."""
    xf = x.float()
    var = torch.clamp(xf.pow(2).mean(-1, keepdim=True), min=eps)
    out = xf * torch.rsqrt(var) * (1.0 + w.float())
    info = torch.finfo(x.dtype)
    out = out.clamp(min=info.min, max=info.max)
    out = torch.where(torch.isnan(out), torch.tensor(0.0), out)
    return out.type_as(x)


def rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def embed_boxes(sd: SD, d, boxes: torch.Tensor) -> torch.Tensor:
    """BboxEmbedding.forward (layout/model/decoder.py:36-60). boxes: long [B, T, 7]."""
    p = "decoder.model.embed_tokens."
    cx, cy, w, h, xs, ys, label = boxes.to(torch.long).unbind(dim=-1)
    xa = ((xs - d.bbox_size // 2) / 2).to(torch.long)
    ya = ((ys - d.bbox_size // 2) / 2).to(torch.long)
    cl = lambda t: t.clamp(0, d.bbox_size).to(torch.long)
    x1, y1 = cl(cx - w // 2 - xa), cl(cy - h // 2 - ya)
    x2, y2 = cl(cx + w // 2 - xa), cl(cy + h // 2 + ya)
    x3, y3 = cl(cx + w // 2 + xa), cl(cy + h // 2 + ya)
    x4, y4 = cl(cx - w // 2 + xa), cl(cy - h // 2 - ya)
    E = lambda nm, idx: sd[p + nm + "_embed.weight"][idx]
    label_e = E("label", label)
    size_e = E("w", w) + E("h", h) + E("cx", cx) + E("cy", cy)
    skew_e = E("xskew", xs) + E("yskew", ys)
    corner_e = E("x1", x1) + E("y1", y1) + E("x2", x2) + E("y2", y2) + E("x3", x3) + E("y3", y3) + E("x4", x4) + E("y4", y4)
    return label_e + size_e + skew_e + corner_e


def embed_table_tokens(sd: SD, d, boxes: torch.Tensor) -> torch.Tensor:
    """LabelEmbedding.forward of the table-recognition decoder (table_rec/model/decoder.py:46-73). boxes: long [B, T, 10] =
    (cx, cy, w, h, xskew, yskew, category, merges, colspan, is_header); is_header is not embedded; x2 / y2 / x4 / y4 tables exist but
    only x1, y1, x3, y3 are read."""
    p = "decoder.model.embed_tokens."
    boxes = boxes.to(torch.long).clamp(0, d.vocab_size)
    cx, cy, w, h, xs, ys, cat, mer, col, _ = boxes.unbind(dim=-1)
    xa = ((xs - d.bbox_size // 2) / 2).to(torch.long)
    ya = ((ys - d.bbox_size // 2) / 2).to(torch.long)
    cl = lambda t: t.clamp(0, d.bbox_size).to(torch.long)
    x1, y1 = cl(cx - w // 2 - xa), cl(cy - h // 2 - ya)
    x3, y3 = cl(cx + w // 2 + xa), cl(cy + h // 2 + ya)
    E = lambda nm, idx: sd[p + nm + "_embed.weight"][idx]
    size_e = E("w", w) + E("h", h) + E("cx", cx) + E("cy", cy)
    skew_e = E("xskew", xs) + E("yskew", ys)
    corner_e = E("x1", x1) + E("y1", y1) + E("x3", x3) + E("y3", y3)
    prop_e = E("category", cat) + E("merge", mer) + E("colspan", col)
    return torch.cat([size_e + skew_e + corner_e, prop_e], dim=-1)


class LayoutDecoderState:
    """Caches of one batch: cross-attention K / V of the encoder states (computed at the first call, adetr/decoder.py:167-173) and
    the growing self-attention K / V (dynamic cache, :300-311)."""

    def __init__(self, n_layers: int):
        self.cross_k: List[Optional[torch.Tensor]] = [None] * n_layers
        self.cross_v: List[Optional[torch.Tensor]] = [None] * n_layers
        self.self_k: List[Optional[torch.Tensor]] = [None] * n_layers
        self.self_v: List[Optional[torch.Tensor]] = [None] * n_layers


def decoder_forward(sd: SD, d, boxes: torch.Tensor, enc: torch.Tensor, pos0: int, st: LayoutDecoderState):
    """SuryaLayoutDecoder.forward (layout/model/decoder.py:96-131) for T new tokens at positions pos0 .. pos0 + T - 1:
    returns (bbox_logits [B, T, 6] after sigmoid, class_logits [B, T, label_count]).
    With a table_rec TableDecoderConfig: SuryaTableRecDecoder.forward (table_rec/model/decoder.py:115-154) -- LabelEmbedding, the plain
    residual flow (double_residual_flow = False, config.py:219), one bias-free head per box property -- returning
    (sigmoid(bbox) [B, T, 6], {property: logits [B, T, n]})."""
    B, T, _ = boxes.shape
    nq, nkv, hd = d.num_attention_heads, d.num_key_value_heads, d.head_dim
    table = hasattr(d, "box_embed_size")
    x = embed_table_tokens(sd, d, boxes) if table else embed_boxes(sd, d, boxes)
    pos = torch.arange(pos0, pos0 + T, dtype=torch.float32)
    inv = 1.0 / (d.rope_theta ** (torch.arange(0, hd, 2, dtype=torch.int64).float() / hd))
    fr = pos[:, None] * inv[None, :]
    emb = torch.cat((fr, fr), dim=-1)
    cos, sin = emb.cos().to(x.dtype)[None, None], emb.sin().to(x.dtype)[None, None]
    # causal mask of _update_causal_mask (:600-632) restricted to the keys that exist
    past = pos0
    keys = past + T
    cm = torch.zeros((T, keys), dtype=x.dtype)
    if T > 1:
        cm = torch.triu(torch.full((T, keys), torch.finfo(x.dtype).min, dtype=x.dtype), diagonal=past + 1)
    rep = lambda t: t[:, :, None].expand(B, nkv, nq // nkv, t.shape[2], hd).reshape(B, nq, t.shape[2], hd)
    for li in range(d.num_hidden_layers):
        p = f"decoder.model.layers.{li}."
        raw = x
        # cross attention on the encoder states (:126-190)
        h = adetr_rms_norm(x, sd[p + "cross_pre_norm.weight"], d.rms_norm_eps)
        q = F.linear(h, sd[p + "cross_attn_block.q_proj.weight"]).view(B, T, nq, hd).transpose(1, 2)
        if st.cross_k[li] is None:
            st.cross_k[li] = F.linear(enc, sd[p + "cross_attn_block.k_proj.weight"]).view(B, -1, nkv, hd).transpose(1, 2)
            st.cross_v[li] = F.linear(enc, sd[p + "cross_attn_block.v_proj.weight"]).view(B, -1, nkv, hd).transpose(1, 2)
        a = F.scaled_dot_product_attention(q, rep(st.cross_k[li]), rep(st.cross_v[li]), scale=hd ** -0.5)
        a = a.transpose(1, 2).reshape(B, T, nq * hd)
        cross = F.linear(a, sd[p + "cross_attn_block.o_proj.weight"], sd[p + "cross_attn_block.o_proj.bias"]) + raw
        # self attention with RoPE and the KV cache (:213-270); double residual flow (:430-457): its input is the cross-attention
        # output, its residual the layer's RAW input
        h = adetr_rms_norm(cross, sd[p + "temporal_pre_norm.weight"], d.rms_norm_eps)
        q = F.linear(h, sd[p + "temporal_block.q_proj.weight"]).view(B, T, nq, hd).transpose(1, 2)
        k = F.linear(h, sd[p + "temporal_block.k_proj.weight"]).view(B, T, nkv, hd).transpose(1, 2)
        v = F.linear(h, sd[p + "temporal_block.v_proj.weight"]).view(B, T, nkv, hd).transpose(1, 2)
        q = q * cos + rotate_half(q) * sin
        k = k * cos + rotate_half(k) * sin
        st.self_k[li] = k if st.self_k[li] is None else torch.cat([st.self_k[li], k], dim=2)
        st.self_v[li] = v if st.self_v[li] is None else torch.cat([st.self_v[li], v], dim=2)
        a = F.scaled_dot_product_attention(q, rep(st.self_k[li]), rep(st.self_v[li]), attn_mask=cm[None, None], scale=hd ** -0.5)
        a = a.transpose(1, 2).reshape(B, T, nq * hd)
        # layout adds the layer's RAW input here (double_res_forward, adetr/decoder.py:419-457), table_rec the cross-attention output (:395-417)
        res = F.linear(a, sd[p + "temporal_block.o_proj.weight"], sd[p + "temporal_block.o_proj.bias"]) + (cross if table else raw)
        h = adetr_rms_norm(res, sd[p + "channel_pre_norm.weight"], d.rms_norm_eps)
        h = F.gelu(F.linear(h, sd[p + "mlp_block.gate_proj.weight"]), approximate="tanh") * F.linear(h, sd[p + "mlp_block.up_proj.weight"])
        x = F.linear(h, sd[p + "mlp_block.down_proj.weight"]) + res
    x = adetr_rms_norm(x, sd["decoder.model.final_norm.weight"], d.rms_norm_eps)
    x = F.layer_norm(x, (d.hidden_size,), sd["decoder.pre_output_norm.weight"], sd["decoder.pre_output_norm.bias"], d.layer_norm_eps)
    if table:
        props = {k: F.linear(x, sd[f"decoder.box_property_heads.{k}.weight"]) for k, _ in d.head_widths()}
        return torch.sigmoid(props.pop("bbox")), props
    cls = F.linear(x, sd["decoder.lm_head.weight"])
    box = torch.sigmoid(F.linear(x, sd["decoder.bbox_head.weight"], sd["decoder.bbox_head.bias"]))
    return box, cls


@torch.inference_mode()
def generate(sd: SD, cfg, pixel_values: torch.Tensor, max_boxes: int, record: bool = False):
    """The model side of LayoutPredictor.batch_layout_detection's loop (surya/layout/__init__.py:95-131): start token, then feed
    back (box_logits * bbox_size, argmax class) until every image emitted </S> / <PAD> or max_boxes tokens were decoded. Returns
    per step the float inputs the reference would build (box preds, class preds) and, if asked, the logits."""
    d = cfg.decoder
    enc = encoder_forward(sd, cfg.encoder, pixel_values)
    B = pixel_values.shape[0]
    boxes = torch.tensor([[[d.bos_token_id] * 7] + [[d.pause_token_id] * 7] * d.pause_token_count] * B, dtype=torch.long)
    st = LayoutDecoderState(d.num_hidden_layers)
    pos, count = 0, 0
    all_done = torch.zeros(B, dtype=torch.bool)
    steps = []
    while count < max_boxes:
        box, cls = decoder_forward(sd, d, boxes, enc, pos, st)
        pos += boxes.shape[1]
        bl, cl = box[:, -1, :], cls[:, -1, :]
        cp = cl.argmax(-1)
        bp = bl * d.bbox_size
        all_done = all_done | (cp == d.eos_token_id) | (cp == d.pad_token_id)
        steps.append({"box_preds": bp.clone(), "class_preds": cp.clone(), **({"class_logits": cl.clone(), "bbox_logits": bl.clone()} if record else {})})
        if all_done.all():
            break
        nxt = torch.cat([bp.unsqueeze(1), cp.unsqueeze(1).unsqueeze(1).to(bp.dtype)], dim=-1)
        count += boxes.shape[1]
        boxes = nxt.to(torch.long)
    return enc, steps


# ------------------------------------------------------------------------------------------------- the fed-back token (host rule)
# What the reference's loops do with a step's outputs before the next decoder call -- restated per row with the reference's own
# tensor operations, as the checker of the device-fed decode runs (surya_layout_decode_steps) and of the predictors' host rule.
PAGE_HEADER_FOOTER_IDS = (9, 10)          # "PageFooter", "PageHeader" in ID_TO_LABEL (surya/layout/model/config.py:16-34)


def polygon_of_prediction(pred: torch.Tensor, img_size, bbox_scaler, skew_scaler, skew_min=0.001):
    """surya/layout/util.py:4-40: corner arithmetic on the tensor (its dtype), `.item() * scale` in Python floats."""
    w_scale, h_scale = img_size[0] / bbox_scaler, img_size[1] / bbox_scaler
    cx, cy, width, height = pred[0], pred[1], pred[2], pred[3]
    x1, y1, x2, y2 = cx - width / 2, cy - height / 2, cx + width / 2, cy + height / 2
    skew_x = torch.floor((pred[4] - skew_scaler) / 2)
    skew_y = torch.floor((pred[5] - skew_scaler) / 2)
    if abs(skew_x.item()) < skew_min:
        skew_x = torch.zeros_like(skew_x)
    if abs(skew_y.item()) < skew_min:
        skew_y = torch.zeros_like(skew_y)
    flat = [x1 - skew_x, y1 - skew_y, x2 - skew_x, y1 + skew_y, x2 + skew_x, y2 + skew_y, x1 + skew_x, y2 - skew_y]
    return [[flat[2 * i].item() * w_scale, flat[2 * i + 1].item() * h_scale] for i in range(4)]


def fed_token_layout(class_logits: torch.Tensor, box_logits: torch.Tensor, d, page_size=None, relabel_ids=None) -> torch.Tensor:
    """surya/layout/__init__.py:117-131, 158-177 for ONE page: class_logits [labels], box_logits [6] in the model dtype; page_size =
    (width, height) of the slice or None (rule off); relabel_ids overrides the two class ids the rule applies to (tests make the
    rule fire on synthetic weights that way). Returns the int64 token [7] the next decoder call receives."""
    class_pred = class_logits.argmax(-1)
    box_preds = box_logits * d.bbox_size
    tok = torch.cat([box_preds, class_pred[None].to(box_preds.dtype)], -1)              # batch_decoder_input: the model dtype
    ids = tuple(relabel_ids) if relabel_ids is not None else tuple(i + d.special_token_count for i in PAGE_HEADER_FOOTER_IDS)
    if page_size is not None and int(class_pred) in ids:
        poly = polygon_of_prediction(tok, page_size, d.bbox_size, d.skew_scaler)
        w, h = page_size
        if poly[0][1] < h * .8 and poly[2][1] > h * .2 and poly[0][0] < w * .8 and poly[2][0] > w * .2:
            logits = class_logits.clone()
            logits[int(class_pred)] = 0
            tok[6] = logits.argmax(-1).item()
    return tok.to(torch.long)


def fed_token_table(class_logits: torch.Tensor, box_logits: torch.Tensor, d, box_dim: int = 1024, special_tokens: int = 5) -> torch.Tensor:
    """surya/table_rec/__init__.py:80-118 + LabelShaper.dict_to_labels (surya/table_rec/shaper.py:12-51) for ONE row: class_logits = the
    property heads side by side [category | merges | colspan | is_header], box_logits [6]. Returns the int64 token [10]."""
    o, vals = 0, {}
    for k, n in d.head_widths():
        if k == "bbox":
            continue
        seg = class_logits[o:o + n]
        o += n
        if k == "colspan":                                   # regression head: clamp(min = 1), round (:96-98)
            vals[k] = int(torch.round(torch.clamp(seg, min=1))[0].item())
        else:                                                # classification: argmax - special tokens (:91-93); dict_to_labels adds them back
            vals[k] = int(seg.argmax(-1).item()) - special_tokens
    bbox = (box_logits * box_dim).tolist()                   # (:99-100)
    bbox = [0 if v < 0 else (box_dim if v > box_dim else v) for v in bbox]          # shaper.py:24-27
    vec = bbox + [vals["category"] + special_tokens, vals["merges"] + special_tokens, vals["colspan"], vals["is_header"] + special_tokens]
    return torch.tensor(vec, dtype=torch.float64).to(torch.long)                    # torch.tensor(labels, dtype = long): truncation
