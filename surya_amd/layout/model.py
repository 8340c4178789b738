"""HipLayoutModel: the layout model (Donut-Swin encoder + ADETR decoder) behind libsurya_amd.so's surya_layout_* entry points.

Python here is plumbing only (weight re-layout at load, device memory via torch, ctypes marshalling); all arithmetic runs in the
HIP library. There is no fallback: constructing this without the built library or without a GPU raises."""
from __future__ import annotations

import ctypes as C
from typing import List

import numpy as np
import torch

from .. import _lib as L
from .config import LayoutConfig

(LW_PATCH_W, LW_PATCH_B, LW_EMB_LN_W, LW_EMB_LN_B, LW_POS_EMB, LW_DEC_FNORM, LW_DEC_LN_W, LW_DEC_LN_B, LW_DEC_LM_W, LW_DEC_BB_W, LW_DEC_BB_B,
 LW_DEC_INVFREQ, LW_DEC_ZERO_BIAS, LW_EMB_TABLES) = range(14)
LW_GLOBALS = LW_EMB_TABLES + 17
FAMILY_LAYOUT, FAMILY_TABLE = 0, 1
LS_COUNT, LB_COUNT, LD_COUNT = 4, 13, 12
EMB_ORDER = ("w", "h", "cx", "cy", "xskew", "yskew", "x1", "y1", "x2", "y2", "x3", "y3", "x4", "y4", "label")


class LayoutConfigC(C.Structure):
    _fields_ = [("img_h", C.c_int32), ("img_w", C.c_int32), ("patch", C.c_int32), ("embed_dim", C.c_int32), ("n_stages", C.c_int32),
                ("depths", C.c_int32 * 8), ("heads", C.c_int32 * 8), ("kv_heads", C.c_int32 * 8), ("window", C.c_int32),
                ("enc_eps", C.c_float), ("encoder_length", C.c_int32), ("dec_layers", C.c_int32), ("dec_hidden", C.c_int32),
                ("dec_inter", C.c_int32), ("dec_heads", C.c_int32), ("dec_kv_heads", C.c_int32), ("vocab", C.c_int32),
                ("label_count", C.c_int32), ("bbox_size", C.c_int32), ("rms_eps", C.c_float), ("ln_eps", C.c_float),
                ("max_batch", C.c_int32), ("max_boxes", C.c_int32), ("dtype", C.c_int32),
                ("family", C.c_int32), ("box_embed", C.c_int32), ("category_count", C.c_int32), ("merge_count", C.c_int32)]


class LayoutFeedbackC(C.Structure):
    """surya_layout_feedback (include/surya_amd.h)."""
    _fields_ = [("skew_scaler", C.c_int32), ("relabel_ids", C.c_int32 * 2), ("head_widths", C.c_int32 * 4), ("page_sizes", C.POINTER(C.c_int32))]


RING_STEPS = 16            # LayoutModel::RING_STEPS: the longest device-fed run


def _relative_position_index(ws: int) -> torch.Tensor:
    """surya/common/donut/encoder.py:348-359."""
    coords = torch.stack(torch.meshgrid([torch.arange(ws), torch.arange(ws)], indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def _sincos_2d(width: int, height: int, dim: int, temperature: float = 10000.0) -> torch.Tensor:
    """DonutSwinStage.build_2d_sincos_position_embedding (donut/encoder.py:735-757), with the reference's (w, h) meshgrid order."""
    gw, gh = torch.meshgrid(torch.arange(int(width), dtype=torch.float32), torch.arange(int(height), dtype=torch.float32), indexing="ij")
    pos_dim = dim // 4
    omega = 1.0 / (temperature ** (torch.arange(pos_dim, dtype=torch.float32) / pos_dim))
    ow, oh = gw.flatten()[..., None] @ omega[None], gh.flatten()[..., None] @ omega[None]
    return torch.cat([ow.sin(), ow.cos(), oh.sin(), oh.cos()], dim=1)


def _interleave(g: torch.Tensor, u: torch.Tensor) -> torch.Tensor:
    out = g.new_zeros((2 * g.shape[0],) + tuple(g.shape[1:]))
    out[0::2] = g
    out[1::2] = u
    return out


def is_table_config(cfg) -> bool:
    return hasattr(cfg.decoder, "box_embed_size")


def repack_layout_weights(cfg, sd, dtype: torch.dtype, device) -> List[torch.Tensor]:
    """Reference state dict (`encoder.*` = DonutSwinLayoutModel / table_rec's DonutSwinModel, `decoder.*` = SuryaLayoutDecoder /
    SuryaTableRecDecoder) -> the table of include/surya_amd.h (SA_LW_* / SA_LS_* / SA_LB_* / SA_LD_*)."""
    e, d = cfg.encoder, cfg.decoder
    out: List[torch.Tensor] = []

    def put(t, dt=None):
        out.append(t.to(device=device, dtype=dt or dtype).contiguous())

    f = lambda k: sd[k].float()
    E = e.embed_dim
    pw = f("encoder.embeddings.patch_embeddings.projection.weight").reshape(E, -1)
    pwp = torch.zeros((E, 64))
    pwp[:, : pw.shape[1]] = pw
    put(pwp); put(f("encoder.embeddings.patch_embeddings.projection.bias"))
    put(f("encoder.embeddings.norm.weight")); put(f("encoder.embeddings.norm.bias"))
    put(f("encoder.position_embeddings")[0])
    put(f("decoder.model.final_norm.weight")); put(f("decoder.pre_output_norm.weight")); put(f("decoder.pre_output_norm.bias"))
    table = is_table_config(cfg)
    if table:
        # the four non-bbox property heads stacked in BOX_PROPERTIES order (category | merges | colspan | is_header) take the lm_head
        # slot, the bbox head (Linear without bias, sigmoid in the kernel) the bbox slot (table_rec/model/decoder.py:88-93, :147-151)
        put(torch.cat([f(f"decoder.box_property_heads.{k}.weight") for k, _ in d.head_widths() if k != "bbox"], 0))
        put(f("decoder.box_property_heads.bbox.weight")); put(torch.zeros(6))
    else:
        put(f("decoder.lm_head.weight")); put(f("decoder.bbox_head.weight")); put(f("decoder.bbox_head.bias"))
    hd = d.head_dim
    put(1.0 / (d.rope_theta ** (torch.arange(0, hd, 2, dtype=torch.int64).float() / hd)), torch.float32)
    put(torch.zeros((d.num_attention_heads + 2 * d.num_key_value_heads) * hd))
    # The embedding kernels index these tables with clamped corner values in [0, bbox_size] and with class ids below the counts of
    # the config; the pointer table carries no sizes, so the row counts are checked HERE (the reference would raise an index error).
    def rows(key, need, what):
        t = f(key)
        if t.shape[0] < need:
            raise ValueError(f"{key}: {t.shape[0]} rows, but {what} needs {need} (config / checkpoint mismatch)")
        return t

    if d.vocab_size <= d.bbox_size:
        raise ValueError(f"decoder.vocab_size {d.vocab_size} must exceed bbox_size {d.bbox_size}: box corners take values 0..bbox_size")
    for nm in EMB_ORDER[:14]:
        put(rows(f"decoder.model.embed_tokens.{nm}_embed.weight", d.bbox_size + 1, "bbox_size + 1 corner values"))
    if table:
        put(rows("decoder.model.embed_tokens.category_embed.weight", d.category_count, "category_count"))
        put(rows("decoder.model.embed_tokens.merge_embed.weight", d.merge_count, "merge_count"))
        put(rows("decoder.model.embed_tokens.colspan_embed.weight", d.bbox_size + 1, "bbox_size + 1 colspan values"))
    else:
        put(rows("decoder.model.embed_tokens.label_embed.weight", d.label_count, "label_count")); put(torch.zeros(4)); put(torch.zeros(4))
    assert len(out) == LW_GLOBALS
    gh, gw = e.grid
    ws = e.window_size
    rel_idx = _relative_position_index(ws).view(-1)
    for si, depth in enumerate(e.depths):
        dim = E * 2 ** si
        res = (gh // 2 ** si, gw // 2 ** si)
        sincos = _sincos_2d(res[1], res[0], dim) if e.use_positional_embeddings else torch.zeros((res[0] * res[1], dim))
        put(sincos)
        if si < len(e.depths) - 1:
            p = f"encoder.encoder.layers.{si}.downsample."
            put(f(p + "norm.weight")); put(f(p + "norm.bias")); put(f(p + "reduction.weight"))
        else:
            put(torch.zeros(4)); put(torch.zeros(4)); put(torch.zeros(4))
        nh = e.num_heads[si]
        for bi in range(depth):
            p = f"encoder.encoder.layers.{si}.blocks.{bi}."
            put(f(p + "layernorm_before.weight")); put(f(p + "layernorm_before.bias"))
            put(torch.cat([f(p + "attention.self.query.weight"), f(p + "attention.self.key.weight"), f(p + "attention.self.value.weight")], 0))
            put(torch.cat([f(p + "attention.self.query.bias"), f(p + "attention.self.key.bias"), f(p + "attention.self.value.bias")], 0))
            # the reference adds the bias in the model dtype: round it there before it is widened to fp32 for the kernel
            bias = f(p + "attention.self.relative_position_bias_table")[rel_idx].view(ws * ws, ws * ws, nh).permute(2, 0, 1)
            put(bias.to(dtype).float(), torch.float32)
            put(f(p + "attention.output.dense.weight")); put(f(p + "attention.output.dense.bias"))
            put(f(p + "layernorm_after.weight")); put(f(p + "layernorm_after.bias"))
            put(f(p + "intermediate.dense.weight")); put(f(p + "intermediate.dense.bias"))
            put(f(p + "output.dense.weight")); put(f(p + "output.dense.bias"))
    for li in range(d.num_hidden_layers):
        p = f"decoder.model.layers.{li}."
        put(f(p + "cross_pre_norm.weight"))
        put(f(p + "cross_attn_block.q_proj.weight"))
        put(torch.cat([f(p + "cross_attn_block.k_proj.weight"), f(p + "cross_attn_block.v_proj.weight")], 0))
        put(f(p + "cross_attn_block.o_proj.weight")); put(f(p + "cross_attn_block.o_proj.bias"))
        put(f(p + "temporal_pre_norm.weight"))
        put(torch.cat([f(p + "temporal_block.q_proj.weight"), f(p + "temporal_block.k_proj.weight"), f(p + "temporal_block.v_proj.weight")], 0))
        put(f(p + "temporal_block.o_proj.weight")); put(f(p + "temporal_block.o_proj.bias"))
        put(f(p + "channel_pre_norm.weight"))
        put(_interleave(f(p + "mlp_block.gate_proj.weight"), f(p + "mlp_block.up_proj.weight")))
        put(f(p + "mlp_block.down_proj.weight"))
    return out


class HipLayoutModel:
    def __init__(self, cfg, state_dict, *, dtype: torch.dtype = torch.bfloat16, device="cuda:0", max_batch: int = 32,
                 max_boxes: int = 100):
        """cfg: layout.config.LayoutConfig or table_rec.config.TableRecConfig (the two callers of the model family)."""
        if not torch.cuda.is_available():
            raise L.SuryaAmdError("HipLayoutModel needs a GPU (MI355X); there is no CPU fallback")
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("dtype must be float32 (reference mode) or bfloat16")
        self.lib = L.lib()
        self.cfg, self.dtype, self.device = cfg, dtype, torch.device(device)
        self.max_batch, self.max_boxes = max_batch, max_boxes
        torch.cuda.set_device(self.device)
        self.weights = repack_layout_weights(cfg, state_dict, dtype, self.device)
        e, d = cfg.encoder, cfg.decoder
        self.is_table = is_table_config(cfg)
        self.tok_width = 10 if self.is_table else 7
        # rows of the class-logit output: layout = label_count; table = the non-bbox property heads side by side
        self.label_count = sum(n for k, n in d.head_widths() if k != "bbox") if self.is_table else d.label_count
        c = LayoutConfigC(img_h=e.image_size[0], img_w=e.image_size[1], patch=e.patch_size, embed_dim=e.embed_dim, n_stages=len(e.depths),
                          window=e.window_size, enc_eps=e.layer_norm_eps, encoder_length=e.encoder_length, dec_layers=d.num_hidden_layers,
                          dec_hidden=d.hidden_size, dec_inter=d.intermediate_size, dec_heads=d.num_attention_heads,
                          dec_kv_heads=d.num_key_value_heads, vocab=d.vocab_size, label_count=self.label_count, bbox_size=d.bbox_size,
                          rms_eps=d.rms_norm_eps, ln_eps=d.layer_norm_eps, max_batch=max_batch, max_boxes=max_boxes,
                          dtype=L.DTYPE_F32 if dtype == torch.float32 else L.DTYPE_BF16,
                          family=FAMILY_TABLE if self.is_table else FAMILY_LAYOUT, box_embed=d.box_embed_size if self.is_table else d.hidden_size,
                          category_count=d.category_count if self.is_table else 0, merge_count=d.merge_count if self.is_table else 0)
        for i, (dep, nh, nkv) in enumerate(zip(e.depths, e.num_heads, e.num_kv_heads)):
            c.depths[i], c.heads[i], c.kv_heads[i] = dep, nh, nkv
        self.c = c
        table = (C.c_void_p * len(self.weights))(*[t.data_ptr() for t in self.weights])
        self.handle = C.c_void_p()
        L.check(self.lib.surya_layout_create(C.byref(c), table, len(self.weights), C.byref(self.handle)), "surya_layout_create")
        self._cls = np.zeros((max_batch, self.label_count), np.float32)
        self._box = np.zeros((max_batch, 6), np.float32)
        self.batch = self.encoded = 0

    def __del__(self):
        h = getattr(self, "handle", None)
        if h:
            self.lib.surya_layout_destroy(h)
            self.handle = None

    @property
    def config(self):
        return self.cfg

    def eval(self):
        return self

    def to(self, device_dtype=None):
        return self

    @property
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def encode(self, pixel_values: torch.Tensor):
        """pixel_values: cuda fp32 [B, 3, H, W]; keeps the encoder states and the cross-attention K / V inside the handle."""
        assert pixel_values.is_cuda and pixel_values.dtype == torch.float32 and pixel_values.is_contiguous()
        B = pixel_values.shape[0]
        assert tuple(pixel_values.shape[1:]) == (3,) + tuple(self.cfg.encoder.image_size) and B <= self.max_batch
        L.check(self.lib.surya_layout_encode(self.handle, L.ptr(pixel_values), C.c_int(B), self._stream), "surya_layout_encode")
        self.batch = self.encoded = B

    def encode_host(self, pixel_values: torch.Tensor):
        """encode() for a CPU fp32 tensor: pinned staging + asynchronous upload on the current stream."""
        self.encode(pixel_values.float().pin_memory().to(self.device, non_blocking=True).contiguous())

    def select(self, src_index):
        """Re-batch the decoder: row i of the following decode steps cross-attends the encoder states of image src_index[i] of the
        last encode() (table recognition decodes every ROW of a table against its image, table_rec/__init__.py:196-230)."""
        idx = np.ascontiguousarray(src_index, np.int32)
        assert idx.ndim == 1 and 0 < idx.size <= self.max_batch
        L.check(self.lib.surya_layout_select(self.handle, L.np_ptr(idx), C.c_int(idx.size)), "surya_layout_select")
        self.batch = int(idx.size)

    MAX_PROMPT = 64

    def prefill(self, boxes: np.ndarray):
        """boxes: int32 [B, T, token width], the decoder prompt of every row (T <= MAX_PROMPT); one pass over all T positions
        (surya_layout_prefill). Returns (class_logits, bbox) of the LAST prompt token; decode steps continue at position T.
        Falls back to T decode steps when the prompt does not fit the engine's borrowed workspaces."""
        b = np.ascontiguousarray(boxes, np.int32).reshape(self.batch, -1, self.tok_width)
        T = b.shape[1]
        if T <= self.MAX_PROMPT:
            rc = self.lib.surya_layout_prefill(self.handle, L.np_ptr(b), C.c_int(self.batch), C.c_int(T), L.np_ptr(self._cls, C.c_float),
                                               L.np_ptr(self._box, C.c_float), self._stream)
            if rc == L.SA_OK:
                return self._cls[: self.batch].copy(), self._box[: self.batch].copy()
            if rc != L.SA_ERR_ARG:                           # SA_ERR_ARG = the prompt does not fit the borrowed workspaces: step by step below
                L.check(rc, "surya_layout_prefill")
        out = None
        for t in range(T):
            out = self.decode_step(b[:, t], t)
        return out

    def decode_step(self, boxes: np.ndarray, position: int):
        """boxes: int32 [B, 7] (table family: [B, 10]); returns (class_logits [B, label_count], bbox [B, 6]) numpy copies."""
        b = np.ascontiguousarray(boxes, np.int32).reshape(self.batch, self.tok_width)
        L.check(self.lib.surya_layout_decode_step(self.handle, L.np_ptr(b), C.c_int(self.batch), C.c_int(position),
                                                  L.np_ptr(self._cls, C.c_float), L.np_ptr(self._box, C.c_float), self._stream),
                "surya_layout_decode_step")
        return self._cls[: self.batch].copy(), self._box[: self.batch].copy()

    # ---------------------------------------------------------------------------------------------- device-fed decode steps (round 4)
    def set_feedback(self, page_sizes=None):
        """The constants of the fed-back token rule and, for the layout family, each row's slice size [(width, height)] for the
        PageHeader / PageFooter rule (None = rule off). After encode() / select(), before the first decode_steps()."""
        d = self.cfg.decoder
        fb = LayoutFeedbackC()
        if self.is_table:
            widths = [n for k, n in d.head_widths() if k != "bbox"]
            fb.head_widths[:] = widths
            fb.relabel_ids[:] = [-1, -1]
        else:
            from .config import ID_TO_LABEL
            ids = {v: k for k, v in ID_TO_LABEL.items()}
            fb.skew_scaler = d.skew_scaler
            fb.relabel_ids[:] = [ids["PageHeader"] + d.special_token_count, ids["PageFooter"] + d.special_token_count]
        keep = None
        if page_sizes is not None and not self.is_table:
            keep = np.ascontiguousarray(page_sizes, np.int32).reshape(self.batch, 2)
            fb.page_sizes = L.np_ptr(keep)
        L.check(self.lib.surya_layout_set_feedback(self.handle, C.byref(fb), C.c_int(self.batch), self._stream), "surya_layout_set_feedback")

    def decode_steps(self, boxes, position: int, n_steps: int, ring: int = 0):
        """Enqueue n_steps (<= RING_STEPS) decode steps from cache position `position` whose fed-back tokens stay on the device; boxes
        int32 [B, token width] feeds the first of them from the host, None continues from the previous run. Records go to ring `ring`."""
        b = None
        if boxes is not None:
            b = np.ascontiguousarray(boxes, np.int32).reshape(self.batch, self.tok_width)
        L.check(self.lib.surya_layout_decode_steps(self.handle, L.np_ptr(b) if b is not None else None, C.c_int(self.batch), C.c_int(position),
                                                   C.c_int(n_steps), C.c_int(ring), self._stream), "surya_layout_decode_steps")

    def wait_steps(self, n_steps: int, ring: int = 0):
        """(class_logits [n, B, label_count], bbox [n, B, 6], fed_tokens [n, B, token width]) of the run enqueued into ring `ring`."""
        cls = np.empty((n_steps, self.batch, self.label_count), np.float32)
        box = np.empty((n_steps, self.batch, 6), np.float32)
        tok = np.empty((n_steps, self.batch, self.tok_width), np.int32)
        L.check(self.lib.surya_layout_wait_steps(self.handle, C.c_int(ring), C.c_int(self.batch), C.c_int(n_steps), L.np_ptr(cls, C.c_float),
                                                 L.np_ptr(box, C.c_float), L.np_ptr(tok)), "surya_layout_wait_steps")
        return cls, box, tok

    def encoder_states(self) -> torch.Tensor:
        e = self.cfg.encoder
        n = (e.grid[0] >> (len(e.depths) - 1)) * (e.grid[1] >> (len(e.depths) - 1))
        out = torch.empty((self.encoded, n, e.hidden_size), dtype=self.dtype, device=self.device)
        L.check(self.lib.surya_layout_encoder_states(self.handle, L.ptr(out), C.c_int(self.encoded), self._stream), "surya_layout_encoder_states")
        return out


class FedRuns:
    """Step-at-a-time view of the device-fed decode runs for the predictors' greedy loops (layout/predictor.py, table_rec/predictor.py).

    The loops keep the reference's shape -- one model call per emitted box, the next token derived on the host from that call's
    outputs -- but the model calls are served from runs of `k` steps that the device feeds itself (surya_layout_decode_steps), the run
    after the current one already enqueued while the host works through this one. `step(nxt)` takes the token the HOST derived for
    the step and returns that step's (class_logits, bbox); it checks the host's token against the one the device fed (they are the
    same rule implemented twice: a mismatch would mean the recorded outputs belong to a different token stream, so it raises).

    `model` is a HipLayoutModel (or the CPU oracle stand-in of the tests with the same three methods)."""

    def __init__(self, model, position: int, max_steps: int, k: int = 8):
        self.m, self.pos, self.left, self.k = model, position, max_steps, max(1, min(int(k), RING_STEPS))
        self.buf = None            # (cls, box, tok) of the run being consumed
        self.cur = 0
        self.pending = None        # (ring, n) of the run enqueued behind it
        self.ring = 0
        self.fed = None            # the token the device fed to the step that comes next

    def _enqueue(self, boxes):
        n = min(self.k, self.left)
        if n <= 0:
            return None
        self.m.decode_steps(boxes, self.pos, n, self.ring)
        run = (self.ring, n)
        self.pos += n
        self.left -= n
        self.ring ^= 1
        return run

    def step(self, nxt: np.ndarray, live: np.ndarray | None = None):
        """`live` (optional bool [rows]): the rows whose stream the host still follows. A finished row keeps being decoded (the batch
        is dense) and the device keeps applying its token rule to it, while the host -- like the reference, which `continue`s on done
        rows before the PageHeader / PageFooter rule (surya/layout/__init__.py:150-157) -- feeds it the raw class. The two tokens of
        such a row may differ and nobody reads either: rows are independent sequences, so only live rows are compared."""
        nxt = np.ascontiguousarray(nxt, np.int32)
        if self.buf is None or self.cur >= self.buf[0].shape[0]:
            if self.buf is None:                                  # the first step: fed from the host
                run = self._enqueue(nxt)
            else:
                run = self.pending
            if run is None:
                raise L.SuryaAmdError("FedRuns.step: more steps requested than max_steps")
            self.pending = self._enqueue(None)                    # keep the device busy while the host works through `run`
            self.buf = self.m.wait_steps(run[1], run[0])
            self.cur = 0
        diff = None
        if self.fed is not None:
            diff = self.fed != nxt
            if live is not None:
                diff = diff & np.asarray(live, bool).reshape((-1,) + (1,) * (diff.ndim - 1))
        if diff is not None and diff.any():
            bad = np.argwhere(diff)[0]
            raise L.SuryaAmdError(f"device-fed token differs from the host's rule at row {int(bad[0])}, component {int(bad[1])}: "
                                  f"device {self.fed[bad[0]].tolist()} vs host {nxt[bad[0]].tolist()}")
        cls, box, tok = self.buf
        i = self.cur
        self.cur += 1
        self.fed = tok[i]
        return cls[i], box[i]
