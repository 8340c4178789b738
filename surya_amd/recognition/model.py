"""HipRecModel: the recognition model behind libsurya_amd.so.

Python here is plumbing only (device memory via torch, stream handle, ctypes marshalling); all arithmetic runs
in the HIP library. There is no fallback: constructing this without the built library or without a GPU raises.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np
import torch

from .. import _lib as L
from ..config import RecConfig
from .weights import repack_rec_weights, pad64


class HipRecModel:
    def __init__(self, cfg: RecConfig, state_dict, *, image_token_id: int, pad_token_id: int, eos_token_id: int,
                 dtype: torch.dtype = torch.bfloat16, device="cuda:0", max_slots: int = 256, max_kv_len: int = 512,
                 max_patches: int = 65536, max_prefill_tokens: Optional[int] = None, broadcast_weights: bool = False,
                 process_group=None, decode_fp8: Optional[bool] = None):
        if not torch.cuda.is_available():
            raise L.SuryaAmdError("HipRecModel needs a GPU (MI355X); there is no CPU fallback")
        self.lib = L.lib()
        self.cfg = cfg
        self.device = torch.device(device)
        self.dtype = dtype
        torch.cuda.set_device(self.device)
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("dtype must be float32 (reference mode) or bfloat16")
        e, d = cfg.encoder, cfg.decoder
        self.max_slots = max_slots
        self.vocab = d.vocab_size
        if max_prefill_tokens is None:
            max_prefill_tokens = max_slots * 96
        if broadcast_weights:
            # rank 0 repacks; the other ranks (state_dict may be None there) receive the kernel-layout tensors over RCCL
            from .. import dist as sdist
            rank, _ = sdist.world_info(process_group)
            mine = repack_rec_weights(cfg, state_dict, dtype, self.device) if rank == 0 else None
            self.weights = sdist.share_weights(mine, self.device, src=0, group=process_group)
        else:
            self.weights = repack_rec_weights(cfg, state_dict, dtype, self.device)   # keeps tensors alive
        mask = 0
        for i in e.fullatt_block_indexes:
            mask |= 1 << i
        c = L.RecConfigC(
            enc_depth=e.depth, enc_hidden=e.hidden_size, enc_inter=e.intermediate_size,
            enc_inter_pad=pad64(e.intermediate_size), enc_heads=e.num_heads, patch_dim=e.patch_dim,
            patch_dim_pad=pad64(e.patch_dim), merge=e.spatial_merge_size,
            window_tokens=e.window_size // e.spatial_merge_size // e.patch_size, enc_out_hidden=e.out_hidden_size,
            fullatt_mask=mask, enc_eps=e.rms_norm_eps, vocab=d.vocab_size, dec_hidden=d.hidden_size,
            dec_inter=pad64(d.intermediate_size), dec_layers=d.num_hidden_layers, dec_heads=d.num_attention_heads,
            dec_kv_heads=d.num_key_value_heads, dec_head_dim=d.head_dim, dec_eps=d.rms_norm_eps,
            bbox_size=cfg.bbox_size, embed_multiplier=cfg.image_embed_encoding_multiplier,
            image_token_id=image_token_id, pad_token_id=pad_token_id, eos_token_id=eos_token_id, max_slots=max_slots,
            max_kv_len=max_kv_len, max_patches=max_patches, max_prefill_tokens=max_prefill_tokens,
            dtype=L.DTYPE_F32 if dtype == torch.float32 else L.DTYPE_BF16)
        self.c = c
        table = (C.c_void_p * len(self.weights))(*[t.data_ptr() for t in self.weights])
        self.handle = C.c_void_p()
        L.check(self.lib.surya_rec_create(C.byref(c), table, len(self.weights), C.byref(self.handle)), "surya_rec_create")
        self._tok = np.zeros((L.SA_MAX_STEPS, max_slots), np.int32)
        self._score = np.zeros((L.SA_MAX_STEPS, max_slots), np.float32)
        self._bbox = np.zeros((L.SA_MAX_STEPS, max_slots, 6), np.int32)
        self.mx_weights = None
        self.decode_fp8 = False
        if decode_fp8 is None:
            from ..settings import settings
            decode_fp8 = settings.RECOGNITION_DECODE_FP8
        if decode_fp8:
            self.set_decode_fp8(True)
        self.kv_fp8 = False
        from ..settings import settings as _settings
        if _settings.RECOGNITION_KV_FP8:
            self.set_kv_fp8(True)

    def set_decode_fp8(self, on: bool):
        """Decode steps on MXFP8 weights and activations (surya_rec_set_mx_weights; bf16 models only). The e4m3 / e8m0 twins
        of the decoder projections and lm_head are quantised once from the kernel-layout bf16 table and stay resident next to
        it (prefill keeps using the bf16 weights)."""
        if on and self.dtype != torch.bfloat16:
            raise ValueError("the fp8 decode path exists for bfloat16 models only")
        if on:
            if self.mx_weights is None:
                from ..mx import repack_rec_mx_weights
                self.mx_weights = repack_rec_mx_weights(self.cfg, self.weights, self.device)
            table = (C.c_void_p * len(self.mx_weights))(*[t.data_ptr() for t in self.mx_weights])
            L.check(self.lib.surya_rec_set_mx_weights(self.handle, table, C.c_int(len(self.mx_weights))), "surya_rec_set_mx_weights")
        else:
            L.check(self.lib.surya_rec_set_mx_weights(self.handle, None, C.c_int(0)), "surya_rec_set_mx_weights")
        torch.cuda.synchronize(self.device)
        self.decode_fp8 = bool(on)

    def set_kv_fp8(self, on: bool):
        """Decode steps on an FP8 (e4m3 + one power-of-two scale per token and kv head) KV cache (surya_rec_set_kv_fp8,
        csrc/decode_attn_kv8.h; bf16 models only). Prefill still attends over the bf16 cache; lines prefilled after the call decode
        from the fp8 arrays. Switch while no line is in flight."""
        if on and self.dtype != torch.bfloat16:
            raise ValueError("the fp8 KV cache exists for bfloat16 models only")
        L.check(self.lib.surya_rec_set_kv_fp8(self.handle, C.c_int(1 if on else 0)), "surya_rec_set_kv_fp8")
        self.kv_fp8 = bool(on)

    def __del__(self):
        h = getattr(self, "handle", None)
        if h:
            self.lib.surya_rec_destroy(h)
            self.handle = None

    @property
    def config(self):
        """The reference reads `model.config.bbox_size` (recognition/__init__.py:315) and friends off the module."""
        return self.cfg

    def to(self, device_dtype=None):
        """BasePredictor.to calls model.to(...) (surya/common/predictor.py:31-35). Weights, KV slots and workspaces of a
        handle live on the device it was created on, in the dtype it was created with; asking for the same placement is
        a no-op, anything else needs a new handle (construct a new predictor) -- refuse loudly rather than pretend."""
        if device_dtype is None:
            return self
        want_dev, want_dt = None, None
        if isinstance(device_dtype, torch.dtype):
            want_dt = device_dtype
        else:
            want_dev = torch.device("cuda:0" if device_dtype == "cuda" else device_dtype)
        if (want_dt is not None and want_dt != self.dtype) or (want_dev is not None and want_dev != self.device):
            raise NotImplementedError(f"HipRecModel lives on {self.device} as {self.dtype}; create a new predictor for "
                                      f"{device_dtype} (the HIP handle owns device memory; there is no CPU path)")
        return self

    def eval(self):
        return self

    @property
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ---------------------------------------------------------------------------------------------- calls
    def prefill(self, tiles: torch.Tensor, grid_hw, input_ids: Sequence[Sequence[int]], slot_ids: Sequence[int]):
        """tiles: cuda fp32 [P, patch_dim], or None to consume embeddings queued by encode_ahead; grid_hw: [n_images, 2];
        input_ids: per-sequence prompt ids."""
        torch.cuda.set_device(self.device)
        grid = np.ascontiguousarray(np.asarray(grid_hw, np.int32).reshape(-1, 2))
        offs = np.zeros(len(input_ids) + 1, np.int32)
        offs[1:] = np.cumsum([len(s) for s in input_ids])
        flat = np.ascontiguousarray(np.concatenate([np.asarray(s, np.int32) for s in input_ids]))
        slots = np.ascontiguousarray(np.asarray(slot_ids, np.int32))
        if tiles is not None:
            assert tiles.is_cuda and tiles.dtype == torch.float32 and tiles.is_contiguous()
            assert tiles.shape[0] == int((grid[:, 0] * grid[:, 1]).sum()) and tiles.shape[1] == self.cfg.encoder.patch_dim
        L.check(self.lib.surya_rec_prefill(self.handle, L.ptr(tiles), L.np_ptr(grid), C.c_int(len(grid)), L.np_ptr(flat),
                                           L.np_ptr(offs), L.np_ptr(slots), C.c_int(len(slots)), self._stream),
                "surya_rec_prefill")

    def encode_ahead(self, tiles: torch.Tensor, grid_hw):
        """Encode the images of the next prompts on the library's low-priority stream (overlaps with decode steps); the
        following prefill(None, ...) calls consume the embeddings in this order."""
        torch.cuda.set_device(self.device)
        grid = np.ascontiguousarray(np.asarray(grid_hw, np.int32).reshape(-1, 2))
        assert tiles.is_cuda and tiles.dtype == torch.float32 and tiles.is_contiguous()
        assert tiles.shape[0] == int((grid[:, 0] * grid[:, 1]).sum()) and tiles.shape[1] == self.cfg.encoder.patch_dim
        L.check(self.lib.surya_rec_encode_ahead(self.handle, L.ptr(tiles), L.np_ptr(grid), C.c_int(len(grid)), self._stream),
                "surya_rec_encode_ahead")

    def discard_ahead(self):
        """Forget look-ahead embeddings nobody consumed (a loop that ended early, e.g. on an exception): the next encode_ahead starts clean."""
        L.check(self.lib.surya_rec_encode_ahead(self.handle, None, None, C.c_int(0), self._stream), "surya_rec_encode_ahead(discard)")

    def set_active(self, slots: Sequence[int]):
        a = np.ascontiguousarray(np.asarray(slots, np.int32))
        L.check(self.lib.surya_rec_set_active(self.handle, L.np_ptr(a), C.c_int(len(a)), self._stream), "surya_rec_set_active")

    def decode(self, n_steps: int = 1):
        L.check(self.lib.surya_rec_decode(self.handle, C.c_int(n_steps), self._stream), "surya_rec_decode")

    def read_outputs(self, n_steps: int = 1):
        """Sync; returns (tokens [n_steps, slots], scores, bboxes [n_steps, slots, 6]) numpy views."""
        L.check(self.lib.surya_rec_read_outputs(self.handle, C.c_int(n_steps), L.np_ptr(self._tok),
                                                L.np_ptr(self._score, C.c_float), L.np_ptr(self._bbox), self._stream),
                "surya_rec_read_outputs")
        return self._tok[:n_steps], self._score[:n_steps], self._bbox[:n_steps]

    def decode_async(self, n_steps: int, ring: int):
        """Enqueue n_steps (<= SA_MAX_STEPS / 2) decode steps whose outputs land in ring half `ring`; no sync."""
        L.check(self.lib.surya_rec_decode_async(self.handle, C.c_int(n_steps), C.c_int(ring), self._stream), "surya_rec_decode_async")

    def wait_outputs(self, n_steps: int, ring: int):
        """Block until the decode_async call on `ring` finished (later calls may still be queued); copies of the outputs."""
        h = L.SA_MAX_STEPS // 2
        tok, sc, bb = self._tok[ring * h:], self._score[ring * h:], self._bbox[ring * h:]
        L.check(self.lib.surya_rec_wait_outputs(self.handle, C.c_int(n_steps), C.c_int(ring), L.np_ptr(tok), L.np_ptr(sc, C.c_float),
                                                L.np_ptr(bb)), "surya_rec_wait_outputs")
        return tok[:n_steps], sc[:n_steps], bb[:n_steps]

    def set_next_tokens(self, slots, tokens):
        s = np.ascontiguousarray(np.asarray(slots, np.int32))
        t = np.ascontiguousarray(np.asarray(tokens, np.int32))
        L.check(self.lib.surya_rec_set_next_tokens(self.handle, L.np_ptr(s), L.np_ptr(t), C.c_int(len(s)), self._stream),
                "surya_rec_set_next_tokens")

    def encode_only(self, tiles: torch.Tensor, grid_hw) -> torch.Tensor:
        grid = np.ascontiguousarray(np.asarray(grid_hw, np.int32).reshape(-1, 2))
        ntok = int((grid[:, 0] * grid[:, 1]).sum()) // (self.cfg.encoder.spatial_merge_size ** 2)
        out = torch.empty((ntok, self.cfg.decoder.hidden_size), dtype=self.dtype, device=self.device)
        L.check(self.lib.surya_rec_encode_only(self.handle, L.ptr(tiles), L.np_ptr(grid), C.c_int(len(grid)), L.ptr(out),
                                               self._stream), "surya_rec_encode_only")
        return out

    def last_logits(self) -> torch.Tensor:
        buf = torch.empty((self.max_slots, self.vocab), dtype=torch.float32, device=self.device)
        rows = C.c_int(0)
        L.check(self.lib.surya_rec_copy_last_logits(self.handle, L.ptr(buf), C.c_int(self.max_slots), C.byref(rows),
                                                    self._stream), "surya_rec_copy_last_logits")
        return buf[: rows.value]
