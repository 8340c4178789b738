"""Host side of the MXFP8 decode weights (csrc/gemm_mx.h, include/surya_amd.h: surya_rec_set_mx_weights).

OCP Microscaling FP8: e4m3 elements with one shared E8M0 (power-of-two) scale per 32 consecutive input features of a row.
Scale rule (the same one the device kernels apply to activations, csrc/common.h mx_block_exp): the smallest power of two
that brings the block's absmax to <= 448, so nothing saturates; elements are rounded to nearest even. The reference has no
fp8 mode (its weights run in the checkpoint dtype, surya/recognition/loader.py:25-58); BASELINE.json configs[4] asks for it.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch

from . import _lib as L
from .config import RecConfig

BLOCK = 32


def quantize_mx(w: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """[rows, K] float -> (uint8 e4m3 [rows, K], uint8 e8m0 [rows, K / 32])."""
    rows, K = w.shape
    if K % BLOCK:
        raise ValueError("MXFP8 rows must be whole 32-element blocks")
    b = w.detach().float().cpu().reshape(rows, K // BLOCK, BLOCK)
    amax = b.abs().amax(-1)
    frac, ex = torch.frexp(amax)                       # amax = frac * 2^ex, frac in [0.5, 1); 448 = 0.875 * 2^9
    e = torch.where(frac <= 0.875, ex - 9, ex - 8)
    e = torch.where(amax > 0, e, torch.full_like(e, -127)).clamp_(-127, 127)
    q = torch.ldexp(b, -e.unsqueeze(-1)).to(torch.float8_e4m3fn)
    return q.view(torch.uint8).reshape(rows, K).contiguous(), (e + 127).to(torch.uint8).contiguous()


def tile_major_scales(s: torch.Tensor) -> torch.Tensor:
    """[rows, K / 32] -> the kernel layout [K / 128, rows, 4] (include/surya_amd.h: the 4 block scales of a row inside one
    128-wide K-tile form a dword, dwords of consecutive rows are contiguous)."""
    rows, nb = s.shape
    if nb % 4:
        raise ValueError("MXFP8 GEMM operands need K to be a multiple of 128")
    return s.reshape(rows, nb // 4, 4).permute(1, 0, 2).contiguous()


def dequantize_mx(q: torch.Tensor, s: torch.Tensor) -> torch.Tensor:
    rows, K = q.shape
    v = q.cpu().view(torch.float8_e4m3fn).float().reshape(rows, K // BLOCK, BLOCK)
    return torch.ldexp(v, (s.cpu().to(torch.int32) - 127).unsqueeze(-1)).reshape(rows, K)


def repack_rec_mx_weights(cfg: RecConfig, weights: List[torch.Tensor], device) -> List[torch.Tensor]:
    """MXFP8 twins of the decoder projections + lm_head, built from the kernel-layout table repack_rec_weights produced
    (so the fused / interleaved row order is already right). Order = SA_MX_* of include/surya_amd.h."""
    e, d = cfg.encoder, cfg.decoder
    base = L.RW_GLOBALS + e.depth * L.RE_COUNT
    out: List[torch.Tensor] = []

    def put(t):
        q, s = quantize_mx(t)
        out.append(q.to(device))
        out.append(tile_major_scales(s).to(device))

    for l in range(d.num_hidden_layers):
        b = base + l * L.RD_COUNT
        for k in (L.RD_QKV_W, L.RD_O_W, L.RD_GU_W, L.RD_DOWN_W):
            put(weights[b + k])
    put(weights[L.RW_LM_W])
    return out
