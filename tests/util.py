"""Shared test helpers: synthetic prompts at the post-processor boundary (image_tiles, grid_thw, input_ids)."""
from __future__ import annotations

import numpy as np
import torch

from surya_amd.config import RecConfig


def crop_grid(h: int, w: int):
    """Patch grid of a crop after scale_to_fit + round-up to multiples of 28 (processor/__init__.py:141-230)."""
    import math
    if h * w < 168 * 168:
        s = (168 * 168 / (h * w)) ** 0.5
        w, h = math.ceil(w * s), math.ceil(h * s)
    hb, wb = math.ceil(h / 28) * 28, math.ceil(w / 28) * 28
    return hb // 14, wb // 14


def make_prompts(cfg: RecConfig, grids, seed: int = 5, task_bos: str = "<OCR-WB>"):
    """Random-normal tiles + prompt ids [IMAGE]*n + REG1..4 + BOS + EOI for each (gh, gw) grid."""
    g = torch.Generator().manual_seed(seed)
    P = sum(h * w for h, w in grids)
    tiles = torch.randn(P, cfg.encoder.patch_dim, generator=g)
    seqs = []
    for h, w in grids:
        n = h * w // 4
        seqs.append([cfg.image_token_id] * n + [cfg.token_id(f"<REG{i}>") for i in range(1, 5)]
                    + [cfg.token_id(task_bos), cfg.token_id("<EOI>")])
    return tiles, seqs


def left_pad_batch(cfg: RecConfig, seqs):
    """Reference-style left padded batch + mask + position ids (processor/__init__.py:386-403)."""
    S = max(len(s) for s in seqs)
    pad = cfg.pad_token_id
    ids = torch.tensor([[pad] * (S - len(s)) + list(s) for s in seqs], dtype=torch.long)
    am = ids.ne(pad)
    pos = am.cumsum(-1) - 1
    pos[pos < 0] = 0
    pos = am.long() * pos
    return ids, am.long(), pos
