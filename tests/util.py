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


def bench_line_inputs(cfg: RecConfig, n: int, seed: int = 1234, pick=None, task: str = "ocr_with_boxes"):
    """The boundary tensors bench.py feeds the device loop, built on the host without a GPU: surya_amd.synth.make_line_crops
    (widest first, as RecognitionPredictor orders them) -> scale_to_fit -> SuryaOCRProcessor. Returns
    (tiles [sum P, 588] fp32 CPU, grids [(gh, gw)], prompt ids) for the lines in `pick` (default: all)."""
    from surya_amd.recognition.predictor import RecognitionModelLoader, RecognitionPredictor
    from surya_amd.synth import make_line_crops
    proc = RecognitionModelLoader({"config": cfg, "state_dict": {}}).processor()
    crops = make_line_crops(n, seed=seed)
    crops.sort(key=lambda c: -c.shape[1])
    pick = list(range(n)) if pick is None else list(pick)
    size = RecognitionPredictor.tasks[task]["img_size"]
    tiles, grids, seqs = [], [], []
    for i in pick:
        img = proc.scale_to_fit(crops[i].astype(np.float32), size)
        out = proc([{"task": task, "inputs": [{"type": "image", "image": img, "rotated": False},
                                              {"type": "text", "text": "", "math": True}]}])
        tiles.append(out["image_tiles"]); grids.append(tuple(int(x) for x in out["grid_hw"][0])); seqs.append(out["input_ids"][0])
    return torch.from_numpy(np.concatenate(tiles, 0)), grids, seqs
