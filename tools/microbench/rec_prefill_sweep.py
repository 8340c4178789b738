"""Recognition prefill (encoder + decoder prefill of 256 bench-shaped lines, REC-FULL bf16) under launch-policy knobs, one process:
ms per prefill. `python tools/microbench/rec_prefill_sweep.py [key=value ...]`"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from surya_amd import _lib as L
from surya_amd.config import rec_config
from surya_amd.recognition.model import HipRecModel
from surya_amd.synth import make_rec_weights
from util import make_prompts, crop_grid

lib = L.lib()
cfg = rec_config("REC-FULL")
n = 256
m = HipRecModel(cfg, make_rec_weights(cfg, 0), image_token_id=cfg.image_token_id, pad_token_id=cfg.pad_token_id, eos_token_id=cfg.eos_token_id,
                dtype=torch.bfloat16, max_slots=n, max_kv_len=160, max_patches=65536, max_prefill_tokens=n * 72)
rng = np.random.default_rng(1234)
grids = [crop_grid(64, int(w)) for w in sorted(rng.integers(128, 513, size=n), reverse=True)]
tiles, seqs = make_prompts(cfg, grids, seed=3)
tiles = tiles.cuda().contiguous()
slots = list(range(n))


def run(reps=4):
    m.prefill(tiles, grids, seqs, slots)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        m.prefill(tiles, grids, seqs, slots)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


variants = [dict(), dict(persist=1), dict(bigtile=0), dict()] + [dict(**{k: int(v)}) for k, v in
                                                                                             (a.split("=") for a in sys.argv[1:])]
base = dict(bigtile=1, glds=2, bigtile_min_k=0, persist=0)
for v in variants:
    for k, val in {**base, **v}.items():
        L.check(lib.surya_set_tuning(k.encode(), C.c_int(val)), k)
    ms = min(run() for _ in range(3))
    print(f"{str(v):40s} {ms:7.3f} ms per prefill of {n} lines", flush=True)
