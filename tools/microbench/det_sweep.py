"""Detection forward A/B inside one process: DET-DEFAULT, 16 pages 1024^2, bf16, ms per forward under launch-policy knobs
(surya_set_tuning). `python tools/microbench/det_sweep.py`"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from surya_amd import _lib as L
from surya_amd.config import det_config
from surya_amd.detection.model import HipDetModel
from surya_amd.synth import make_det_weights

lib = L.lib()
cfg = det_config("DET-DEFAULT")
m = HipDetModel(cfg, make_det_weights(cfg, 0), height=1024, width=1024, dtype=torch.bfloat16, max_batch=16)
x = torch.randn(16, 3, 1024, 1024, device="cuda")


def run(n=8):
    for _ in range(2):
        m.forward(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        m.forward(x)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


variants = [dict(), dict(bigtile=0), dict(bigtile_min_k=320), dict(bigtile_min_k=576), dict(bigtile_min_k=1100), dict(bigtile_min_k=2100)] + [dict(**{k: int(v)}) for k, v in
                                                                                 (a.split("=") for a in sys.argv[1:])]
base = dict(bigtile=1, glds=2, bigtile_min_k=0)
for v in variants:
    for k, val in {**base, **v}.items():
        L.check(lib.surya_set_tuning(k.encode(), C.c_int(val)), k)
    ms = min(run() for _ in range(3))
    print(f"{str(v):40s} {ms:7.3f} ms / 16 pages = {16e3 / ms:7.1f} pages/s", flush=True)
