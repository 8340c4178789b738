"""Host profile of DetectionPredictor.__call__ on 16 US-letter pages (1275 x 1650: every page needs the double LANCZOS resize to the
processor size). `python tools/hostbench/det_letter_profile.py` on a GPU box: cProfile, top cumulative entries + wall clock."""
import cProfile
import os
import pstats
import sys
import time

import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from surya_amd.config import det_config
from surya_amd.detection.predictor import DetectionModelLoader, DetectionPredictor
from surya_amd.synth import make_det_weights, make_pages

cfg = det_config("DET-DEFAULT")


class Loader(DetectionModelLoader):
    def model(self, device=None, dtype=None, max_batch=None):
        return super().model("cuda:0", torch.bfloat16, max_batch=16)


class Pred(DetectionPredictor):
    model_loader_cls = Loader
    batch_size = 16


pred = Pred(checkpoint={"config": cfg, "state_dict": make_det_weights(cfg, 0), "size": 1024})
letter = [Image.fromarray(p).resize((1275, 1650), Image.Resampling.BILINEAR) for p in make_pages(16, 1024, seed=3)]
for _ in range(2):
    pred(letter)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    pred(letter)
torch.cuda.synchronize()
print(f"wall {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms per 16 letter pages")
pr = cProfile.Profile()
pr.enable()
pred(letter)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr, stream=sys.stdout).sort_stats("cumulative").print_stats(30)
