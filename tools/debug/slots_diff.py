"""What differs between recognition_batch_size = 256 and > 256 on one engine (REC-FULL bf16, conditioned weights)? tokens / scores / boxes."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from surya_amd.config import rec_config
from surya_amd.settings import settings
from surya_amd.synth import make_rec_weights, make_line_crops
from surya_amd.recognition.predictor import RecognitionPredictor, RecognitionModelLoader
from surya_amd.recognition.schema import TaskNames

cfg = rec_config("REC-FULL")
sd = make_rec_weights(cfg, 0, recipe="conditioned")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 384
N = int(sys.argv[2]) if len(sys.argv) > 2 else 330

class Loader(RecognitionModelLoader):
    def model(self, device=None, dtype_=None, **caps):
        return super().model("cuda:0", torch.bfloat16, max_slots=S, max_kv_len=192, max_patches=S * 260, max_prefill_tokens=S * 72)

class Pred(RecognitionPredictor):
    model_loader_cls = Loader
    batch_size = S

settings.RECOGNITION_MAX_TOKENS = 24
pred = Pred(checkpoint={"config": cfg, "state_dict": sd})
crops = [c.astype(np.float32) for c in make_line_crops(N, seed=11)]
crops.sort(key=lambda c: -c.shape[1])
flat = {"slices": crops, "input_text": [None] * N, "task_names": [TaskNames.ocr_with_boxes] * N}
prep = pred.prepare_lines(flat, math_mode=True)
res = {}
for slots in (64, 256, S):
    toks, boxes, scores = pred.generate(prep, slots)
    res[slots] = ([list(t) for t in toks], boxes.numpy().copy(), [list(s) for s in scores])
for slots in (256, S):
    t0, b0, s0 = res[64]; t1, b1, s1 = res[slots]
    tl = sum(int(a != b) for a, b in zip(t0, t1))
    sl = sum(int(a != b) for a, b in zip(s0, s1))
    smax = max((abs(x - y) for a, b in zip(s0, s1) if len(a) == len(b) for x, y in zip(a, b)), default=0.0)
    print(f"slots {slots} vs 64: lines with different tokens {tl}/{N}, different scores {sl}/{N} (max |d score| {smax:.3e}), boxes equal {bool((b0 == b1).all())}")
    if tl:
        i = next(i for i, (a, b) in enumerate(zip(t0, t1)) if a != b)
        k = next(k for k, (x, y) in enumerate(zip(t0[i], t1[i])) if x != y)
        print("  first differing line", i, "at step", k, t0[i][:k + 2], t1[i][:k + 2])
