"""GPU: BASELINE.json configs[4] -- LaTeX OCR is RecognitionPredictor with task block_without_boxes (surya/recognition/
__init__.py:97-101, surya/scripts/ocr_latex.py:22-31, benchmark/texify.py:45-48): 384 x 384 equation crops, whole-image bbox,
img_size budget 1024 x 512, a 768-token horizon (prompt of 196 image tokens + 6, KV growing to ~970 rows).

fp32 mode: greedy token ids of every crop equal the oracle's over a long horizon (the decode attention walks many KV tiles,
prefill attention runs 202-token causal segments); predictor call shape and output schema of the texify callers."""
import numpy as np
import pytest
import torch
from PIL import Image

from oracle import rec_oracle as ro
from surya_amd.config import rec_config
from surya_amd.settings import settings
from surya_amd.synth import make_rec_weights
from util import left_pad_batch

pytestmark = pytest.mark.gpu


def equation_crops(n, size=384, seed=77):
    rng = np.random.default_rng(seed)
    crops = []
    for _ in range(n):
        img = np.full((size, size, 3), 255, np.uint8)
        for _ in range(int(rng.integers(12, 40))):
            x, y = int(rng.integers(10, size - 60)), int(rng.integers(10, size - 30))
            w, h = int(rng.integers(4, 50)), int(rng.integers(2, 24))
            img[y:y + h, x:x + w] = rng.integers(0, 90, size=3, dtype=np.uint8)
        crops.append(img)
    return crops


def test_texify_block_without_boxes_long_horizon_fp32(hip_lib):
    from surya_amd.recognition.predictor import RecognitionPredictor, RecognitionModelLoader
    from surya_amd.recognition.schema import TaskNames
    cfg = rec_config("REC-SMALL")
    sd = make_rec_weights(cfg, 0)
    T = 160

    class Loader(RecognitionModelLoader):
        def model(self, device=None, dtype_=None, **caps):
            return super().model("cuda:0", torch.float32, max_slots=4, max_kv_len=202 + T + 32, max_patches=4096, max_prefill_tokens=1024)

    class Pred(RecognitionPredictor):
        model_loader_cls = Loader
        batch_size = 4

    old = settings.RECOGNITION_MAX_TOKENS
    settings.RECOGNITION_MAX_TOKENS = T
    try:
        pred = Pred(checkpoint={"config": cfg, "state_dict": sd})
        crops = equation_crops(5)
        images = [Image.fromarray(c) for c in crops]
        tasks = [TaskNames.block_without_boxes] * len(images)
        bboxes = [[[0, 0, im.width, im.height]] for im in images]
        out = pred(images, tasks, bboxes=bboxes)                      # the texify callers' call shape
        assert len(out) == 5 and all(len(r.text_lines) == 1 for r in out)
        assert all(isinstance(r.text_lines[0].text, str) for r in out)
        # token-level check against the oracle on the predictor's own prompts
        flat = pred.slice_bboxes(images, task_names=tasks, bboxes=bboxes)
        prep = pred.prepare_lines(flat, math_mode=True)
        assert all(g == (28, 28) for g in prep["grids"]) and all(len(p) == 196 + 6 for p in prep["prompt_ids"])
        toks, _, _ = pred.generate(prep, 4)
        tiles = prep["tiles"].cpu()
        om = ro.OracleRecModel(cfg, sd, cfg.image_token_id)
        ids, am, pos = left_pad_batch(cfg, prep["prompt_ids"])
        ref, _, _, _ = ro.generate(om, ids, tiles, [(1, 28, 28)] * 5, am, pos, T, cfg.eos_token_id, cfg.pad_token_id, cfg.nop_token_id)
        for i in range(5):
            assert toks[i] == ref[i], (i, len(toks[i]), len(ref[i]))
        assert max(len(t) for t in toks) >= 40                      # the horizon was really walked (repeat rule fires at >= 40)
    finally:
        settings.RECOGNITION_MAX_TOKENS = old
