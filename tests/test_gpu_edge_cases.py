"""GPU: edge cases of the predictors (the cases the reference's own tests and call sites exercise implicitly): extreme crop
geometries (a few pixels; far beyond the task's maximum area; tall; already a multiple of 28), pages without lines, a
one-token budget, more lines than KV slots, blank pages through both post-processing paths. Recognition results are compared
with the oracle's greedy tokens on the predictor's own pre-processed prompts (fp32 mode: bit-exact is the bar)."""
import numpy as np
import pytest
import torch
from PIL import Image

from surya_amd.config import det_config
from surya_amd.settings import settings
from surya_amd.synth import make_det_weights, make_line_crops, make_pages

from test_gpu_predictors import make_rec_predictor, oracle_tokens

pytestmark = pytest.mark.gpu


def test_extreme_crop_geometries_tokens_equal_oracle(hip_lib):
    cfg, sd, pred = make_rec_predictor(max_slots=4, max_tokens=8)
    rng = np.random.default_rng(7)
    shapes = [(3, 5), (1, 1), (20, 1500), (300, 40), (56, 112), (64, 513), (29, 27), (700, 900)]
    crops = [rng.integers(0, 256, size=(h, w, 3)).astype(np.float32) for h, w in shapes]
    crops.sort(key=lambda c: -c.shape[1])
    prep, ref = oracle_tokens(cfg, sd, pred, crops, 8)
    for (gh, gw) in prep["grids"]:
        assert gh % 2 == 0 and gw % 2 == 0 and gh >= 2 and gw >= 2              # whole merge blocks, never empty
    toks, boxes, scores = pred.generate(prep, 4)                                # 8 lines through 4 slots: refills
    assert [list(t) for t in toks] == ref
    assert all(len(s) == len(t) >= 1 for s, t in zip(scores, toks))


def test_one_token_budget_and_more_lines_than_slots(hip_lib):
    cfg, sd, pred = make_rec_predictor(max_slots=2, max_tokens=1)
    crops = [c.astype(np.float32) for c in make_line_crops(7, seed=9)]
    crops.sort(key=lambda c: -c.shape[1])
    prep, ref = oracle_tokens(cfg, sd, pred, crops, 1)
    toks, _, _ = pred.generate(prep, 2)
    # the reference admits a line with its prefill token and gives it one stop-rule step (recognition/__init__.py:583-595):
    # the first token is the oracle's, a line never exceeds prefill token + one step
    assert [t[0] for t in toks] == [r[0] for r in ref]
    assert all(1 <= len(t) <= 2 for t in toks)
    settings.RECOGNITION_MAX_TOKENS = 16


def test_pages_without_lines(hip_lib):
    cfg, sd, pred = make_rec_predictor(max_slots=4, max_tokens=6)
    pages = [Image.fromarray(p) for p in make_pages(3, 128, seed=4)]
    out = pred(pages, bboxes=[[], [[5, 5, 100, 40]], []])
    assert [len(r.text_lines) for r in out] == [0, 1, 0]
    assert out[0].image_bbox == [0, 0, 128, 128]
    none = pred(pages, bboxes=[[], [], []])
    assert none == [] or all(len(r.text_lines) == 0 for r in none)          # no slices at all: the reference returns early
    out = pred(pages[:1], polygons=[[[[5, 5], [100, 6], [101, 40], [4, 41]]]])
    assert len(out[0].text_lines) == 1 and out[0].text_lines[0].polygon == [[5.0, 5.0], [100.0, 6.0], [101.0, 40.0], [4.0, 41.0]]


@pytest.mark.parametrize("value", [0, 255, 128])
def test_blank_pages_device_and_host_postprocessing_agree(hip_lib, value):
    from surya_amd.detection.predictor import DetectionPredictor
    cfg = det_config("DET-TINY")
    sd = make_det_weights(cfg, 0)
    pages = [Image.fromarray(np.full((256, 256, 3), value, np.uint8)), Image.fromarray(make_pages(1, 256, seed=3)[0])]
    res = {}
    for host in (0, 1):
        pred = DetectionPredictor(checkpoint={"config": cfg, "state_dict": sd, "size": 256}, dtype=torch.float32)
        pred.device_postprocess = not host
        res[host] = pred(pages)
    for a, b in zip(res[0], res[1]):
        assert [x.polygon for x in a.bboxes] == [x.polygon for x in b.bboxes]
        assert np.allclose([x.confidence for x in a.bboxes], [x.confidence for x in b.bboxes], atol=1e-5)
    assert DetectionPredictor(checkpoint={"config": cfg, "state_dict": sd, "size": 256}, dtype=torch.float32)([]) == []
