"""CPU, build container only: live cross-check of the oracle against the real reference modules imported from
/root/reference through oracle/ref_shim (skipped where the reference is absent, e.g. on the GPU box)."""
import pytest
import torch

from oracle import ref_shim

pytestmark = pytest.mark.skipif(not ref_shim.available(), reason="/root/reference not present")


def test_live_rec_small_head_dim_80_and_gqa():
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
    from oracle.make_golden import build_reference_rec
    from oracle import rec_oracle as ro
    from surya_amd.config import rec_config
    from surya_amd.synth import make_rec_weights
    from util import make_prompts, left_pad_batch
    from transformers import DynamicCache
    cfg = rec_config("REC-SMALL")
    sd = make_rec_weights(cfg, 0)
    ref = build_reference_rec(cfg, sd, "sdpa")
    grids = [(6, 38), (10, 18)]
    tiles, seqs = make_prompts(cfg, grids)
    ids, am, pos = left_pad_batch(cfg, seqs)
    with torch.inference_mode():
        out = ref(input_ids=ids, image_tiles=tiles, grid_thw=torch.tensor([(1, h, w) for h, w in grids]), attention_mask=am,
                  position_ids=pos, past_key_values=DynamicCache(), use_cache=True, logits_to_keep=1, encoder_chunk_size=4096)
    om = ro.OracleRecModel(cfg, sd, cfg.image_token_id)
    lm, bb = om.prefill(ids, tiles, [(1, h, w) for h, w in grids], am, pos)
    assert (lm - out.lm_logits).abs().max().item() <= 1e-4 * out.lm_logits.abs().max().item()
    assert (bb - out.bbox_logits).abs().max().item() <= 1e-5
    assert torch.equal(lm.argmax(-1), out.lm_logits.argmax(-1))


# ---------------------------------------------------------------------------------------------------------------------
# Host logic of the predictors against the reference's own functions on randomised inputs.
# These reference modules are plain Python; they import through oracle/ref_shim like the model modules above.

def _ref(modname):
    """Import a reference submodule WITHOUT running its package __init__ (surya/recognition/__init__.py imports names that
    transformers 5.x no longer has): the package is registered as a bare namespace pointing at the reference directory."""
    import importlib, os, sys, types
    ref_shim.install()
    pkg = modname.rsplit(".", 1)[0]
    if pkg not in sys.modules and pkg != "surya":
        importlib.import_module("surya")
        m = types.ModuleType(pkg)
        m.__path__ = [os.path.join(ref_shim.REFERENCE_ROOT, *pkg.split("."))]
        sys.modules[pkg] = m
    return importlib.import_module(modname)


def test_live_detect_repeat_token_random_streams():
    import random
    ru = _ref("surya.recognition.util")
    from surya_amd.recognition.postprocess import detect_repeat_token
    rng = random.Random(3)
    for _ in range(4000):
        n = rng.randint(0, 90)
        vocab = rng.choice([1, 2, 3, 5, 8, 50])
        toks = [rng.randrange(vocab) for _ in range(n)]
        if rng.random() < 0.5 and n > 10:                       # plant a periodic tail
            period = rng.randint(1, 6)
            tail = [rng.randrange(vocab) for _ in range(period)]
            toks = toks[: rng.randint(0, 20)] + tail * rng.randint(3, 30)
        assert detect_repeat_token(toks) == bool(ru.detect_repeat_token(toks)), toks


def test_live_prediction_to_polygon_batch_random():
    import numpy as np
    ru = _ref("surya.recognition.util")
    from surya_amd.recognition.postprocess import prediction_to_polygon_batch
    rng = np.random.default_rng(5)
    for _ in range(50):
        n, t = int(rng.integers(1, 6)), int(rng.integers(1, 12))
        pred = rng.integers(0, 1025, size=(n, t, 6)).astype(np.float32)
        sizes = [(int(rng.integers(20, 400)), int(rng.integers(20, 1200))) for _ in range(n)]
        ref = ru.prediction_to_polygon_batch(torch.from_numpy(pred.copy()), sizes, 1025, 512).numpy()
        got = prediction_to_polygon_batch(pred, sizes, 1025, 512)
        assert np.array_equal(got, ref)


def test_live_math_tag_cleaning_and_unwrap():
    import random
    ru = _ref("surya.recognition.util")
    from surya_amd.recognition import postprocess as pp
    rng = random.Random(11)
    pieces = ["<math>", "</math>", '<math display="block">', "x^2", " ", "a", "\\frac{1}{2}", "<b>", "</b>", "text", "\\(", "\\)",
              "<br>", "$", "1 + 1", "<math display='inline'>", "\n"]
    for _ in range(3000):
        s = "".join(rng.choice(pieces) for _ in range(rng.randint(0, 12)))
        assert pp.clean_math_tags(s) == ru.clean_math_tags(s), s
        assert pp.unwrap_math(s) == ru.unwrap_math(s), s


def test_live_clean_close_polygons_random():
    import random
    ru = _ref("surya.recognition.util")
    from surya_amd.recognition.postprocess import clean_close_polygons
    rng = random.Random(17)
    for _ in range(500):
        polys = []
        for _ in range(rng.randint(0, 8)):
            x, y = rng.randint(0, 100), rng.randint(0, 50)
            w, h = rng.choice([0, 1, 2, 5, 30]), rng.choice([0, 1, 3, 12])
            polys.append([[x, y], [x + w, y], [x + w, y + h], [x, y + h]])
            if rng.random() < 0.4:                               # near-duplicate of the previous box
                polys.append([[px + rng.choice([0, 0, 1]), py] for px, py in polys[-1]])
        assert clean_close_polygons([[[float(a), float(b)] for a, b in p] for p in polys]) == \
               ru.clean_close_polygons([[[a, b] for a, b in p] for p in polys])


def test_live_detection_split_rules():
    from PIL import Image
    du = _ref("surya.detection.util")
    from surya_amd.detection.predictor import get_total_splits, split_image
    import numpy as np
    for w, h in [(800, 600), (1000, 1400), (1000, 1401), (700, 3000), (333, 5001), (1200, 1200)]:
        for height in (512, 1024, 1200):
            assert get_total_splits((w, h), height) == du.get_total_splits((w, h), height)
            img = Image.fromarray((np.arange(w * h * 3, dtype=np.uint32) % 251).astype(np.uint8).reshape(h, w, 3))
            ours, ours_h = split_image(img, height)
            ref, ref_h = du.split_image(img, height)
            assert ours_h == ref_h and len(ours) == len(ref)
            for a, b in zip(ours, ref):
                assert a.size == b.size and np.array_equal(np.asarray(a), np.asarray(b))


def test_live_fix_unbalanced_tags_and_words_from_chars():
    import random
    rp = _ref("surya.recognition.postprocessing")
    ru = _ref("surya.recognition.util")
    rs = _ref("surya.recognition.schema")
    rpoly = _ref("surya.common.polygon")
    from surya_amd.recognition import postprocess as pp
    from surya_amd.recognition.schema import TextChar
    from surya_amd.common.geometry import PolygonBox
    special = {"formatting": ["<b>", "</b>", "<i>", "</i>", "<u>", "</u>", "<sup>", "</sup>", "<br>"],
               "math_external": ["<math>", "</math>"], "system": ["<S>"]}
    rng = random.Random(23)
    texts = ["a", "b", " ", "<b>", "</b>", "<i>", "</i>", "<math>", "</math>", "<br>", "<br/>", "<sup>", "</sup>", "<x>", "W", "é",
             '<math display="block">', "<u/>"]

    def poly(x):
        return [[x, 2.0], [x + 7.0, 2.0], [x + 7.0, 14.0], [x, 14.0]]

    for _ in range(600):
        n = rng.randint(0, 14)
        items = [(rng.choice(texts), rng.random() < 0.8, float(8 * i + rng.randint(0, 3))) for i in range(n)]
        ours = [TextChar(text=t, confidence=0.5, polygon=poly(x), bbox_valid=v) for t, v, x in items]
        refs = [rs.TextChar(text=t, confidence=0.5, polygon=poly(x), bbox_valid=v) for t, v, x in items]
        a = pp.fix_unbalanced_tags(list(ours), special)
        b = rp.fix_unbalanced_tags(list(refs), special)
        assert [(c.text, c.bbox_valid, c.polygon) for c in a] == [(c.text, c.bbox_valid, c.polygon) for c in b]
        line = [[0.0, 0.0], [8.0 * n + 20, 0.0], [8.0 * n + 20, 16.0], [0.0, 16.0]]
        wa = pp.words_from_chars(ours, PolygonBox(polygon=line))
        wb = ru.words_from_chars(refs, rpoly.PolygonBox(polygon=line))
        assert [(w.text, w.polygon, w.bbox_valid) for w in wa] == [(w.text, w.polygon, w.bbox_valid) for w in wb]


def test_live_polygon_box_semantics_random():
    import random
    rpoly = _ref("surya.common.polygon")
    from surya_amd.common.geometry import PolygonBox
    rng = random.Random(29)

    def rnd_poly():
        x, y = rng.uniform(0, 200), rng.uniform(0, 100)
        w, h = rng.uniform(1, 80), rng.uniform(1, 40)
        return [[x, y], [x + w, y + rng.uniform(-2, 2)], [x + w, y + h], [x, y + h + rng.uniform(-2, 2)]]

    for _ in range(400):
        pa, pb = rnd_poly(), rnd_poly()
        a, b = PolygonBox(polygon=[list(p) for p in pa]), PolygonBox(polygon=[list(p) for p in pb])
        ra, rb = rpoly.PolygonBox(polygon=[list(p) for p in pa]), rpoly.PolygonBox(polygon=[list(p) for p in pb])
        assert a.bbox == ra.bbox and a.area == ra.area and a.height == ra.height and a.width == ra.width
        assert a.intersection_area(b) == ra.intersection_area(rb)
        assert a.intersection_pct(b) == ra.intersection_pct(rb)
        assert a.center == ra.center
        a.merge(b); ra.merge(rb)
        assert a.polygon == ra.polygon
        a.rescale((200, 100), (400, 300)); ra.rescale((200, 100), (400, 300))
        assert a.polygon == ra.polygon
