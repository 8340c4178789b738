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


@pytest.mark.parametrize("seed,return_words", [(0, False), (1, True), (2, False)])
def test_live_output_assembly_against_reference_get_bboxes_text(seed, return_words):
    """SURVEY 8(a) R16 against the reference's OWN code: RecognitionPredictor.get_bboxes_text of /root/reference (imported through
    ref_shim.import_recognition, called as a plain function on a stand-in `self` that carries our processor / tokenizer) followed
    by the per-line tail of its __call__ (recognition/__init__.py:886-925: rescale / shift / clamp per character, tag fixing, math
    clean-up, TextLine) executed with the reference's own classes and helpers -- versus our batched assembly
    (RecognitionPredictor._assemble_batch) on the same random token streams: every field of every line, character and word."""
    import numpy as np
    from types import SimpleNamespace
    sr = ref_shim.import_recognition()
    from surya.common.polygon import PolygonBox as RefBox
    from surya.recognition.postprocessing import fix_unbalanced_tags as ref_fix
    from surya.recognition.util import clean_math_tags as ref_clean, unwrap_math as ref_unwrap, words_from_chars as ref_words, \
        prediction_to_polygon_batch as ref_polys
    from surya.recognition.schema import TextLine as RefLine
    from surya_amd.recognition.predictor import RecognitionPredictor
    from surya_amd.recognition.processor import SuryaOCRProcessor
    from surya_amd.recognition.tokenizer import ByteMathTokenizer, OCRTokenizer

    rng = np.random.default_rng(700 + seed)
    tok = OCRTokenizer(None, ByteMathTokenizer(256), reserve_special=64)
    proc = SuryaOCRProcessor(tok)
    ours = object.__new__(RecognitionPredictor)
    ours.processor = proc
    ref_self = SimpleNamespace(processor=proc, tasks=sr.RecognitionPredictor.tasks)
    sysm = tok.system_tokens
    specials = [v for k, v in tok.SPECIAL_TOKEN_MAPPING.items() if k not in sysm] + [sysm["<NO-MATH>"]]
    n, bbox_size = 48, 1025
    flat = {"polygons": [], "res_scales": [], "slices": [], "task_names": ["ocr_with_boxes"] * n}
    items, tokens_l, scores_l, rows_l = [], [], [], []
    for li in range(n):
        T = int(rng.integers(1, 46))
        toks, mode = [], rng.random()
        while len(toks) < T:
            r = rng.random() if mode < 0.6 else 0.0
            if r < 0.55:
                for ch in rng.choice(list("abc xyzÄ漢😀<>/"), size=int(rng.integers(1, 6))):
                    raw = ch.encode("utf-16le")
                    toks += [raw[i] + (raw[i + 1] << 8) + tok.special_token_offset for i in range(0, len(raw), 2)]
            elif r < 0.78:
                toks.append(int(rng.choice(specials)))
            elif r < 0.95:
                toks += [int(x) for x in rng.integers(32, 127, size=int(rng.integers(1, 5)))]
            else:
                toks.append(int(rng.choice([proc.eos_token_id, proc.pad_token_id])))
        toks = toks[:T]
        if li % 17 == 5:
            toks[int(rng.integers(0, T))] = proc.no_output_token
        rows = np.sort(rng.integers(0, bbox_size, size=(T, 6)), axis=0).astype(np.float32)
        for t in range(1, T):
            if rng.random() < 0.3:
                rows[t] = rows[t - 1]
        sc = rng.random(T).astype(np.float32).tolist()
        flat["polygons"].append([[10.5, 20.25], [400.0, 21.0], [401.0, 90.75], [11.0, 88.0]] if li % 2 else [7, 9, 300, 52])
        flat["res_scales"].append((1.0, 1.0) if li % 3 else (1.37, 2.2))
        flat["slices"].append(np.zeros((int(rng.integers(20, 80)), int(rng.integers(100, 600)), 3), np.uint8))
        items.append((li, li, toks, sc, rows))
        tokens_l.append(toks); scores_l.append(sc); rows_l.append(rows)

    got = ours._assemble_batch(flat, items, False, return_words, bbox_size)

    # the reference: box tokens -> polygons per line (its own prediction_to_polygon_batch, :603), get_bboxes_text, __call__'s tail
    import torch
    ref_lines = []
    for li in range(n):
        polys = ref_polys(torch.from_numpy(rows_l[li])[None], [flat["slices"][li].shape], bbox_size, bbox_size // 2)
        chars = sr.RecognitionPredictor.get_bboxes_text(ref_self, {"slices": [flat["slices"][li]], "task_names": ["ocr_with_boxes"]},
                                                        [tokens_l[li]], [scores_l[li]], polys)[0]
        polygon, res_scale = flat["polygons"][li], flat["res_scales"][li]
        if not chars:
            ref_lines.append(RefLine(text="", polygon=polygon, chars=[], confidence=1, original_text_good=True))
            continue
        confidence = float(np.mean([c.confidence for c in chars]))
        box = RefBox(polygon=polygon)
        for c in chars:
            c.rescale(res_scale, (1, 1)); c.shift(box.bbox[0], box.bbox[1]); c.clamp(box.bbox)
        chars = ref_fix(chars, tok.special_tokens)
        text = ref_clean(ref_unwrap("".join(c.text for c in chars)))
        ref_lines.append(RefLine(text=text, polygon=polygon, chars=chars, confidence=confidence,
                                 words=ref_words(chars, box) if return_words else []))
    assert len(got) == n
    for li, (g, r) in enumerate(zip(got, ref_lines)):
        gd, rd = g.model_dump(), r.model_dump()
        assert gd["text"] == rd["text"], li
        assert gd == rd, (li, gd, rd)


@pytest.mark.parametrize("n_lines,max_tokens,slots,sps,ahead", [(23, 12, 4, 4, True), (64, 20, 8, 2, False), (30, 64, 8, 4, True),
                                                                 (57, 33, 16, 8, True), (9, 45, 16, 1, True)])
def test_live_device_loop_against_reference_prediction_loop(n_lines, max_tokens, slots, sps, ahead):
    """SURVEY 8(a) R3 / R14 against the reference's OWN scheduler: RecognitionPredictor.prediction_loop of /root/reference
    (recognition/__init__.py:501-607) runs as it is -- its admission rule (min_prefill_ratio), slot table, stop rules and token
    budgets -- on a stand-in `self` whose prefill / decode return scripted tokens with the reference's own slot bookkeeping
    (:354-420: the first empty slots, queue order). Our device loop (RecognitionPredictor.generate: pipelined decode calls,
    look-ahead encoding, array-form bookkeeping) runs the same scripted lines through the contract-checking fake model of
    tests/test_scheduler_cpu.py. Tokens, scores and box rows of every line must be identical: the two schedulers may differ in
    WHEN a line is stepped, never in what it emits."""
    import numpy as np
    from collections import deque
    from types import SimpleNamespace
    import test_scheduler_cpu as ts
    from surya_amd.settings import settings as ours_settings
    sr = ref_shim.import_recognition()
    from surya.settings import settings as ref_settings
    R = sr.RecognitionPredictor

    class RefStub:
        prediction_loop = R.prediction_loop
        setup_cache = R.setup_cache
        num_empty_slots = R.num_empty_slots
        num_active_slots = R.num_active_slots
        min_prefill_ratio = R.min_prefill_ratio
        tasks = R.tasks
        disable_tqdm = True

        def __init__(self):
            self.kv_cache, self.prompt_queue, self.batch_prompt_mapping = None, deque(), None
            self.processor = SimpleNamespace(eos_token_id=ts.EOS, pad_token_id=ts.PAD, no_output_token=ts.NOP)
            self.pos = {}

        def get_batch_size(self):
            return slots

        def maybe_trim_cache_padding(self, x):
            return x

        def _out(self, rows):          # rows: [(line, t)] -> ContinuousBatchOutput-like (preds [n, 1], scores [n, 1], bbox [n, 1, 6])
            preds = torch.tensor([[ts.script(ln, t)] for ln, t in rows], dtype=torch.long)
            scores = torch.full((len(rows), 1), 0.25)
            bbox = torch.tensor([[[float(ln)] * 6] for ln, _ in rows])
            return SimpleNamespace(preds=preds, scores=scores, bbox_preds=bbox)

        def prefill(self, current_inputs=None):
            prompts = [self.prompt_queue.popleft() for _ in range(min(self.num_empty_slots, len(self.prompt_queue)))]
            empty = [k for k, v in self.batch_prompt_mapping.items() if v is None][: len(prompts)]
            for s, p in zip(empty, prompts):
                self.batch_prompt_mapping[s] = p.id
                self.pos[p.id] = 1
            return None, self._out([(p.id, 0) for p in prompts]), empty

        def decode(self, current_inputs=None):
            rows = []
            for s in range(slots):
                ln = self.batch_prompt_mapping[s]
                if ln is None:
                    rows.append((0, 0))          # inactive slots still produce a row; the loop ignores it
                else:
                    rows.append((ln, self.pos[ln])); self.pos[ln] += 1
            return None, self._out(rows)

    old_ref = ref_settings.RECOGNITION_MAX_TOKENS
    ref_settings.RECOGNITION_MAX_TOKENS = max_tokens
    try:
        flat = {"slices": [None] * n_lines, "input_text": [None] * n_lines, "task_names": ["ocr_with_boxes"] * n_lines}
        ref_tokens, ref_boxes, ref_scores = RefStub().prediction_loop(flat, slots)
    finally:
        ref_settings.RECOGNITION_MAX_TOKENS = old_ref

    old = (ours_settings.RECOGNITION_STEPS_PER_SYNC, ours_settings.RECOGNITION_ENCODE_AHEAD)
    ours_settings.RECOGNITION_STEPS_PER_SYNC, ours_settings.RECOGNITION_ENCODE_AHEAD = sps, ahead
    try:
        pred, prep = ts.make(n_lines, max_tokens, slots)
        prep["max_tokens"] = {i: max_tokens for i in range(n_lines)}            # the reference's budget is per task, not per line
        toks, boxes, scores = pred.generate(prep, slots)
    finally:
        ours_settings.RECOGNITION_STEPS_PER_SYNC, ours_settings.RECOGNITION_ENCODE_AHEAD = old
    assert toks == ref_tokens
    for i in range(n_lines):
        assert len(scores[i]) == len(ref_scores[i])
        L_ = min(len(toks[i]), max_tokens)
        assert np.array_equal(boxes[i, :L_].numpy(), ref_boxes[i, :L_].numpy()), i


def test_live_process_outputs_against_reference():
    """the oracle's process_outputs (what greedy_head_kernel is compared with on the GPU) against the reference's
    own RecognitionPredictor.process_outputs (recognition/__init__.py:294-325) on random logits incl. forced eos / pad winners:
    preds, next input ids, scores (0 where done), bbox ints."""
    from types import SimpleNamespace
    from oracle import rec_oracle as ro
    sr = ref_shim.import_recognition()
    g = torch.Generator().manual_seed(5)
    B, V, eos, pad, bbox_size = 37, 997, 3, 4, 1025
    lm = torch.randn(B, 1, V, generator=g) * 3
    lm[5, 0, eos] = 50.0
    lm[9, 0, pad] = 50.0
    bb = torch.rand(B, 1, 6, generator=g)
    stub = SimpleNamespace(processor=SimpleNamespace(eos_token_id=eos, pad_token_id=pad), device_pad_token=torch.tensor(pad),
                           model=SimpleNamespace(config=SimpleNamespace(bbox_size=bbox_size)))
    ref = sr.RecognitionPredictor.process_outputs(stub, {"lm_logits": lm, "bbox_logits": bb})
    got = ro.process_outputs(lm, bb, eos, pad, bbox_size)
    names = ("input_ids", "preds", "bbox_preds", "done", "scores")
    got = got if isinstance(got, dict) else dict(zip(names, got)) if isinstance(got, (tuple, list)) else {n: getattr(got, n) for n in names}
    for n in names:
        r = getattr(ref, n)
        assert got[n].shape == r.shape and torch.equal(got[n], r), n
