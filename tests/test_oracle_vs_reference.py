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


def test_live_tokenizer_against_reference_inner_ocr_tokenizer():
    """SURVEY 8(a) R5 (prompt ids) / R16 (decode): our OCRTokenizer against the reference's own InnerOCRTokenizer
    (common/surya/processor/tokenizer.py:26-221) built on the SAME stand-in math tokenizer and tag table: token ids of random
    strings with formatting / math / system tags, html entities, astral characters and math spans (terminated ones: the reference
    never returns on an unterminated <math>, see tokenizer.py note), and decode of random id streams from all three ranges."""
    import random
    rt = _ref("surya.common.surya.processor.tokenizer")
    from surya_amd.recognition.tokenizer import ByteMathTokenizer, OCRTokenizer, DEFAULT_SPECIAL_TOKENS
    math_tok = ByteMathTokenizer(300)
    ours = OCRTokenizer(DEFAULT_SPECIAL_TOKENS, math_tok)
    ref = rt.InnerOCRTokenizer(special_tokens=DEFAULT_SPECIAL_TOKENS, qwen_tokenizer=math_tok)
    assert ours.qwen_offset == ref.qwen_token_offset and ours.special_token_offset == ref.qwen_token_offset + ref.SPECIAL_TOKEN_OFFSET
    assert ours.SPECIAL_TOKEN_MAPPING == ref.SPECIAL_TOKEN_MAPPING and ours.vocab_size == ref.vocab_size + ref.qwen_token_offset
    rng = random.Random(11)
    tags = [t for k in ("formatting", "math_external", "system") for t in DEFAULT_SPECIAL_TOKENS.get(k, [])]
    pieces = list("abc xyz0189.,;-ÄøЖ漢字😀🧪") + ["&lt;", "&amp;", "&gt;", "<", ">", "x^2", "\\frac{a}{b}", "<br>", "</"]
    for _ in range(1500):
        parts = []
        for _ in range(rng.randint(0, 12)):
            r = rng.random()
            if r < 0.25:
                parts.append(rng.choice(tags))
            elif r < 0.35:
                parts.append("<math>" + "".join(rng.choice(pieces) for _ in range(rng.randint(0, 5))) + "</math>")
            else:
                parts.append(rng.choice(pieces))
        text = "".join(parts)
        if text.count("<math") > text.count("</math>"):
            text += "</math>"                               # keep spans terminated (the reference loops forever otherwise)
        assert ours._tokenize_ocr(text) == ref._tokenize(text), text
    specials = sorted(ref.REVERSE_SPECIAL_TOKEN_MAPPING)
    for _ in range(1500):
        ids = []
        for _ in range(rng.randint(0, 30)):
            r = rng.random()
            if r < 0.5:
                ids.append(ours.special_token_offset + rng.choice([rng.randint(0x20, 0x7e), rng.randint(0xa0, 0xd7ff), rng.randint(0xd800, 0xdfff)]))
            elif r < 0.75:
                ids.append(rng.choice(specials))
            else:
                ids.append(rng.randint(0, 255))
        assert ours._decode_ocr(ids) == ref.decode(ids), ids


def test_live_processor_call_against_reference():
    """the reference's own SuryaOCRProcessor.__call__ (common/surya/processor/__init__.py:330-420, with
    _process_ocr_with_boxes / _process_text_input / _process_image_input / _process_and_tile) and RecognitionPredictor.prepare_input
    (:259-292) against ours on crops that need no cv2 resize (sides already multiples of 28, area inside the task's bounds): image
    tiles bit for bit, grids, and per sequence the prompt ids (the reference's left-padded rows with the pad stripped), for all
    three tasks, with and without input text, math mode on and off. The reference's position ids are what packed prompts imply."""
    import numpy as np
    from types import SimpleNamespace
    sr = ref_shim.import_recognition()
    rp = _ref("surya.common.surya.processor")
    from surya_amd.recognition.processor import SuryaOCRProcessor
    from surya_amd.recognition.tokenizer import ByteMathTokenizer, OCRTokenizer
    from surya_amd.recognition.predictor import RecognitionPredictor
    tok = OCRTokenizer(None, ByteMathTokenizer(256), reserve_special=64)
    ours = SuryaOCRProcessor(tok)
    ref = rp.SuryaOCRProcessor(ocr_tokenizer=tok, blank_bbox_token_id=1025, num_register_tokens=4, patch_size=14, merge_size=2,
                               model_device="cpu")
    for name in ("image_token_id", "pad_token_id", "eos_token_id", "eoi_token_id", "no_output_token", "image_rotated_token", "nomath_token"):
        assert getattr(ours, name) == getattr(ref, name), name
    assert ours.bos_token_id == ref.bos_token_id and ours.register_token_ids == ref.register_token_ids
    rng = np.random.default_rng(3)
    shapes = [(56, 504), (168, 168), (84, 532), (196, 336), (392, 392), (28 * 7, 28 * 9)]
    tasks = ["ocr_with_boxes", "ocr_without_boxes", "block_without_boxes", "ocr_with_boxes", "ocr_with_boxes", "block_without_boxes"]
    texts = [None, "ab <b>c</b>", "", "x <math>a^2</math> y", "  padded  ", None]
    maths = [True, False, True, True, False, False]
    images = [rng.integers(0, 256, size=(h, w, 3)).astype(np.float32) for h, w in shapes]
    ours_pred = object.__new__(RecognitionPredictor)
    ours_pred.processor = ours
    ref_self = SimpleNamespace(processor=ref, tasks=sr.RecognitionPredictor.tasks)
    ref_batch = sr.RecognitionPredictor.prepare_input(ref_self, tasks, images, texts, maths)
    our_batch = ours_pred.prepare_input(tasks, images, texts, maths)
    for rb, ob in zip(ref_batch, our_batch):
        assert rb["task"] == ob["task"] and rb["inputs"][1] == ob["inputs"][1]
        assert np.array_equal(rb["inputs"][0]["image"], ob["inputs"][0]["image"])
    r = ref(ref_batch, padding_side="left")
    o = ours(our_batch)
    assert np.array_equal(r["image_tiles"].numpy(), o["image_tiles"])
    assert np.array_equal(r["grid_thw"].numpy()[:, 1:], o["grid_hw"]) and (r["grid_thw"].numpy()[:, 0] == 1).all()
    for i, seq in enumerate(o["input_ids"]):
        row, am, pos = r["input_ids"][i], r["attention_mask"][i], r["position_ids"][i]
        assert row[am].tolist() == list(seq), i
        assert pos[am].tolist() == list(range(len(seq))), i           # packed prompts: positions 0..L-1 of the real tokens


@pytest.mark.parametrize("sort_lines,return_words", [(False, False), (True, True)])
def test_live_recognition_call_against_reference_call(sort_lines, return_words):
    """SURVEY 8(a) R1 + R2 (bbox slicing) + R16 end to end on the host: the reference's own RecognitionPredictor.__call__
    (recognition/__init__.py:773-942) -- slice_bboxes, widest-first sort, restore order, polygons, get_bboxes_text, per-page
    regrouping, sort_text_lines, OCRResult -- runs with its prediction_loop replaced by a scripted one (tokens / scores / boxes
    are a function of each crop's pixels, so they do not depend on the order lines are processed in); ours runs with the same
    scripted device loop behind prepare_lines / generate. The OCRResults must be identical, field for field."""
    import numpy as np
    from types import SimpleNamespace
    from PIL import Image
    sr = ref_shim.import_recognition()
    R = sr.RecognitionPredictor
    from surya_amd.recognition.predictor import RecognitionPredictor
    from surya_amd.recognition.processor import SuryaOCRProcessor
    from surya_amd.recognition.tokenizer import ByteMathTokenizer, OCRTokenizer
    tok = OCRTokenizer(None, ByteMathTokenizer(256), reserve_special=64)
    proc = SuryaOCRProcessor(tok)
    sysm = tok.system_tokens
    specials = [v for k, v in tok.SPECIAL_TOKEN_MAPPING.items() if k not in sysm] + [sysm["<NO-MATH>"]]
    T_MAX, bbox_size = 40, 1025

    def scripted(crop):
        rng = np.random.default_rng(int(crop.sum()) % (2 ** 31) + 7 * crop.shape[1] + crop.shape[0])
        T = int(rng.integers(1, T_MAX))
        toks = []
        while len(toks) < T:
            r = rng.random()
            if r < 0.6:
                for ch in rng.choice(list("abc xyzÄ漢😀"), size=int(rng.integers(1, 6))):
                    raw = ch.encode("utf-16le")
                    toks += [raw[i] + (raw[i + 1] << 8) + tok.special_token_offset for i in range(0, len(raw), 2)]
            elif r < 0.8:
                toks.append(int(rng.choice(specials)))
            elif r < 0.95:
                toks += [int(x) for x in rng.integers(32, 127, size=int(rng.integers(1, 5)))]
            else:
                toks.append(int(proc.eos_token_id))
        toks = toks[:T]
        rows = np.sort(rng.integers(0, bbox_size, size=(T, 6)), axis=0).astype(np.float32)
        return toks, rng.random(T).astype(np.float32).tolist(), rows

    class RefCall:
        __call__ = R.__call__
        slice_bboxes = R.slice_bboxes
        get_bboxes_text = R.get_bboxes_text
        tasks = R.tasks

        def __init__(self):
            self.processor = proc
            self.model = SimpleNamespace(config=SimpleNamespace(bbox_size=bbox_size))

        def prediction_loop(self, flat, recognition_batch_size=None, math_mode=True):
            n = len(flat["slices"])
            boxes = torch.zeros(n, T_MAX, 6)
            toks, scores = [], []
            for k, crop in enumerate(flat["slices"]):
                t, s, rows = scripted(crop)
                toks.append(t); scores.append(s); boxes[k, : len(t)] = torch.from_numpy(rows)
            return toks, boxes, scores

    ours = object.__new__(RecognitionPredictor)
    ours.processor = proc
    ours.model = SimpleNamespace(cfg=SimpleNamespace(bbox_size=bbox_size))
    ours.device_preprocess = False
    ours.shard_lines = False
    ours.prepare_lines = lambda flat, math_mode=True: flat

    def generate(prep, recognition_batch_size=None, on_done=None, on_flush=None):
        n = len(prep["slices"])
        boxes = np.zeros((n, T_MAX, 6), np.float32)
        toks, scores = [], []
        for k, crop in enumerate(prep["slices"]):
            t, s, rows = scripted(crop)
            toks.append(t); scores.append(s); boxes[k, : len(t)] = rows
            on_done(k, t, s, boxes[k, : len(t)])
            if k % 5 == 4:
                on_flush()
        return toks, torch.from_numpy(boxes), scores
    ours.generate = generate

    rng = np.random.default_rng(1)
    pages = [Image.fromarray(rng.integers(0, 256, size=(300, 420, 3), dtype=np.uint8)) for _ in range(3)]
    bboxes = [[[10, 10, 200, 40], [15, 60, 400, 95], [5, 120, 90, 150], [0, 0, 0, 0]],
              [],
              [[30, 200, 410, 260], [30, 20, 100, 50], [120, 20, 300, 52], [500, 400, 900, 800]]]
    kw = dict(bboxes=bboxes, sort_lines=sort_lines, return_words=return_words)
    ref_out = RefCall()(pages, **kw)
    our_out = ours(pages, **kw)
    assert len(ref_out) == len(our_out) == 3
    for a, b in zip(our_out, ref_out):
        assert a.model_dump() == b.model_dump()


def test_live_detection_batching_against_reference_batch_detection():
    """the reference's own DetectionPredictor.batch_detection (detection/__init__.py:64-155: greedy packing by
    strip count, convert, split_image, the double LANCZOS prepare_image, stacking, stitching the strips' maps back per page with the
    last strip's padding cut) runs unmodified with a stand-in model whose "logits" are a fixed function of the pixel values; our
    reference-layout path (DetectionPredictor.batch_detection, the checker of the device path) runs with the same stand-in. Pages of
    mixed sizes incl. tall ones that are cut into strips, several batch sizes: identical per-page maps and sizes, batch by batch."""
    import numpy as np
    from types import SimpleNamespace
    from PIL import Image
    ref_shim.install()
    import surya.detection as sd
    from surya.detection.processor import SegformerImageProcessor as RefProc
    from surya_amd.detection.predictor import DetectionPredictor, SegformerImageProcessor as OurProc
    size = 256

    def fake_logits(x):                                     # [B, 3, H, W] normalised pixels -> [B, 2, H, W]
        x = x.float()
        return torch.stack([x.mean(1) * 0.5 + x[:, 0] * 0.25, x[:, 2] - x[:, 1]], 1)

    class RefModel:
        dtype, device = torch.float32, torch.device("cpu")
        config = SimpleNamespace(num_labels=2)
        def __call__(self, pixel_values):
            return SimpleNamespace(logits=fake_logits(pixel_values))

    class RefStub:
        batch_detection = sd.DetectionPredictor.batch_detection
        prepare_image = sd.DetectionPredictor.prepare_image
        disable_tqdm = True
        def __init__(self):
            self.model = RefModel()
            self.processor = RefProc(size={"height": size, "width": size})
        def get_batch_size(self):
            return 4

    ours = object.__new__(DetectionPredictor)
    ours.processor = OurProc({"height": size, "width": size})
    ours.model = SimpleNamespace(max_batch=64, cfg=SimpleNamespace(num_labels=2), device="cpu", forward=lambda chunk: fake_logits(chunk))
    rng = np.random.default_rng(2)
    from surya_amd.settings import settings as our_settings
    from surya.settings import settings as ref_settings
    old = (our_settings.DETECTOR_IMAGE_CHUNK_HEIGHT, ref_settings.DETECTOR_IMAGE_CHUNK_HEIGHT)
    our_settings.DETECTOR_IMAGE_CHUNK_HEIGHT = ref_settings.DETECTOR_IMAGE_CHUNK_HEIGHT = 300
    try:
        shapes = [(256, 256), (200, 310), (640, 250), (90, 700), (301, 256), (1000, 120), (256, 256)]          # (h, w)
        pages = [Image.fromarray(rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)) for h, w in shapes]
        pages[3] = pages[3].convert("L")                                                                   # a non-RGB page
        for bs in (1, 3, 4, 16):
            ref_gen = list(RefStub().batch_detection([p.copy() for p in pages], batch_size=bs))
            our_gen = list(ours.batch_detection([p.copy() for p in pages], batch_size=bs))
            assert len(ref_gen) == len(our_gen)
            for (rp, rs), (op, os_) in zip(ref_gen, our_gen):
                assert list(rs) == list(os_) and len(rp) == len(op)
                for a, b in zip(rp, op):
                    assert len(a) == len(b) == 2
                    for k in range(2):
                        assert a[k].shape == b[k].shape and np.array_equal(a[k], b[k])
    finally:
        our_settings.DETECTOR_IMAGE_CHUNK_HEIGHT, ref_settings.DETECTOR_IMAGE_CHUNK_HEIGHT = old


def test_live_heatmap_to_boxes_glue_against_reference_with_stand_in_cv2():
    """SURVEY 8(a) D8, the part that CAN be pinned without OpenCV: the reference's own detect_boxes / get_detected_boxes /
    get_and_clean_boxes / clean_boxes / parallel_get_boxes (detection/heatmap.py:14-184, common/util.py:9-36) run unmodified with a
    stand-in `cv2` whose five primitives are OUR restatements (4-connected labelling + statistics, rectangular dilation with the
    centre anchor, minimum-area rectangle, clockwise boxPoints) -- so everything around those primitives is the reference's code:
    dynamic thresholds, the area-10 filter, niter / buffer windows, the per-component maximum, the near-square rule, the start-corner
    roll, confidence normalisation, rescale to the page, fit_to_bounds, the containment filter, the y-expansion. Against our
    parallel_get_boxes on the same maps: identical polygons, confidences and page boxes. (The primitives themselves stay pinned to
    hand-derived vectors, tests/test_cv2_conventions.py: OpenCV is not in this image.)"""
    import sys
    import numpy as np
    from scipy import ndimage
    ref_shim.install()
    import cv2 as cv2_stub
    from surya_amd.detection import heatmap as ours

    def cc_with_stats(img, connectivity=4):
        assert connectivity == 4
        labels, count = ndimage.label(img > 0, structure=[[0, 1, 0], [1, 1, 1], [0, 1, 0]])
        stats = np.zeros((count + 1, 5), np.int32)
        for k, sl in enumerate(ndimage.find_objects(labels), start=1):
            stats[k] = [sl[1].start, sl[0].start, sl[1].stop - sl[1].start, sl[0].stop - sl[0].start, int((labels[sl] == k).sum())]
        return count + 1, labels.astype(np.int32), stats, None

    def box_points(rect):
        box = rect[1]
        c = box.mean(0)
        return box[np.argsort(np.arctan2(box[:, 1] - c[1], box[:, 0] - c[0]))].astype(np.float32)      # clockwise on screen

    added = dict(CC_STAT_LEFT=0, CC_STAT_TOP=1, CC_STAT_WIDTH=2, CC_STAT_HEIGHT=3, CC_STAT_AREA=4, MORPH_RECT=0,
                 connectedComponentsWithStats=cc_with_stats,
                 getStructuringElement=lambda shape, ksize: np.ones((ksize[1], ksize[0]), np.uint8),
                 dilate=lambda src, kernel: ours.dilate_rect(src, kernel.shape[0]),
                 minAreaRect=lambda pts: ("rect", ours.min_area_rect_points(np.asarray(pts))),
                 boxPoints=box_points)
    for k, v in added.items():
        setattr(cv2_stub, k, v)
    try:
        ref_shim.purge_bare_namespaces()
        import surya.detection.heatmap as rh
        rng = np.random.default_rng(4)
        n_boxes = 0
        for trial in range(6):
            H, W = (96, 160) if trial % 2 else (128, 128)
            m = ndimage.gaussian_filter(rng.random((H, W)), sigma=[1.5, 4.0][trial % 2] if trial < 4 else 1.0)
            m = (m - m.min()) / (m.max() - m.min())
            m = np.clip((m - 0.45) * 3.0, 0, 1).astype(np.float32)
            if trial == 5:
                m[:] = 0.0                                                  # blank page
                m[10:14, 10:14] = 0.9                                       # + a 16-pixel square component
            aff = rng.random((H, W)).astype(np.float32)
            orig = (W * 3 + 7, H * 2 + 5) if trial % 3 else (W, H)
            ref_res = rh.parallel_get_boxes([m.copy(), aff], orig)
            our_res = ours.parallel_get_boxes([m.copy(), aff], orig)
            assert [b.polygon for b in our_res.bboxes] == [b.polygon for b in ref_res.bboxes], trial
            assert [b.confidence for b in our_res.bboxes] == [float(b.confidence) for b in ref_res.bboxes], trial
            assert our_res.image_bbox == ref_res.image_bbox
            n_boxes += len(ref_res.bboxes)
        assert n_boxes > 20
    finally:
        for k in added:
            delattr(cv2_stub, k)


def test_live_crop_and_resize_glue_against_reference_with_stand_in_cv2():
    """SURVEY 8(a) R2 / R4 / R5, the part that can be pinned without OpenCV: the reference's own slice_polys_from_image /
    slice_and_pad_poly (input/processing.py:57-101) and prepare_input -> SuryaOCRProcessor.__call__ (scale_to_fit, the round-up to
    multiples of 28, normalise, patchify) run unmodified with a stand-in `cv2` whose fillPoly / resize are OUR restatements of
    those primitives -- so the crop box, pad value, validity rules, the area clamp's floor / ceil arithmetic and the tiling
    are the reference's code. Against ours: identical crops; identical tiles, grids and prompt ids for crops far below the minimum
    area, far above the task's maximum, and with sides that are no multiple of 28."""
    import numpy as np
    from types import SimpleNamespace
    ref_shim.install()
    import cv2 as cv2_stub
    from surya_amd.common import imageops
    from surya_amd.recognition import predictor as op

    def fill_poly(mask, polys, value):
        for pts in polys:
            mask[imageops.fill_poly_mask(mask.shape[0], mask.shape[1], np.asarray(pts)) > 0] = value
        return mask

    added = dict(INTER_LANCZOS4=4, INTER_CUBIC=2, ROTATE_90_COUNTERCLOCKWISE=2, fillPoly=fill_poly,
                 resize=lambda img, size, interpolation=None: imageops.resize(img, size[0], size[1], "lanczos4" if interpolation == 4 else "cubic"))
    for k, v in added.items():
        setattr(cv2_stub, k, v)
    try:
        ref_shim.purge_bare_namespaces()
        sr = ref_shim.import_recognition()
        import surya.input.processing as rip
        import surya.common.surya.processor as rp
        from surya_amd.recognition.processor import SuryaOCRProcessor
        from surya_amd.recognition.tokenizer import ByteMathTokenizer, OCRTokenizer
        rng = np.random.default_rng(9)
        page = rng.integers(0, 256, size=(200, 300, 3)).astype(np.float32)
        polys = [[[10, 20], [120, 22], [118, 60], [12, 58]], [[50, 100], [200, 90], [120, 150]], [[5, 5], [40, 5], [40, 30], [5, 30]],
                 [[100, 100], [100, 100], [100, 100], [100, 100]], [[250, 150], [299, 160], [290, 199], [240, 190]], [[30, 70], [90, 70]]]
        for a, b in zip(op.slice_polys_from_image(page, polys), rip.slice_polys_from_image(page, polys)):
            assert a.shape == b.shape and np.array_equal(a, b)

        tok = OCRTokenizer(None, ByteMathTokenizer(256), reserve_special=64)
        ours = SuryaOCRProcessor(tok)
        ref = rp.SuryaOCRProcessor(ocr_tokenizer=tok, blank_bbox_token_id=1025, num_register_tokens=4, patch_size=14, merge_size=2,
                                   model_device="cpu")

        # detect_and_slice_bboxes (:138-198): detector polygons -> crops, with and without a high-resolution copy of the page
        from PIL import Image
        from surya.common.polygon import PolygonBox as RefBox
        from surya_amd.common.geometry import PolygonBox as OurBox
        pages = [Image.fromarray(rng.integers(0, 256, size=(200, 300, 3), dtype=np.uint8)) for _ in range(3)]
        highres = [None, pages[1].resize((750, 450)), None]
        det_polys = [[[[10, 20], [120, 22], [118, 60], [12, 58]], [[50, 100], [200, 90], [210, 140], [60, 150]]], [],
                     [[[5, 5], [40, 5], [40, 30], [5, 30]]]]
        det_polys[1] = [[[30, 40], [250, 44], [248, 90], [28, 86]]]

        def fake_det(box_cls):
            def det(images, batch_size=None):
                out = []
                for polys in det_polys:
                    boxes = [box_cls(polygon=p) for p in polys]
                    for bx in boxes:
                        bx.rescale((1, 1), (1, 1))            # as in get_and_clean_boxes: leaves int corners (heatmap.py:131-133)
                    out.append(SimpleNamespace(bboxes=boxes))
                return out
            return det
        ref_flat = sr.RecognitionPredictor.detect_and_slice_bboxes(SimpleNamespace(processor=ref), pages, ["ocr_with_boxes"] * 3,
                                                                   fake_det(RefBox), highres_images=highres)
        our_pred0 = object.__new__(op.RecognitionPredictor)
        our_pred0.processor, our_pred0.device_preprocess = ours, False
        our_flat = our_pred0.detect_and_slice_bboxes(pages, ["ocr_with_boxes"] * 3, fake_det(OurBox), highres_images=highres)
        assert set(ref_flat) == set(our_flat)
        for key in ("slice_map", "polygons", "task_names", "input_text"):
            assert ref_flat[key] == our_flat[key], key
        assert [tuple(x) for x in ref_flat["res_scales"]] == [tuple(x) for x in our_flat["res_scales"]]
        for x, y in zip(ref_flat["slices"], our_flat["slices"]):
            assert x.shape == y.shape and np.array_equal(x, y)

        shapes = [(9, 31), (40, 333), (64, 513), (300, 2000), (1200, 90), (57, 57), (168, 169), (27, 1100)]
        tasks = ["ocr_with_boxes", "ocr_without_boxes", "block_without_boxes", "ocr_with_boxes", "block_without_boxes", "ocr_with_boxes",
                 "ocr_with_boxes", "ocr_without_boxes"]
        images = [rng.integers(0, 256, size=(h, w, 3)).astype(np.float32) for h, w in shapes]
        texts, maths = [None] * len(shapes), [True] * len(shapes)
        our_pred = object.__new__(op.RecognitionPredictor)
        our_pred.processor = ours
        ref_self = SimpleNamespace(processor=ref, tasks=sr.RecognitionPredictor.tasks)
        rb = sr.RecognitionPredictor.prepare_input(ref_self, tasks, images, texts, maths)
        ob = our_pred.prepare_input(tasks, images, texts, maths)
        for x, y in zip(rb, ob):
            assert x["inputs"][0]["image"].shape == y["inputs"][0]["image"].shape
            assert np.array_equal(x["inputs"][0]["image"], y["inputs"][0]["image"])
        r, o = ref(rb, padding_side="left"), ours(ob)
        assert np.array_equal(r["grid_thw"].numpy()[:, 1:], o["grid_hw"])
        assert np.array_equal(r["image_tiles"].numpy(), o["image_tiles"])
        for i, seq in enumerate(o["input_ids"]):
            assert r["input_ids"][i][r["attention_mask"][i]].tolist() == list(seq), i
    finally:
        for k in added:
            delattr(cv2_stub, k)


def test_live_device_path_geometry_against_reference_shapes():
    """The host integers of the DEVICE pre-processing path (recognition/preprocess_gpu.py: LineRef rectangles and fit_sizes, which
    size the kernels' work) against the shapes the reference's own functions produce: slice_bboxes_from_image / slice_and_pad_poly
    crop shapes for random boxes and quadrilaterals (incl. out-of-page and degenerate ones), and scale_to_fit -> _process_and_tile
    sizes for random crop sizes under every task's bounds (cv2.resize stood in by an allocator of the requested size: only the
    size arithmetic is under test)."""
    import numpy as np
    ref_shim.install()
    import cv2 as cv2_stub
    from surya_amd.recognition.preprocess_gpu import bbox_ref, poly_ref, fit_sizes
    added = dict(INTER_LANCZOS4=4, INTER_CUBIC=2, fillPoly=lambda mask, polys, value: mask,
                 resize=lambda img, size, interpolation=None: np.zeros((size[1], size[0], 3), np.float32))
    for k, v in added.items():
        setattr(cv2_stub, k, v)
    try:
        ref_shim.purge_bare_namespaces()
        sr = ref_shim.import_recognition()
        import surya.input.processing as rip
        import surya.common.surya.processor as rp
        rng = np.random.default_rng(21)
        H, W = 180, 260
        page = np.zeros((H, W, 3), np.float32)
        for _ in range(400):
            b = [int(v) for v in rng.integers(-20, 300, size=4)]
            assert bbox_ref(0, W, H, b).shape == rip.slice_bboxes_from_image(page, [b])[0].shape, b
        for _ in range(400):
            pts = [[int(rng.integers(0, W + 30)), int(rng.integers(0, H + 30))] for _ in range(4)]
            assert poly_ref(0, W, H, pts).shape == rip.slice_and_pad_poly(page, pts).shape, pts
        proc = object.__new__(rp.SuryaOCRProcessor)
        proc.patch_size, proc.merge_size = 14, 2
        proc.rescale_factor = rp.SuryaOCRProcessor.rescale_factor
        proc.image_mean = np.array(rp.SuryaOCRProcessor.image_mean, np.float32)
        proc.image_std = np.array(rp.SuryaOCRProcessor.image_std, np.float32)
        for task, spec in sr.RecognitionPredictor.tasks.items():
            for _ in range(150):
                h, w = int(rng.integers(1, 1500)), int(rng.integers(1, 2500))
                mid = rp.SuryaOCRProcessor.scale_to_fit(np.zeros((h, w, 3), np.float32), spec["img_size"])
                _, (t, gh, gw) = proc._process_and_tile(mid)
                (mh, mw), (oh, ow) = fit_sizes(h, w, spec["img_size"])
                assert (mh, mw) == mid.shape[:2] and (oh, ow) == (gh * 14, gw * 14) and t == 1, (task, h, w)
    finally:
        for k in added:
            delattr(cv2_stub, k)


def test_live_checkpoint_directory_written_by_the_reference(tmp_path):
    """A checkpoint directory written by the REFERENCE's own SuryaModel.save_pretrained (+ the tiny Qwen2 BPE files) loads through
    RecognitionModelLoader with the same configuration, tensors and token-id layout the reference's own loader / tokenizer derive
    from it (surya/recognition/loader.py:25-82, processor/tokenizer.py:224-260); and tests/ckpt_util.write_rec_checkpoint -- the
    writer the travelling tests use on the GPU box -- produces a config.json that agrees with the reference's on every key it writes."""
    import json, os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import ckpt_util as cu
    from oracle.make_golden import build_reference_rec
    from surya_amd.recognition.predictor import RecognitionModelLoader
    from surya_amd.synth import make_rec_weights
    ref_shim.install()
    cfg = cu.tiny_checkpoint_config()
    sd = make_rec_weights(cfg, 3)
    ref = build_reference_rec(cfg, sd, "sdpa")
    special = cu.special_ocr_tokens()
    ref.config.special_ocr_tokens = special
    ref.config.tie_word_embeddings = False          # the synthetic weight sets carry an untied lm_head (surya_amd/synth.py)
    path = str(tmp_path / "from_reference")
    for m in ref.modules():                         # transformers 5.x wants a dict here; the reference (4.x era) declares a list
        if isinstance(getattr(m, "_tied_weights_keys", None), (list, tuple)):
            m._tied_weights_keys = {}
    ref.save_pretrained(path, safe_serialization=True)
    cu.write_tokenizer_files(path)
    # -- our loader on the reference-written directory
    ld = RecognitionModelLoader(path)
    ld._resolve()
    assert ld._cfg.encoder == cfg.encoder and ld._cfg.decoder == cfg.decoder
    assert (ld._cfg.bbox_size, ld._cfg.num_register_tokens) == (ref.config.bbox_size, ref.config.num_register_tokens)
    rsd = ref.state_dict()
    assert set(ld._sd) <= set(rsd) and all(torch.equal(ld._sd[k], rsd[k]) for k in ld._sd)
    assert {k for k in rsd if k not in ld._sd} <= {"lm_head.weight"}            # only a tied head may be left out of the file
    # -- token-id layout vs the reference's own tokenizer on the same directory
    rt = _ref("surya.common.surya.processor.tokenizer")
    rtok = rt.SuryaOCRTokenizer(special_tokens=special, model_checkpoint=path)
    tok = ld.tokenizer()
    assert (tok.qwen_offset, tok.special_token_offset) == (rtok.qwen_offset, rtok.special_token_offset)
    assert tok.system_tokens == rtok.system_tokens and tok.SPECIAL_TOKEN_MAPPING == dict(rtok.SPECIAL_TOKEN_MAPPING)
    for text in ['plain text', 'a<b>x</b><math display="inline">\\frac{1}{2} x^2</math>é\U0001d11e', '<i>it</i> &amp; <br>']:
        ours = tok([text], ["ocr_with_boxes"])["input_ids"][0]
        theirs = rtok([text], ["ocr_with_boxes"])["input_ids"][0]
        assert list(ours) == list(theirs), text
        assert tok.decode(ours) == rtok.decode(theirs, task="ocr_with_boxes")
    # -- our writer vs the reference's config.json
    ours_dir = cu.write_rec_checkpoint(str(tmp_path / "ours"), cfg, sd, special)
    a, b = json.load(open(os.path.join(ours_dir, "config.json"))), json.load(open(os.path.join(path, "config.json")))
    for k, v in a.items():
        if isinstance(v, dict) and k != "special_ocr_tokens":
            for kk, vv in v.items():
                assert b[k][kk] == vv, (k, kk, b[k].get(kk), vv)
        else:
            assert b[k] == v, (k, b.get(k), v)


def test_live_detector_checkpoint_directory_written_by_the_reference(tmp_path):
    """EfficientViTForSemanticSegmentation.save_pretrained + SegformerImageProcessor.save_pretrained (the reference's own writers,
    surya/detection/loader.py:23-63 reads them back) -> DetectionModelLoader: same configuration, tensors and processor settings; and
    tests/ckpt_util.write_det_checkpoint agrees with the reference's files on every key it writes."""
    import dataclasses, json, os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import ckpt_util as cu
    ref_shim.install()
    from surya.detection.model.config import EfficientViTConfig
    from surya.detection.model.encoderdecoder import EfficientViTForSemanticSegmentation
    from surya.detection.processor import SegformerImageProcessor
    from surya_amd.config import det_config
    from surya_amd.detection.predictor import DetectionModelLoader
    from surya_amd.synth import make_det_weights
    c = det_config("DET-TINY")
    sd = make_det_weights(c, 5)
    rc = EfficientViTConfig(widths=c.widths, depths=c.depths, head_dim=c.head_dim, decoder_layer_hidden_size=c.decoder_layer_hidden_size,
                            decoder_hidden_size=c.decoder_hidden_size, num_labels=c.num_labels)
    m = EfficientViTForSemanticSegmentation(rc).eval()
    m.load_state_dict(sd, strict=True)
    path = str(tmp_path / "det_from_reference")
    m.save_pretrained(path, safe_serialization=True)
    SegformerImageProcessor(size={"height": 256, "width": 256}).save_pretrained(path)
    ld = DetectionModelLoader(path)
    assert dataclasses.replace(ld._cfg, name=c.name) == c and ld._size == 256
    rsd = m.state_dict()
    assert set(ld._sd) == set(rsd) and all(torch.equal(ld._sd[k], rsd[k]) for k in rsd)
    p = ld.processor()
    import numpy as np
    rp = SegformerImageProcessor.from_pretrained(path)
    assert p.size == dict(rp.size) and np.allclose(p.image_mean, rp.image_mean) and np.allclose(p.image_std, rp.image_std)
    ours = cu.write_det_checkpoint(str(tmp_path / "det_ours"), c, sd, size=256)
    for fn in ("config.json", "preprocessor_config.json"):
        a, b = json.load(open(os.path.join(ours, fn))), json.load(open(os.path.join(path, fn)))
        for k, v in a.items():
            assert k in b and (b[k] == v or (isinstance(v, float) and abs(b[k] - v) < 1e-12)), (fn, k, b.get(k), v)


def test_live_layout_host_logic_against_reference():
    """SURVEY 8(f) rank 4, host side: the reference's own ImageSlicer (surya/layout/slicer.py), prediction_to_polygon
    (surya/layout/util.py) and SuryaEncoderImageProcessor (surya/common/donut/processor.py; cv2.resize stood in by our INTER_CUBIC
    restatement rounded to uint8) against surya_amd.layout's restatements: identical slices / positions / counts, polygons, merged
    LayoutResults and pixel_values."""
    import numpy as np
    from PIL import Image
    ref_shim.install_layout()
    import cv2 as cv2_stub
    from surya_amd.common import imageops
    from surya_amd.layout import predictor as lp, slicer as ls
    from surya_amd.layout.schema import LayoutBox as OB, LayoutResult as OR

    def resize(img, size, interpolation=None):
        assert interpolation == 2                      # PIL's BILINEAR constant, read by cv2 as INTER_CUBIC
        out = imageops.resize(img.astype(np.float32), size[0], size[1], "cubic")
        return np.clip(np.rint(out), 0, 255).astype(img.dtype)

    added = dict(INTER_LANCZOS4=4, INTER_CUBIC=2, resize=resize)
    for k, v in added.items():
        setattr(cv2_stub, k, v)
    try:
        rs = ref_shim.import_submodule("surya.layout.slicer")
        ru = ref_shim.import_submodule("surya.layout.util")
        rsch = ref_shim.import_submodule("surya.layout.schema")
        rproc = ref_shim.import_submodule("surya.common.donut.processor")
        rng = np.random.default_rng(4)
        mins, sizes = {"height": 1500, "width": 1500}, {"height": 1200, "width": 1200}
        a, b = ls.ImageSlicer(mins, sizes), rs.ImageSlicer(mins, sizes)
        imgs = [Image.fromarray(rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)) for h, w in ((800, 600), (1600, 900), (1000, 5200), (1501, 1501))]
        assert [a.slice_count(i) for i in imgs] == [b.slice_count(i) for i in imgs]
        (sa_, pa), (sb_, pb) = a.slice(imgs), b.slice(imgs)
        assert pa == pb and [s.size for s in sa_] == [s.size for s in sb_]
        assert all(np.array_equal(np.asarray(x), np.asarray(y)) for x, y in zip(sa_, sb_))
        # prediction_to_polygon on random tokens
        for _ in range(300):
            tok = torch.tensor(rng.integers(0, 1025, size=7).astype(np.float32))
            size = (int(rng.integers(50, 3000)), int(rng.integers(50, 3000)))
            assert lp.prediction_to_polygon(tok.numpy(), size, 1024, 512) == ru.prediction_to_polygon(tok, size, 1024, 512)
            # what the loop really passes: sigmoid * 1024 in the model dtype -- the reference's corner arithmetic is tensor arithmetic in
            # THAT dtype (fp32 on the CPU path, bf16 on its GPU path), only the final scaling is Python floats
            frac = torch.tensor((rng.random(7) * 1024).astype(np.float32))
            assert lp.prediction_to_polygon(frac.numpy(), size, 1024, 512) == ru.prediction_to_polygon(frac, size, 1024, 512)
            half = frac.to(torch.bfloat16)
            assert lp.prediction_to_polygon(half.float().numpy(), size, 1024, 512, dtype="bfloat16") == ru.prediction_to_polygon(half, size, 1024, 512)
        # merging per-slice results: same boxes on both sides
        labels = ["Text", "Picture", "Figure", "Table", "SectionHeader"]

        def results(BoxCls, ResCls):
            r = np.random.default_rng(7)
            out = []
            for size in ((1300, 900), (1300, 900), (1300, 400)):
                bx = []
                for z in range(6):
                    x0, y0 = r.uniform(0, size[0] - 200), r.uniform(-30, size[1] - 100)
                    w, h = r.uniform(40, 400), r.uniform(20, 200)
                    bx.append(BoxCls(polygon=[[x0, y0], [x0 + w, y0], [x0 + w, y0 + h], [x0, y0 + h]], label=labels[int(r.integers(0, 5))],
                                     position=z, top_k={"Text": 0.5}, confidence=0.5))
                out.append(ResCls(bboxes=bx, image_bbox=[0, 0, size[0], size[1]]))
            return out

        pos = [(0, 0, 0), (0, 0, 1), (0, 0, 2)]
        ja, jb = a.join(results(OB, OR), pos), b.join(results(rsch.LayoutBox, rsch.LayoutResult), pos)
        assert len(ja) == len(jb) == 1 and ja[0].image_bbox == jb[0].image_bbox and ja[0].sliced == jb[0].sliced
        assert [(x.polygon, x.label, x.position) for x in ja[0].bboxes] == [(y.polygon, y.label, y.position) for y in jb[0].bboxes]
        # image processor
        ours = lp.LayoutImageProcessor({"height": 768, "width": 768})
        theirs = rproc.SuryaEncoderImageProcessor(max_size={"height": 768, "width": 768})
        page = [imgs[0], imgs[1].crop((0, 0, 900, 768))]
        pa_, pb_ = ours(page)["pixel_values"], theirs(page)["pixel_values"]
        for x, y in zip(pa_, pb_):
            assert x.shape == (3, 768, 768) and np.array_equal(x, np.asarray(y))
    finally:
        for k in added:
            delattr(cv2_stub, k)


def _load_reference_table_predictor():
    import importlib.util
    import os
    mods = ref_shim.import_table_modules()
    spec = importlib.util.spec_from_file_location("ref_table_rec_pkg", os.path.join(ref_shim.REFERENCE_ROOT, "surya", "table_rec", "__init__.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return mods, m


def test_live_table_host_logic_against_reference():
    """Table recognition, host side, against the reference's own code imported from /root/reference:
      * LabelShaper (surya/table_rec/shaper.py) on random polygons / property dicts;
      * SuryaTableRecProcessor's prompts (table_rec/processor.py) for query items with and without columns;
      * TableRecPredictor.decode_batch_predictions (table_rec/__init__.py:236-387) on synthetic row / column / spanning-cell
        predictions (merges, colspans, headers);
      * the WHOLE predictor, both passes, with the reference's predictor driving the reference's DonutSwinModel + SuryaTableRecDecoder
        (TABLE-TINY synthetic weights, dynamic cache) and ours driving the CPU oracle behind HipLayoutModel's interface
        (tests/table_util.py): identical TableResults."""
    import copy
    import sys
    from types import SimpleNamespace
    import numpy as np
    from PIL import Image
    (cfgm, encm, decm, rshaper_mod), rpkg = _load_reference_table_predictor()
    rproc_mod = ref_shim.import_submodule("surya.table_rec.processor")
    from oracle.make_golden_table import build_reference_table
    from surya_amd.synth import make_table_weights
    from surya_amd.table_rec import predictor as tp
    from surya_amd.table_rec.config import table_config
    from surya_amd.table_rec.processor import TableRecProcessor
    from surya_amd.table_rec.shaper import LabelShaper
    from table_util import OracleTableModel
    rng = np.random.default_rng(12)
    ours, ref = LabelShaper(), rshaper_mod.LabelShaper()
    assert ours.component_idx_dict() == ref.component_idx_dict() and ours.property_keys == ref.property_keys
    for key in ours.property_keys:
        assert ours.get_box_property(key) == ref.get_box_property(key)
        assert ours.get_box_property(key, add_special_tokens=False) == ref.get_box_property(key, add_special_tokens=False)

    def items(n, with_bbox=False):
        out = []
        for _ in range(n):
            x0, y0 = rng.uniform(-50, 900), rng.uniform(-50, 900)
            w, h = rng.uniform(5, 500), rng.uniform(5, 300)
            sk = rng.uniform(-8, 8, size=2)
            poly = [[x0 - sk[0], y0 - sk[1]], [x0 + w - sk[0], y0 + sk[1]], [x0 + w + sk[0], y0 + h + sk[1]], [x0 + sk[0], y0 + h - sk[1]]]
            it = {"polygon": poly, "category": int(rng.integers(0, 5)), "colspan": int(rng.integers(0, 4)), "merges": int(rng.integers(0, 4)),
                  "is_header": int(rng.integers(0, 2))}
            out.append(it)
        return out

    a = items(40)
    b = copy.deepcopy(a)
    ca, cb = ours.convert_polygons_to_bboxes(a), ref.convert_polygons_to_bboxes(b)
    assert [x["bbox"] for x in ca] == [list(x["bbox"]) for x in cb]
    la, lb = ours.dict_to_labels(ca), ref.dict_to_labels(cb)
    assert la == lb and [x["bbox"] for x in ca] == [x["bbox"] for x in cb]
    for _ in range(200):
        box = rng.uniform(0, 1024, size=6).tolist()
        assert ours.convert_bbox_to_polygon(list(box)) == ref.convert_bbox_to_polygon(list(box))

    # processor prompts
    rp = object.__new__(rproc_mod.SuryaTableRecProcessor)
    rp.box_size, rp.special_token_count, rp.shaper = (1024, 1024), 5, ref
    rp.token_pad_id, rp.token_eos_id, rp.token_bos_id, rp.token_query_end_id = 0, 1, 1, 4
    op = TableRecProcessor({"height": 128, "width": 128})
    rp.image_processor = lambda images, *a_, **k_: {"pixel_values": op.image_processor(images)["pixel_values"]}
    imgs = [Image.fromarray(rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)) for h, w in ((300, 500), (128, 128), (700, 260))]
    q = [{"polygon": [[0, 0], [im.width, 0], [im.width, im.height], [0, im.height]], "category": 4, "colspan": 0, "merges": 0, "is_header": 0} for im in imgs]
    ra, rb = op(images=imgs, query_items=copy.deepcopy(q)), rp(images=imgs, query_items=copy.deepcopy(q))
    assert np.array_equal(ra["input_ids"], rb["input_ids"].numpy())
    assert all(np.array_equal(x, y) for x, y in zip(ra["pixel_values"], rb["pixel_values"]))
    rows_q, cols_q = items(5), items(3)
    ra = op(images=None, query_items=copy.deepcopy(rows_q), columns=copy.deepcopy(cols_q), convert_images=False)
    rb = rp(images=None, query_items=copy.deepcopy(rows_q), columns=copy.deepcopy(cols_q), convert_images=False)
    assert np.array_equal(ra["input_ids"], rb["input_ids"].numpy()) and ra["input_ids"].shape == (5, 6, 10)

    # assembly of a synthetic table: 4 rows x 3 columns, a colspan-2 cell, a vertical merge, a too-short spanning cell, a header row
    def synthetic():
        def bb(x0, y0, x1, y1):
            return [(x0 + x1) / 2, (y0 + y1) / 2, x1 - x0, y1 - y0, 512.0, 512.0]
        xs, ys = [40, 300, 620, 980], [30, 200, 420, 640, 900]
        rowcol = []
        for r in range(4):
            rowcol.append({"bbox": bb(xs[0], ys[r], xs[3], ys[r + 1]), "category": 1, "merges": 0, "colspan": 1, "is_header": int(r == 0)})
        for c in range(3):
            rowcol.append({"bbox": bb(xs[c], ys[0], xs[c + 1], ys[4]), "category": 2, "merges": 0, "colspan": 1, "is_header": int(c == 0)})
        rowcol.append({"bbox": bb(0, 0, 1024, 1024), "category": 4, "merges": 0, "colspan": 1, "is_header": 0})
        cells = [
            [{"bbox": bb(xs[0], ys[0], xs[2], ys[1]), "category": 3, "merges": 0, "colspan": 2, "is_header": 1}],
            [{"bbox": bb(xs[2], ys[1], xs[3], ys[2]), "category": 3, "merges": 2, "colspan": 1, "is_header": 0},
             {"bbox": bb(xs[0], ys[1], xs[1], ys[1] + 40), "category": 3, "merges": 1, "colspan": 1, "is_header": 0}],
            [{"bbox": bb(xs[2], ys[2], xs[3], ys[3]), "category": 3, "merges": 1, "colspan": 1, "is_header": 0},
             {"bbox": bb(xs[0], ys[2], xs[1], ys[3]), "category": 3, "merges": 0, "colspan": 1, "is_header": 0}],
            [{"bbox": bb(xs[1], ys[3], xs[3], ys[4]), "category": 3, "merges": 0, "colspan": 3, "is_header": 0}],
        ]
        return [rowcol], cells, [(1600, 1100)], [0, 0, 0, 0]

    o_self = SimpleNamespace(processor=op)
    r_self = SimpleNamespace(processor=rp)
    res_a = tp.TableRecPredictor.decode_batch_predictions(o_self, *synthetic(), ours)
    res_b = rpkg.TableRecPredictor.decode_batch_predictions(r_self, *synthetic(), ref)
    assert [r.model_dump() for r in res_a] == [r.model_dump() for r in res_b]
    assert len(res_a[0].rows) == 4 and len(res_a[0].cols) == 3 and any(c.colspan == 2 for c in res_a[0].cells)
    assert any(c.rowspan == 2 for c in res_a[0].cells)                       # the vertical merge happened

    # the whole predictor, both passes
    cfg = table_config("TABLE-TINY")
    sd = make_table_weights(cfg, 0)
    # random heads rarely say "Table-row" / "Table-column": tilt the category head so both passes have work
    sd["decoder.box_property_heads.category.weight"][5 + 1] *= 3.0
    sd["decoder.box_property_heads.category.weight"][5 + 2] *= 2.5
    enc, dec, dec_cfg = build_reference_table(cfg, sd)
    max_tokens = 14
    settings_mod = sys.modules["surya.settings"]
    old_max, old_tp = settings_mod.settings.TABLE_REC_MAX_BOXES, tp.TABLE_REC_MAX_BOXES
    settings_mod.settings.TABLE_REC_MAX_BOXES = max_tokens
    tp.TABLE_REC_MAX_BOXES = max_tokens
    try:
        # The fed-back box numbers are floats TRUNCATED to tokens: where the reference computes 607.0 and the oracle 606.99994 (one fp32
        # ulp apart) the two runs part ways, which says nothing about the host logic. Seed 3's pages have no such coin flip (seed 12's
        # have one).
        prng = np.random.default_rng(3)
        pages = [Image.fromarray(prng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)) for h, w in ((200, 320), (128, 128), (90, 400))]
        rpred = object.__new__(rpkg.TableRecPredictor)
        rp2 = copy.copy(rp)
        rpred.processor = rp2
        rpred.model = SimpleNamespace(encoder=enc, decoder=dec, device=torch.device("cpu"), dtype=torch.float32, config=dec_cfg)
        rpred.disable_tqdm = True
        rpred.get_batch_size = lambda: 2
        want = rpred.batch_table_recognition(pages, batch_size=2)
        opred = object.__new__(tp.TableRecPredictor)
        opred.model = OracleTableModel(cfg, sd, max_batch=8)
        opred.processor = op
        got = opred.batch_table_recognition(pages, batch_size=2)
    finally:
        settings_mod.settings.TABLE_REC_MAX_BOXES = old_max
        tp.TABLE_REC_MAX_BOXES = old_tp
    assert len(got) == len(want) == 3
    n_rows = sum(len(r.rows) for r in want)
    assert n_rows > 0 and sum(len(r.cols) for r in want) > 0, "the synthetic weights produced no rows / columns: the second pass was not exercised"
    for a_, b_ in zip(got, want):
        da, db = a_.model_dump(), b_.model_dump()
        assert da.keys() == db.keys()
        for key in ("rows", "cols", "cells", "unmerged_cells"):
            assert len(da[key]) == len(db[key]), key
            for x, y in zip(da[key], db[key]):
                px, py = np.array(x.pop("polygon"), np.float64), np.array(y.pop("polygon"), np.float64)
                assert np.allclose(px, py, atol=1e-3), (key, px, py)
                x.pop("bbox", None); y.pop("bbox", None)
                assert x == y, (key, x, y)
        assert da["image_bbox"] == db["image_bbox"]


def test_live_layout_and_table_checkpoint_directories_written_by_the_reference(tmp_path):
    """SuryaLayoutModel.save_pretrained / TableRecEncoderDecoderModel.save_pretrained (+ SuryaEncoderImageProcessor.save_pretrained for
    the table processor) -- the reference's own writers; surya/layout/loader.py:25-64 and surya/table_rec/loader.py:19-76 read them back
    -- load through LayoutModelLoader / TableRecModelLoader: same configuration (every field our dataclasses carry), same tensors, same
    image statistics."""
    import dataclasses
    import os
    from oracle.make_golden_layout import build_reference_layout
    from oracle.make_golden_table import build_reference_table
    from surya_amd.layout.config import layout_config
    from surya_amd.layout.predictor import LayoutModelLoader
    from surya_amd.synth import make_layout_weights, make_table_weights
    from surya_amd.table_rec.config import table_config
    from surya_amd.table_rec.predictor import TableRecModelLoader
    cfgm, encm, decm = ref_shim.import_layout_modules()
    edm = ref_shim.import_submodule("surya.layout.model.encoderdecoder")
    cfg = layout_config("LAYOUT-TINY")
    sd = make_layout_weights(cfg, 3)
    enc, dec, dec_cfg = build_reference_layout(cfg, sd)
    full = cfgm.SuryaLayoutConfig(encoder=enc.config, decoder=dec_cfg)
    full._attn_implementation = "eager"
    model = edm.SuryaLayoutModel(full, encoder=enc, decoder=dec)
    path = str(tmp_path / "layout_from_reference")
    model.save_pretrained(path, safe_serialization=True)
    ld = LayoutModelLoader(path)
    assert dataclasses.replace(ld._cfg, name=cfg.name) == cfg, (ld._cfg, cfg)
    rsd = model.state_dict()
    assert set(ld._sd) == set(rsd) and all(torch.equal(ld._sd[k], rsd[k]) for k in rsd)
    assert all(torch.equal(ld._sd[k], sd[k]) for k in sd)           # the names our synthetic / repack code uses ARE the checkpoint's

    tcfgm, tencm, tdecm, _ = ref_shim.import_table_modules()
    tedm = ref_shim.import_submodule("surya.table_rec.model.encoderdecoder")
    rproc = ref_shim.import_submodule("surya.common.donut.processor")
    tcfg = table_config("TABLE-TINY")
    tsd = make_table_weights(tcfg, 4)
    tenc, tdec, tdec_cfg = build_reference_table(tcfg, tsd)
    tfull = tcfgm.SuryaTableRecConfig(encoder=tenc.config, decoder=tdec_cfg)
    tfull._attn_implementation = "eager"
    tmodel = tedm.TableRecEncoderDecoderModel(tfull, encoder=tenc, decoder=tdec)
    tpath = str(tmp_path / "table_from_reference")
    tmodel.save_pretrained(tpath, safe_serialization=True)
    rproc.SuryaEncoderImageProcessor(max_size={"height": 128, "width": 128}, image_mean=[0.4, 0.5, 0.6], image_std=[0.2, 0.25, 0.3]).save_pretrained(tpath)
    tl = TableRecModelLoader(tpath)
    assert dataclasses.replace(tl._cfg, name=tcfg.name) == tcfg, (tl._cfg, tcfg)
    trsd = tmodel.state_dict()
    assert set(tl._sd) == set(trsd) and all(torch.equal(tl._sd[k], trsd[k]) for k in trsd)
    assert all(torch.equal(tl._sd[k], tsd[k]) for k in tsd)
    p = tl.processor()
    import numpy as np
    assert np.allclose(p.image_processor.image_mean, [0.4, 0.5, 0.6]) and np.allclose(p.image_processor.image_std, [0.2, 0.25, 0.3])
    assert p.image_processor.max_size == {"height": 128, "width": 128}
    assert os.path.exists(os.path.join(tpath, "preprocessor_config.json"))


@pytest.mark.parametrize("name,size", [("DET-TINY", 160), ("DET-DEFAULT", 64)])
def test_live_detector_op_list_incl_folded_head_against_the_reference_module(name, size):
    """The op list the HIP interpreter executes (surya_amd/detection/plan.py), run in plain PyTorch (tests/det_plan_interp.py), against
    the REAL EfficientViTForSemanticSegmentation on the same pixels -- for the decode head in its folded form (one 1x1 conv per stage
    with merged weights + sum / ReLU / classifier, the default) and in the reference's own op order. What is pinned here without a GPU:
    the BatchNorm folding, the NHWC weight layouts and the algebra of the fold; the GPU tests pin the kernels to the same numbers."""
    import torch.nn.functional as F
    ref_shim.install()
    from surya.detection.model.config import EfficientViTConfig
    from surya.detection.model.encoderdecoder import EfficientViTForSemanticSegmentation
    from oracle.det_oracle import normalise_pages
    from surya_amd.config import det_config
    from surya_amd.detection.plan import build_det_plan
    from surya_amd.synth import make_det_weights, make_pages
    from det_plan_interp import run_plan
    c = det_config(name)
    rc = EfficientViTConfig(widths=c.widths, depths=c.depths, head_dim=c.head_dim, decoder_layer_hidden_size=c.decoder_layer_hidden_size,
                            decoder_hidden_size=c.decoder_hidden_size, num_labels=c.num_labels)
    m = EfficientViTForSemanticSegmentation(rc).eval()
    sd = make_det_weights(c, 3)
    m.load_state_dict(sd, strict=True)
    x = normalise_pages(make_pages(2, size, seed=31))
    with torch.inference_mode():
        ref_low = m(pixel_values=x).logits                                                        # the module applies the sigmoid itself (:747)
        ref_up = F.interpolate(ref_low, size=(size, size), mode="bilinear", align_corners=False)   # :121-129
        for folded in (True, False):
            low, up = run_plan(build_det_plan(c, sd, size, size, folded_head=folded), x)
            assert (low - ref_low).abs().max().item() <= 2e-5, (folded, (low - ref_low).abs().max().item())
            assert (up - ref_up).abs().max().item() <= 2e-5
    assert ref_low.std().item() > 0.02
