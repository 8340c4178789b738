"""CPU: the array form of RecognitionPredictor's output assembly (get_bboxes_text + _chars_of) against a
per-token restatement of the reference's loop (surya/recognition/__init__.py:609-771 token runs, :905-909 per-char rescale /
shift / clamp) on random token streams: UTF-16 runs (incl. surrogate pairs and repeated boxes), special tags, math-BPE runs,
<NO-MATH>, eos / pad cut-off, <NOP> lines, high-res scale factors."""
import re
from collections import deque
from types import SimpleNamespace

import numpy as np
import pytest

from surya_amd.common.geometry import PolygonBox
from surya_amd.recognition.postprocess import clean_close_polygons
from surya_amd.recognition.processor import NOMATH_TOKEN, SuryaOCRProcessor
from surya_amd.recognition.schema import TaskNames, TextChar
from surya_amd.recognition.tokenizer import ByteMathTokenizer, OCRTokenizer


def reference_loop(proc, tokens, polys, scores):
    """The reference's get_bboxes_text for one line, token by token."""
    tk = proc.ocr_tokenizer
    blank = [[0, 0], [0, 1], [1, 1], [1, 0]]
    if proc.no_output_token in tokens:
        return None
    runs, cur, cur_kind = [], [], None
    for bbox, tid, s in zip(polys, tokens, scores):
        if tid in (proc.eos_token_id, proc.pad_token_id):
            break
        kind = "qwen" if tid < tk.qwen_offset else ("special" if tid < tk.special_token_offset else "ocr")
        if cur and (kind != cur_kind or kind == "special"):
            runs.append((cur_kind, cur)); cur = []
        cur.append((tid, s, bbox)); cur_kind = kind
    if cur:
        runs.append((cur_kind, cur))
    chars = []
    for kind, items in runs:
        ids, confs = [i[0] for i in items], [i[1] for i in items]
        if kind == "ocr":
            text = tk.decode(ids, task=TaskNames.ocr_with_boxes)
            boxes = clean_close_polygons([i[2] for i in items])
            bi = 0
            for ch in text:
                chars.append(TextChar(text=ch, polygon=boxes[bi], confidence=confs[bi], bbox_valid=True))
                if bi < len(boxes) - 1:
                    bi += 1
        elif kind == "special":
            text = tk.decode(ids, task=TaskNames.ocr_without_boxes)
            if text == NOMATH_TOKEN or re.match(r"<SCRIPT-\w+>", text):
                continue
            chars.append(TextChar(text=text, polygon=blank, confidence=confs[0], bbox_valid=False))
        else:
            chars.append(TextChar(text=tk.decode(ids, task=TaskNames.block_without_boxes), polygon=blank, confidence=confs[0], bbox_valid=False))
    return chars


def reference_geometry(chars, res_scale, polygon):
    box = PolygonBox(polygon=polygon)
    for c in chars:
        c.rescale(res_scale, (1, 1)); c.shift(box.bbox[0], box.bbox[1]); c.clamp(box.bbox)
    return chars


@pytest.mark.parametrize("seed", range(6))
def test_vectorised_assembly_equals_per_token_loop(seed):
    from surya_amd.recognition.predictor import RecognitionPredictor
    rng = np.random.default_rng(seed)
    tok = OCRTokenizer(None, ByteMathTokenizer(256), reserve_special=64)
    proc = SuryaOCRProcessor(tok)
    pred = object.__new__(RecognitionPredictor)
    pred.processor = proc
    sysm = tok.system_tokens
    specials = [v for k, v in tok.SPECIAL_TOKEN_MAPPING.items() if k not in sysm] + [sysm["<NO-MATH>"]]
    n_lines, T = 40, 30
    all_tokens, all_polys, all_scores = [], [], []
    for li in range(n_lines):
        toks = []
        while len(toks) < T:
            r = rng.random()
            if r < 0.55:                                   # UTF-16 run; astral chars make surrogate pairs
                for ch in rng.choice(list("abc xyzÄ漢😀"), size=int(rng.integers(1, 6))):
                    raw = ch.encode("utf-16le")
                    toks += [raw[i] + (raw[i + 1] << 8) + tok.special_token_offset for i in range(0, len(raw), 2)]
            elif r < 0.75:
                toks.append(int(rng.choice(specials)))
            elif r < 0.93:
                toks += [int(x) for x in rng.integers(32, 127, size=int(rng.integers(1, 5)))]      # math-BPE bytes
            else:
                toks.append(int(rng.choice([proc.eos_token_id, proc.pad_token_id])))
        toks = toks[:T]
        if li % 13 == 5:
            toks[int(rng.integers(0, T))] = proc.no_output_token
        polys = rng.uniform(-5, 300, size=(T, 4, 2)).astype(np.float32)
        for t in range(1, T):                              # repeated boxes: what multi-unit UTF-16 chars produce
            if rng.random() < 0.3:
                polys[t] = polys[t - 1] + rng.uniform(-0.05, 0.05, size=(4, 2)).astype(np.float32)
        all_tokens.append(toks); all_polys.append(polys); all_scores.append(rng.random(T).astype(np.float32).tolist())
    lines = pred.get_bboxes_text(None, all_tokens, all_scores, np.stack(all_polys))
    for li in range(n_lines):
        ref = reference_loop(proc, all_tokens[li], all_polys[li].tolist(), all_scores[li])
        if ref is None:
            assert lines[li] is None
            continue
        scale = (1, 1) if li % 3 else (1.37, 2.2)
        polygon = [[10.5, 20.25], [400.0, 21.0], [401.0, 90.75], [11.0, 88.0]]
        ref = reference_geometry(ref, scale, polygon)
        got = pred._chars_of(lines[li], scale, PolygonBox(polygon=polygon).bbox)
        assert [c.text for c in got] == [c.text for c in ref], li
        assert [c.bbox_valid for c in got] == [c.bbox_valid for c in ref]
        assert np.allclose([c.confidence for c in got], [c.confidence for c in ref], rtol=0, atol=0)
        assert [c.polygon for c in got] == [c.polygon for c in ref], li       # exact: same truncation, shift, clamp
        assert all(c.bbox == r.bbox for c, r in zip(got, ref))


@pytest.mark.parametrize("seed,return_words,drop", [(0, False, False), (1, True, False), (2, False, True), (3, False, False)])
def test_batched_assembly_equals_per_line(seed, return_words, drop):
    """RecognitionPredictor._assemble_batch (numpy work once per batch of finished lines, fast TextLine construction) against
    _assemble_line per item: every field of every TextLine / TextChar / TextWord, and the set of explicitly given fields."""
    from surya_amd.recognition.predictor import RecognitionPredictor
    rng = np.random.default_rng(100 + seed)
    tok = OCRTokenizer(None, ByteMathTokenizer(256), reserve_special=64)
    proc = SuryaOCRProcessor(tok)
    pred = object.__new__(RecognitionPredictor)
    pred.processor = proc
    sysm = tok.system_tokens
    specials = [v for k, v in tok.SPECIAL_TOKEN_MAPPING.items() if k not in sysm] + [sysm["<NO-MATH>"]]
    n = 60
    flat = {"polygons": [], "res_scales": [], "slices": []}
    items = []
    for li in range(n):
        T = int(rng.integers(1, 50))
        toks = []
        mode = rng.random()
        while len(toks) < T:
            r = rng.random() if mode < 0.6 else 0.0                                   # 40 % of the lines are plain text
            if r < 0.55:
                for ch in rng.choice(list("abc xyzÄ漢😀<"), size=int(rng.integers(1, 6))):
                    raw = ch.encode("utf-16le")
                    toks += [raw[i] + (raw[i + 1] << 8) + tok.special_token_offset for i in range(0, len(raw), 2)]
            elif r < 0.75:
                toks.append(int(rng.choice(specials)))
            elif r < 0.95:
                toks += [int(x) for x in rng.integers(32, 127, size=int(rng.integers(1, 5)))]
            else:
                toks.append(int(rng.choice([proc.eos_token_id, proc.pad_token_id])))
        toks = toks[:T]
        if li % 17 == 5:
            toks[int(rng.integers(0, T))] = proc.no_output_token
        if li % 19 == 7:
            toks = [toks[0], tok.special_token_offset + 65] * 30                        # trips detect_repeat_token
            T = len(toks)
        if li % 23 == 11:
            toks[0] = proc.eos_token_id                                                 # nothing before the stop token
        rows = np.sort(rng.integers(0, 1025, size=(T, 6)), axis=0).astype(np.float32)
        for t in range(1, T):
            if rng.random() < 0.3:
                rows[t] = rows[t - 1]                                                   # repeated boxes (multi-unit chars)
        sc = rng.random(T).astype(np.float32).tolist()
        if li % 11 == 3:
            sc[0] = float("nan")
        flat["polygons"].append([[10.5, 20.25], [400.0, 21.0], [401.0, 90.75], [11.0, 88.0]] if li % 2 else [7, 9, 300, 52])
        flat["res_scales"].append((1.0, 1.0) if li % 3 else (1.37, 2.2))
        flat["slices"].append(np.zeros((int(rng.integers(20, 80)), int(rng.integers(100, 600)), 3), np.uint8))
        items.append((li, li, toks, sc, rows))
    got = pred._assemble_batch(flat, items, drop, return_words, 1025)
    assert len(got) == n
    for it, g in zip(items, got):
        ref = pred._assemble_line(flat, it[0], it[1], it[2], it[3], it[4], drop, return_words, 1025)
        gd, rd = g.model_dump(), ref.model_dump()
        assert gd == rd or (str(gd) == str(rd)), it[0]                                  # str(): NaN-free by construction, exact floats
        assert g.model_fields_set == ref.model_fields_set, it[0]
        assert [c.model_fields_set for c in g.chars] == [c.model_fields_set for c in ref.chars]
        assert type(g.confidence) is type(ref.confidence) or g.confidence == ref.confidence
