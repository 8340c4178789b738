"""Golden vectors of the reference's HOST logic, recorded from the real reference package in the build container (oracle/ref_shim)
so that they travel with the repo (tests/golden/host_reference.json, checked by tests/test_oracle_golden.py anywhere):

  * assembly: random token streams (UTF-16 runs incl. surrogate pairs, special tags, math-BPE runs, eos / pad, <NOP>, repeated boxes)
    through RecognitionPredictor.get_bboxes_text (recognition/__init__.py:609-771) + the per-line tail of __call__ (:886-925) with the
    reference's own TextChar / TextLine / PolygonBox classes and helpers -> TextLine dumps;
  * tokenizer: InnerOCRTokenizer (common/surya/processor/tokenizer.py:26-221) on the stand-in math tokenizer: text -> ids, ids -> text.

    python oracle/make_golden_host.py
"""
import json
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def main():
    from types import SimpleNamespace
    sr = ref_shim.import_recognition()
    from surya.common.polygon import PolygonBox as RefBox
    from surya.recognition.postprocessing import fix_unbalanced_tags as ref_fix
    from surya.recognition.util import clean_math_tags, unwrap_math, words_from_chars, prediction_to_polygon_batch
    from surya.recognition.schema import TextLine as RefLine
    import surya.common.surya.processor.tokenizer as rt
    from surya_amd.recognition.processor import SuryaOCRProcessor
    from surya_amd.recognition.tokenizer import ByteMathTokenizer, OCRTokenizer, DEFAULT_SPECIAL_TOKENS

    tok = OCRTokenizer(None, ByteMathTokenizer(256), reserve_special=64)
    proc = SuryaOCRProcessor(tok)
    ref_self = SimpleNamespace(processor=proc, tasks=sr.RecognitionPredictor.tasks)
    sysm = tok.system_tokens
    specials = [v for k, v in tok.SPECIAL_TOKEN_MAPPING.items() if k not in sysm] + [sysm["<NO-MATH>"]]
    rng = np.random.default_rng(20260921)
    bbox_size, lines = 1025, []
    for li in range(64):
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
        polygon = [[10.5, 20.25], [400.0, 21.0], [401.0, 90.75], [11.0, 88.0]] if li % 2 else [7, 9, 300, 52]
        res_scale = (1.0, 1.0) if li % 3 else (1.37, 2.2)
        shape = [int(rng.integers(20, 80)), int(rng.integers(100, 600)), 3]
        words = bool(li % 4 == 1)
        polys = prediction_to_polygon_batch(torch.from_numpy(rows)[None], [tuple(shape)], bbox_size, bbox_size // 2)
        chars = sr.RecognitionPredictor.get_bboxes_text(ref_self, {"slices": [np.zeros(shape, np.uint8)], "task_names": ["ocr_with_boxes"]},
                                                        [toks], [sc], polys)[0]
        if not chars:
            line = RefLine(text="", polygon=polygon, chars=[], confidence=1, original_text_good=True)
        else:
            confidence = float(np.mean([c.confidence for c in chars]))
            box = RefBox(polygon=polygon)
            for c in chars:
                c.rescale(res_scale, (1, 1)); c.shift(box.bbox[0], box.bbox[1]); c.clamp(box.bbox)
            chars = ref_fix(chars, tok.special_tokens)
            text = clean_math_tags(unwrap_math("".join(c.text for c in chars)))
            line = RefLine(text=text, polygon=polygon, chars=chars, confidence=confidence, words=words_from_chars(chars, box) if words else [])
        lines.append({"tokens": toks, "scores": sc, "rows": rows.tolist(), "polygon": polygon, "res_scale": list(res_scale),
                      "slice_shape": shape, "return_words": words, "expected": line.model_dump()})

    ref_tok = rt.InnerOCRTokenizer(special_tokens=DEFAULT_SPECIAL_TOKENS, qwen_tokenizer=ByteMathTokenizer(300))
    prng = random.Random(77)
    tags = [t for k in ("formatting", "math_external", "system") for t in DEFAULT_SPECIAL_TOKENS.get(k, [])]
    pieces = list("abc xyz0189.,;-ÄøЖ漢字😀🧪") + ["&lt;", "&amp;", "&gt;", "<", ">", "x^2", "\\frac{a}{b}", "<br>", "</"]
    enc = []
    for _ in range(200):
        parts = []
        for _ in range(prng.randint(0, 12)):
            r = prng.random()
            parts.append(prng.choice(tags) if r < 0.25 else ("<math>" + "".join(prng.choice(pieces) for _ in range(prng.randint(0, 5))) + "</math>")
                         if r < 0.35 else prng.choice(pieces))
        text = "".join(parts)
        if text.count("<math") > text.count("</math>"):
            text += "</math>"
        enc.append({"text": text, "ids": ref_tok._tokenize(text)})
    sp = sorted(ref_tok.REVERSE_SPECIAL_TOKEN_MAPPING)
    off = ref_tok.qwen_token_offset + ref_tok.SPECIAL_TOKEN_OFFSET
    dec = []
    for _ in range(200):
        ids = []
        for _ in range(prng.randint(0, 30)):
            r = prng.random()
            ids.append(off + prng.choice([prng.randint(0x20, 0x7e), prng.randint(0xa0, 0xd7ff), prng.randint(0xd800, 0xdfff)]) if r < 0.5
                       else prng.choice(sp) if r < 0.75 else prng.randint(0, 255))
        dec.append({"ids": ids, "text": ref_tok.decode(ids)})
    out = {"note": "recorded by oracle/make_golden_host.py from the real reference (VikParuchuri/surya v0.14.6) host code",
           "assembly": {"bbox_size": bbox_size, "tokenizer": {"math_size": 256, "reserve_special": 64}, "lines": lines},
           "tokenizer": {"math_size": 300, "encode": enc, "decode": dec}}
    path = os.path.join(GOLD, "host_reference.json")
    with open(path, "w", encoding="utf-8") as f:
        json.dump(out, f, ensure_ascii=True)
    print(path, os.path.getsize(path))


if __name__ == "__main__":
    main()
