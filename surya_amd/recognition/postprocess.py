"""Host-side output assembly helpers of RecognitionPredictor.

Behaviour of surya/recognition/util.py (unwrap_math :13-28, clean_math_tags :31-56, detect_repeat_token :59-69,
sort_text_lines :72-97, clean_close_polygons :100-120, words_from_chars :123-154, prediction_to_polygon_batch
:157-206) and surya/recognition/postprocessing.py (fix_unbalanced_tags :74-121). Pure Python / numpy, no device work.
"""
from __future__ import annotations

import re
from typing import Dict, List, Sequence, Tuple

import numpy as np

from ..common.geometry import PolygonBox
from .schema import TextChar, TextLine, TextWord

MATH_SYMBOLS = ["+", "-", "*", "=", "^", "_", "\\", "{", "}"]
_MATH_ONLY = re.compile(r'^\s*<math(?:\s+display="inline")?.*?</math>\s*$', re.DOTALL)
_MATH_BLOCK = re.compile(r"(<math\b[^>]*>)(.*?)</math>", flags=re.I | re.S)
_STRIP_TAGS = re.compile(r"</?(?:br|u|del|mark|i|b|sup|sub)\b[^>]*>", flags=re.I | re.S)
_TAG = re.compile(r"<(/?)([a-z]+)([^>]*)>?", re.IGNORECASE)


def unwrap_math(text: str) -> str:
    """Drop <math> wrappers around short spans that contain no LaTeX-ish symbol."""
    if len(text) > 50:
        return text
    if _MATH_ONLY.match(text) and text.count("<math") == 1 and not any(s in text for s in MATH_SYMBOLS):
        text = re.sub(r"<math.*?>", "", text)
        text = re.sub(r"</math>", "", text)
    return text


def clean_math_tags(html: str) -> str:
    """Strip formatting tags inside well-formed math blocks, drop empty blocks and orphan closing tags."""
    def inner(m):
        body = _STRIP_TAGS.sub("", m.group(2))
        return f"{m.group(1)}{body}</math>" if body.strip() else ""

    cleaned = _MATH_BLOCK.sub(inner, html)
    depth, parts = 0, []
    for tok in re.split(r"(</?math[^>]*>)", cleaned, flags=re.I):
        low = tok.lower()
        if low.startswith("<math"):
            depth += 1
            parts.append(tok)
        elif low == "</math>":
            if depth:
                depth -= 1
                parts.append(tok)
        else:
            parts.append(tok)
    return "".join(parts)


def detect_repeat_token(predicted_tokens: Sequence[int], max_repeats: int = 40) -> bool:
    """True when the last `max_repeats` tokens hold <= 5 distinct ids and the last u ids repeat the u before."""
    if len(predicted_tokens) < max_repeats:
        return False
    last = list(predicted_tokens[-max_repeats:])
    u = len(set(last))
    if u > 5:
        return False
    return last[-u:] == last[-u * 2: -u]


def sort_text_lines(lines, tolerance: float = 1.25):
    """Rough reading order: bucket by y, sort buckets by x. Keeps the reference's quirk that TextLine
    inputs are bucketed by the UNSCALED y (the `/ tolerance` only binds to the dict branch, util.py:78-85)."""
    groups = {}
    for line in lines:
        y = line.bbox[1] if isinstance(line, TextLine) else line["bbox"][1] / tolerance
        groups.setdefault(round(y) * tolerance, []).append(line)
    out = []
    for _, grp in sorted(groups.items()):
        out.extend(sorted(grp, key=lambda l: l.bbox[0] if isinstance(l, TextLine) else l["bbox"][0]))
    return out


def clean_close_polygons(bboxes: List[List[List[float]]], thresh: float = 0.1):
    """Drop a polygon when all 4 corners are within `thresh` of the previous one (multi-unit UTF-16 chars)."""
    if len(bboxes) < 2:
        return bboxes
    kept = [bboxes[0]]
    for prev, cur in zip(bboxes[:-1], bboxes[1:]):
        if any(abs(cur[j][0] - prev[j][0]) > thresh or abs(cur[j][1] - prev[j][1]) > thresh for j in range(4)):
            kept.append(cur)
    return kept


def words_from_chars(chars: List[TextChar], line_box: PolygonBox) -> List[TextWord]:
    words, word = [], None
    for i, ch in enumerate(chars):
        if not ch.bbox_valid:
            if word:
                words.append(word)
                word = None
            continue
        if not word:
            word = TextWord(**ch.model_dump())
            if i == 0:
                word.merge_left(line_box)
        elif not ch.text.strip():
            words.append(word)
            word = None
        else:
            word.merge(ch)
            word.text = word.text + ch.text
            if i == len(chars) - 1:
                word.merge_right(line_box)
    if word:
        words.append(word)
    return words


def prediction_to_polygon_batch(pred: np.ndarray, img_sizes: Sequence[Tuple[int, ...]], bbox_scaler, skew_scaler,
                                skew_min: float = 0.001) -> np.ndarray:
    """(cx, cy, w, h, skew_x, skew_y) tokens [N, T, 6] -> polygons [N, T, 4, 2] scaled to each crop (h, w)."""
    pred = np.asarray(pred, np.float32)
    sizes = np.asarray([s[:2] for s in img_sizes], np.float32)
    w_scale = (sizes[:, 1] / np.float32(bbox_scaler))[:, None, None]
    h_scale = (sizes[:, 0] / np.float32(bbox_scaler))[:, None, None]
    cx, cy, bw, bh = pred[..., 0], pred[..., 1], pred[..., 2], pred[..., 3]
    x1, y1, x2, y2 = cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2
    sx = np.floor((pred[..., 4] - skew_scaler) / 2)
    sy = np.floor((pred[..., 5] - skew_scaler) / 2)
    sx[np.abs(sx) < skew_min] = 0
    sy[np.abs(sy) < skew_min] = 0
    flat = np.stack([x1 - sx, y1 - sy, x2 - sx, y1 + sy, x2 + sx, y2 + sy, x1 + sx, y2 - sy], axis=2)
    polys = flat.reshape(pred.shape[0], pred.shape[1], 4, 2).astype(np.float32)
    polys[..., 0] *= w_scale
    polys[..., 1] *= h_scale
    return polys


def _closing_tag_names(tags: Sequence[str]) -> List[str]:
    names = []
    for t in tags:
        m = _TAG.match(t)
        if m and m.group(1) == "/":
            names.append(m.group(2))
    return names


_CLOSABLE: Dict[int, tuple] = {}


def fix_unbalanced_tags(text_chars: List[TextChar], special_tokens: Dict[str, list]) -> List[TextChar]:
    """Append closing tags for formatting / math tags left open at the end of a line."""
    key = id(special_tokens)                    # the tag table of a tokenizer never changes: parse it once (16 us per call before)
    hit = _CLOSABLE.get(key)
    if hit is None or hit[0] is not special_tokens:
        hit = (special_tokens, _closing_tag_names(special_tokens["formatting"]) + _closing_tag_names(special_tokens["math_external"]))
        _CLOSABLE[key] = hit
    closable = hit[1]
    stack: List[str] = []
    for ch in text_chars:
        if len(ch.text) <= 1:
            continue
        m = _TAG.match(ch.text)
        if not m:
            continue
        name = m.group(2).lower()
        if name not in closable or name == "br":
            continue
        if m.group(3) and m.group(3).strip().endswith("/"):
            continue
        if m.group(1) == "/":
            if stack and stack[-1] == name:
                stack.pop()
        else:
            stack.append(name)
    for name in stack:
        text_chars.append(TextChar(text=f"</{name}>", confidence=0, polygon=[[0, 0], [1, 0], [1, 1], [0, 1]], bbox_valid=False))
    return text_chars
