"""OCR tokenizer: UTF-16 code units + special tags + (pluggable) BPE for math spans.

Behaviour of SuryaOCRTokenizer / InnerOCRTokenizer (surya/common/surya/processor/tokenizer.py:26-320):
id layout  [0, qwen_offset) math BPE  |  [qwen_offset, special_token_offset) tags  |  then one id per UTF-16 code unit.
The Qwen2 BPE vocabulary ships with the checkpoint (unavailable offline); `ByteMathTokenizer` is the synthetic stand-in
(one id per UTF-8 byte) so the three-range decode logic is exercised end to end.
"""
from __future__ import annotations

import html
import re
from typing import Dict, List, Optional, Sequence

from .schema import TaskNames, TASK_NAMES

DEFAULT_SPECIAL_TOKENS: Dict[str, List[str]] = {
    "system": ["</S>", "<PAD>", "<IMAGE>", "<REG1>", "<REG2>", "<REG3>", "<REG4>", "<EOI>", "<NOP>", "<ROT>", "<NO-MATH>",
               "<OCR-WB>", "<OCR-WOB>", "<BLOCKS-WOB>"],
    "formatting": ["<b>", "</b>", "<i>", "</i>", "<u>", "</u>", "<del>", "</del>", "<sup>", "</sup>", "<sub>", "</sub>",
                   "<mark>", "</mark>", "<br>", "<code>", "</code>"],
    "math_external": ["<math>", '<math display="inline">', '<math display="block">', "</math>"],
}
DEFAULT_SPECIAL_TOKENS["all"] = (DEFAULT_SPECIAL_TOKENS["system"] + DEFAULT_SPECIAL_TOKENS["formatting"]
                                 + DEFAULT_SPECIAL_TOKENS["math_external"])


class ByteMathTokenizer:
    """Stand-in for the checkpoint's Qwen2 BPE: ids 0..255 = UTF-8 bytes; `size` ids reserved."""

    def __init__(self, size: int):
        assert size >= 256
        self.size = size

    def __len__(self):
        return self.size

    def __call__(self, text: str):
        return {"input_ids": list(text.encode("utf-8"))}

    def decode(self, ids: Sequence[int]) -> str:
        return bytes(i for i in ids if 0 <= i < 256).decode("utf-8", errors="ignore")


def _prefix_regex(tokens: Sequence[str]):
    """Longest-first alternation anchored at the start (tokenizer.py:18-23); never matches when empty."""
    if not tokens:
        return re.compile(r"(?!x)x")
    return re.compile("^(" + "|".join(re.escape(t) for t in sorted(tokens, key=len, reverse=True)) + ")")


class OCRTokenizer:
    def __init__(self, special_tokens: Optional[Dict[str, List[str]]] = None, math_tokenizer=None, qwen_offset: int = 256,
                 reserve_special: int = 0):
        self.special_tokens = special_tokens or DEFAULT_SPECIAL_TOKENS
        self.math_tokenizer = math_tokenizer or ByteMathTokenizer(qwen_offset)
        self.qwen_offset = len(self.math_tokenizer)
        self.SPECIAL_TOKEN_MAPPING: Dict[str, int] = {}
        for tag in self.special_tokens.get("all", []):
            if tag not in self.SPECIAL_TOKEN_MAPPING:
                self.SPECIAL_TOKEN_MAPPING[tag] = self.qwen_offset + len(self.SPECIAL_TOKEN_MAPPING)
        # synthetic configs reserve a fixed-size tag range (RecConfig.num_special_tokens): a randomly initialised model can
        # emit any id, so every reserved id gets a placeholder tag (a real checkpoint defines all of its tag ids)
        for n in range(len(self.SPECIAL_TOKEN_MAPPING), reserve_special):
            self.SPECIAL_TOKEN_MAPPING[f"<RSV-{n}>"] = self.qwen_offset + n
        self.num_special = len(self.SPECIAL_TOKEN_MAPPING)
        self.REVERSE_SPECIAL_TOKEN_MAPPING = {v: k for k, v in self.SPECIAL_TOKEN_MAPPING.items()}
        self.special_token_offset = self.qwen_offset + self.num_special
        self._system = _prefix_regex(self.special_tokens.get("system", []))
        self._math = _prefix_regex(self.special_tokens.get("math_external", []))
        self._format = _prefix_regex(self.special_tokens.get("formatting", []))
        self.system_tokens = {t: self.SPECIAL_TOKEN_MAPPING[t] for t in self.special_tokens.get("system", [])}

    @property
    def vocab_size(self) -> int:
        return self.special_token_offset + 65536

    # --------------------------------------------------------------------------------------------- encode
    def _tokenize_ocr(self, text: str) -> List[int]:
        """InnerOCRTokenizer._tokenize (tokenizer.py:68-125)."""
        out: List[int] = []
        in_math = False
        text = html.unescape(text)
        while text:
            m = self._system.search(text)
            if m:
                out.append(self.SPECIAL_TOKEN_MAPPING[m.group(1)]); text = text[m.end():]
                continue
            m = self._math.search(text)
            if m:
                tag = m.group(1)
                if tag.startswith("<math"):
                    in_math = True
                elif tag == "</math>":
                    in_math = False
                out.append(self.SPECIAL_TOKEN_MAPPING[tag]); text = text[m.end():]
                continue
            if in_math:
                end = text.find("</math>")
                if end < 0:
                    # unterminated <math>: the whole remainder is math. The reference slices text[:-1] / text[-1:] here
                    # (:99-101) and never terminates; a hang in the pre-processing pool is not behaviour worth keeping.
                    out += self.math_tokenizer(text)["input_ids"]
                    text = ""
                    continue
                span = text[:end]
                out += self.math_tokenizer(span)["input_ids"]
                text = text[end:]
                continue
            m = self._format.search(text)
            if m:
                out.append(self.SPECIAL_TOKEN_MAPPING[m.group(1)]); text = text[m.end():]
                continue
            raw = text[0].encode("utf-16le")
            out += [raw[i] + (raw[i + 1] << 8) + self.special_token_offset for i in range(0, len(raw), 2)]
            text = text[1:]
        return out

    def __call__(self, texts, tasks=None):
        if isinstance(texts, str):
            texts, tasks = [texts], [tasks]
        ids = []
        for text, task in zip(texts, tasks):
            task = task or TaskNames.ocr_with_boxes
            assert task in TASK_NAMES, f"Invalid task: {task}"
            if task in (TaskNames.ocr_with_boxes, TaskNames.ocr_without_boxes):
                ids.append(self._tokenize_ocr(text))
            else:
                ids.append(list(self.math_tokenizer(text)["input_ids"]))
        return {"input_ids": ids}

    # --------------------------------------------------------------------------------------------- decode
    def _decode_ocr(self, token_ids: Sequence[int]) -> str:
        """InnerOCRTokenizer.decode (tokenizer.py:173-221): runs of math ids / UTF-16 ids / single tags."""
        text, buf, buf_math = "", [], False

        def flush():
            nonlocal text, buf, buf_math
            if buf:
                if buf_math:
                    text += self.math_tokenizer.decode(buf)
                else:
                    b = bytearray()
                    for t in buf:
                        n = t - self.special_token_offset
                        b.append(n & 0xFF); b.append((n >> 8) & 0xFF)
                    text += b.decode("utf-16le", errors="ignore")
            buf, buf_math = [], False

        for t in token_ids:
            if t < self.qwen_offset:
                if buf and not buf_math:
                    flush()
                buf.append(t); buf_math = True
            elif t >= self.special_token_offset:
                if buf and buf_math:
                    flush()
                buf.append(t); buf_math = False
            elif t in self.REVERSE_SPECIAL_TOKEN_MAPPING:
                flush()
                text += self.REVERSE_SPECIAL_TOKEN_MAPPING[t]
            else:
                raise ValueError(f'Unexpected token value while decoding, got "{t}" in token_ids {list(token_ids)}')
        flush()
        return text

    def decode(self, token_ids, task: str = TaskNames.ocr_with_boxes) -> str:
        assert task in TASK_NAMES, f"Invalid task: {task}"
        token_ids = [int(t) for t in token_ids]
        if task in (TaskNames.ocr_with_boxes, TaskNames.ocr_without_boxes):
            return self._decode_ocr(token_ids)
        return self.math_tokenizer.decode(token_ids)
