"""Host-side pre-processing of line crops into the model's boundary tensors.

Behaviour of SuryaOCRProcessor (surya/common/surya/processor/__init__.py:42-433): area clamp (scale_to_fit,
:141-178), round-up to multiples of 28 and normalisation (:180-230), patchify in merge-block-major order
(:214-228), prompt ids (:233-329) and -- different from the reference by design -- PACKED sequences instead of a
left-padded batch (:386-403): the HIP path takes per-sequence prompts and derives positions from them, which is
what `position_ids = cumsum(mask) - 1` computes for the real tokens.
cv2 is unavailable: resizes use surya_amd.common.imageops (same geometry, not pinned against OpenCV).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np

from ..common import imageops
from .schema import TaskNames
from .tokenizer import OCRTokenizer

EOS_TOKEN, EOI_TOKEN, IMAGE_TOKEN, PAD_TOKEN = "</S>", "<EOI>", "<IMAGE>", "<PAD>"
NO_OUTPUT_TOKEN, IMAGE_ROTATED_TOKEN, NOMATH_TOKEN = "<NOP>", "<ROT>", "<NO-MATH>"
REGISTER_TOKENS = ["<REG1>", "<REG2>", "<REG3>", "<REG4>"]
BOS_TOKENS = {TaskNames.ocr_with_boxes: "<OCR-WB>", TaskNames.ocr_without_boxes: "<OCR-WOB>",
              TaskNames.block_without_boxes: "<BLOCKS-WOB>"}


class SuryaOCRProcessor:
    rescale_factor = 1 / 255.0
    image_mean = np.array((0.485, 0.456, 0.406), np.float32)
    image_std = np.array((0.229, 0.224, 0.225), np.float32)

    def __init__(self, ocr_tokenizer: OCRTokenizer, num_register_tokens: int = 4, patch_size: int = 14, merge_size: int = 2):
        self.ocr_tokenizer = ocr_tokenizer
        self.patch_size, self.merge_size, self.num_register_tokens = patch_size, merge_size, num_register_tokens
        sys = ocr_tokenizer.system_tokens
        self.register_token_ids = [sys.get(r) for r in REGISTER_TOKENS]
        self.image_token_id, self.pad_token_id = sys.get(IMAGE_TOKEN), sys.get(PAD_TOKEN)
        self.eos_token_id, self.eoi_token_id = sys.get(EOS_TOKEN), sys.get(EOI_TOKEN)
        self.no_output_token, self.image_rotated_token = sys.get(NO_OUTPUT_TOKEN), sys.get(IMAGE_ROTATED_TOKEN)
        self.nomath_token = sys.get(NOMATH_TOKEN)
        self.bos_token_id = {task: sys.get(tok) for task, tok in BOS_TOKENS.items()}
        if num_register_tokens > len(self.register_token_ids):
            raise ValueError("more register tokens requested than defined in the special token mapping")

    def image_processor(self, image) -> np.ndarray:
        return np.asarray(image, dtype=np.float32)          # processor/__init__.py:135-138

    @staticmethod
    def scale_to_fit(img: np.ndarray, max_size: Tuple[int, int], min_size: Tuple[int, int] = (168, 168)) -> np.ndarray:
        """Pixel-COUNT clamp, not a bounding box (:141-178)."""
        h, w = img.shape[:2]
        if w == 0 or h == 0:
            return img
        cur, mx, mn = w * h, max_size[0] * max_size[1], min_size[0] * min_size[1]
        if cur > mx:
            s = (mx / cur) ** 0.5
            nw, nh = math.floor(w * s), math.floor(h * s)
        elif cur < mn:
            s = (mn / cur) ** 0.5
            nw, nh = math.ceil(w * s), math.ceil(h * s)
        else:
            return img
        return imageops.resize(img, nw, nh, "lanczos4")

    def _normalise(self, image: np.ndarray) -> np.ndarray:
        image = image.astype(np.float64) * self.rescale_factor          # /255 in fp64 (:181)
        return (image.astype(np.float32) - self.image_mean) / self.image_std

    def process_and_tile(self, image: np.ndarray) -> Tuple[np.ndarray, Tuple[int, int]]:
        """[H, W, 3] float (0..255) -> tiles [gh*gw, 3*ps*ps] float32 in merge-block-major row order, (gh, gw)."""
        f = self.patch_size * self.merge_size
        h, w = image.shape[:2]
        hb, wb = math.ceil(h / f) * f, math.ceil(w / f) * f
        if (hb, wb) != (h, w):
            image = imageops.resize(image, wb, hb, "cubic")
        image = self._normalise(image)
        ps, m = self.patch_size, self.merge_size
        gh, gw = hb // ps, wb // ps
        x = image.transpose(2, 0, 1).reshape(3, gh // m, m, ps, gw // m, m, ps)
        # (C, bh, dy, py, bw, dx, px) -> (bh, bw, dy, dx, C, py, px): 4 consecutive rows = one 2x2 merge group
        x = x.transpose(1, 4, 2, 5, 0, 3, 6).reshape(gh * gw, 3 * ps * ps)
        return np.ascontiguousarray(x, dtype=np.float32), (gh, gw)

    def prompt_ids(self, n_image_tokens: int, task: str, text: str = "", math_mode: bool = True, rotated: bool = False):
        """[<ROT>] <IMAGE>*n REG1..k BOS [<NO-MATH>] text <EOI> (:247-252, 264-269, 312)."""
        ids = [self.image_token_id] * n_image_tokens + self.register_token_ids[: self.num_register_tokens]
        if rotated:
            ids = [self.image_rotated_token] + ids
        text_ids = list(self.ocr_tokenizer(text, task)["input_ids"][0])
        if not math_mode:
            text_ids.insert(0, self.nomath_token)
        return ids + [self.bos_token_id[task]] + text_ids + [self.eoi_token_id]

    def __call__(self, mixed_batch: List[dict]) -> Dict[str, object]:
        """batch of {"task", "inputs": [image dict, text dict]} -> packed tensors for HipRecModel.prefill."""
        tiles, grids, seqs = [], [], []
        for b in mixed_batch:
            task = b["task"]
            assert task in self.bos_token_id, f"Task {task} has no bos token defined."
            img_in, txt_in = b["inputs"][0], b["inputs"][1]
            assert img_in["type"] == "image" and txt_in["type"] == "text" and len(b["inputs"]) == 2
            t, (gh, gw) = self.process_and_tile(img_in["image"])
            n = t.shape[0] // self.merge_size ** 2
            seqs.append(self.prompt_ids(n, task, txt_in.get("text") or "", txt_in.get("math", False),
                                        img_in.get("rotated", False)))
            tiles.append(t); grids.append((gh, gw))
        return {"image_tiles": np.concatenate(tiles, 0) if tiles else np.zeros((0, 588), np.float32),
                "grid_hw": np.asarray(grids, np.int32).reshape(-1, 2), "input_ids": seqs}
