"""RecognitionPredictor: drop-in for surya.recognition.RecognitionPredictor on MI355X.

Same call signature, class attributes and output schema as the reference
(surya/recognition/__init__.py:77-102, 773-942). What changes is below the plugin seam:
  * `self.model` is a HipRecModel (libsurya_amd.so); prefill / decode / process_outputs and the KV-cache merge
    (reference :294-471, cache.py) are single library calls on a slot-based cache;
  * prompts are pre-processed once up front (thread pool) and their tiles stay resident in HBM;
  * the continuous-batching policy is the reference's (:539-599: prefill when more than `min_prefill_ratio` of the
    slots are empty, same stop rules), but decode runs `RECOGNITION_STEPS_PER_SYNC` device-resident steps per host
    round trip. Per-line outputs do not depend on batch composition: attention and positions are per-sequence, every GEMM
    tile shape walks K in the same order, and the split-K slice count / lm_head partial width depend on the weight shape only
    (tests/test_gpu_fullsize.py demands identical tokens, boxes and scores across slot counts); in fp32 mode the emitted
    tokens are bit-identical to the oracle's.
There is no CPU fallback: without the HIP library and a GPU, construction raises.
"""
from __future__ import annotations

import json
import os
import re
import time
from array import array
from collections import deque
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass
from typing import List, Optional

import numpy as np
import torch
from PIL import Image

from ..common.geometry import PolygonBox, coerce_polygon
from ..common.imageops import fill_poly_mask
from ..common.predictor import BasePredictor, ModelLoader, gc_paused
from ..config import RecConfig, rec_config
from ..settings import settings
from .model import HipRecModel
from .preprocess_gpu import DevicePreprocessor, LineRef, bbox_ref, page_pixels, poly_ref
from .postprocess import (clean_math_tags, detect_repeat_token, fix_unbalanced_tags,
                          prediction_to_polygon_batch, sort_text_lines, unwrap_math, words_from_chars)
from .processor import NOMATH_TOKEN, SuryaOCRProcessor
from .schema import OCRResult, TaskNames, TextChar, TextLine
from .tokenizer import ByteMathTokenizer, OCRTokenizer


_SCRIPT_TAG = re.compile(r"<SCRIPT-\w+>")
_CHAR_FIELDS = frozenset(("polygon", "confidence", "text", "bbox_valid"))
assert _CHAR_FIELDS == frozenset(TextChar.model_fields), "TextChar fields changed: update _text_char"
_new_char, _set = TextChar.__new__, object.__setattr__


def _text_char(polygon, confidence, text, bbox_valid) -> TextChar:
    """TextChar.model_construct(...) with all four fields given, without its per-call field loop (1.9 -> 0.55 us; a page of
    text is ~10^4 of these). Same object state: __dict__, fields_set, no extras, no private attributes."""
    m = _new_char(TextChar)
    _set(m, "__dict__", {"polygon": polygon, "confidence": confidence, "text": text, "bbox_valid": bbox_valid})
    _set(m, "__pydantic_fields_set__", set(_CHAR_FIELDS))
    _set(m, "__pydantic_extra__", None)
    _set(m, "__pydantic_private__", None)
    return m


_LINE_FIELDS = ("polygon", "confidence", "text", "chars", "original_text_good", "words")
assert frozenset(_LINE_FIELDS) == frozenset(TextLine.model_fields), "TextLine fields changed: update _text_line"
_new_line = TextLine.__new__
_BLANK_POLY = np.array([[0, 0], [0, 1], [1, 1], [1, 0]], np.float64)


def _text_line(polygon, confidence, text, chars, words) -> TextLine:
    """TextLine(...) for values that are already in validated form (polygon = coerce_polygon(...), confidence not NaN, chars a
    list of TextChar): the object state validation would produce, without walking the character list again."""
    m = _new_line(TextLine)
    _set(m, "__dict__", {"polygon": polygon, "confidence": confidence, "text": text, "chars": chars,
                         "original_text_good": False, "words": words})
    _set(m, "__pydantic_fields_set__", {"polygon", "confidence", "text", "chars", "words"})     # as TextLine(text=, polygon=, ...)
    _set(m, "__pydantic_extra__", None)
    _set(m, "__pydantic_private__", None)
    return m


# ------------------------------------------------------------------------------------------------ input slicing
def convert_if_not_rgb(images: List[Image.Image]) -> List[Image.Image]:
    return [im if im.mode == "RGB" else im.convert("RGB") for im in images]


def slice_bboxes_from_image(image: np.ndarray, bboxes) -> List[np.ndarray]:
    """Axis-aligned crops, boxes clipped into the page (surya/input/processing.py:35-54)."""
    lines = []
    for bbox in bboxes:
        b = np.clip(np.array(bbox, dtype=np.int32), 0, None)
        if b[3] <= b[1]:
            b[3] = b[1] + 1
        if b[2] <= b[0]:
            b[2] = b[0] + 1
        b[2] = min(b[2], image.shape[1])
        b[3] = min(b[3], image.shape[0])
        lines.append(image[b[1]: b[3], b[0]: b[2]].copy())
    return lines


def slice_and_pad_poly(image: np.ndarray, coordinates) -> np.ndarray:
    """Crop the polygon's bounding box and paint everything outside the polygon with the pad value
    (surya/input/processing.py:64-101; fillPoly replaced by common.imageops.fill_poly_mask)."""
    pts = [(int(c[0]), int(c[1])) for c in coordinates]
    x0, y0 = min(p[0] for p in pts), min(p[1] for p in pts)
    x1, y1 = max(p[0] for p in pts), max(p[1] for p in pts)
    crop = image[y0:y1, x0:x1].copy()
    h, w = crop.shape[:2]
    if y1 <= y0 or x1 <= x0 or len(pts) < 3 or h == 0 or w == 0:
        return crop
    mask = fill_poly_mask(h, w, [(x - x0, y - y0) for x, y in pts])
    crop[mask == 0] = settings.RECOGNITION_PAD_VALUE
    return crop


def slice_polys_from_image(image: np.ndarray, polys) -> List[np.ndarray]:
    return [slice_and_pad_poly(image, p) for p in polys]


# ------------------------------------------------------------------------------------------------------ loader
class RecognitionModelLoader(ModelLoader):
    """`checkpoint` may be None (synthetic config named by SURYA_AMD_REC_CONFIG), a dict
    {"config": RecConfig, "state_dict": {...}}, or a directory holding the reference's HF-format files
    (config.json + *.safetensors; recognition/loader.py:25-82)."""

    def __init__(self, checkpoint=None):
        super().__init__(checkpoint)
        self._cfg: Optional[RecConfig] = None
        self._sd = None
        self._special_tokens = None

    def _resolve(self):
        if self._cfg is not None:
            return
        ck = self.checkpoint
        if isinstance(ck, dict):
            self._cfg, self._sd = ck["config"], ck["state_dict"]
            self._special_tokens = ck.get("special_tokens")
        elif isinstance(ck, str) and os.path.isdir(ck):
            from safetensors.torch import load_file
            with open(os.path.join(ck, "config.json")) as f:
                raw = json.load(f)
            self._cfg = rec_config_from_reference_json(raw)
            self._special_tokens = raw.get("special_ocr_tokens")
            self._sd = None if self._receives_weights() else {}
            for fn in sorted(os.listdir(ck)):
                if fn.endswith(".safetensors") and self._sd is not None:
                    self._sd.update(load_file(os.path.join(ck, fn)))
        else:
            from ..synth import make_rec_weights
            self._cfg = rec_config(ck if isinstance(ck, str) else settings.SURYA_AMD_REC_CONFIG)
            self._sd = None if self._receives_weights() else make_rec_weights(self._cfg, 0)

    @staticmethod
    def _receives_weights() -> bool:
        """SURYA_AMD_BROADCAST_WEIGHTS with an initialised process group: only rank 0 reads / builds the state dict."""
        if not settings.SURYA_AMD_BROADCAST_WEIGHTS:
            return False
        from .. import dist as sdist
        rank, world = sdist.world_info()
        return world > 1 and rank != 0

    def tokenizer(self) -> OCRTokenizer:
        self._resolve()
        if isinstance(self.checkpoint, str) and os.path.isdir(self.checkpoint):
            # Real checkpoint: the id layout is DEFINED by the files (processor/tokenizer.py:224-260) -- the Qwen2 BPE that
            # ships with it sets qwen_offset, special_ocr_tokens["all"] sets the tag range, exactly len(unique tags) wide.
            # No placeholder tags, no byte-tokenizer stand-in: either would shift every UTF-16 id silently.
            from transformers import Qwen2Tokenizer
            math_tok = Qwen2Tokenizer.from_pretrained(self.checkpoint)      # raises if the vocabulary files are missing
            if not self._special_tokens or not self._special_tokens.get("all"):
                raise ValueError(f"{self.checkpoint}/config.json has no special_ocr_tokens; cannot lay out token ids")
            tok = OCRTokenizer(self._special_tokens, math_tok, reserve_special=0)
            # the lm_head may be PADDED beyond the tokenizer (the reference never ties the two sizes); ids the tokenizer does not
            # know can then be emitted and decode to nothing. A tokenizer LARGER than the head cannot be right.
            if tok.vocab_size > self._cfg.decoder.vocab_size:
                raise ValueError(f"token-id layout mismatch: qwen_offset {tok.qwen_offset} + {tok.num_special} tags + 65536 "
                                 f"UTF-16 units = {tok.vocab_size} > decoder.vocab_size = {self._cfg.decoder.vocab_size}")
            if tok.vocab_size < self._cfg.decoder.vocab_size:
                import warnings
                warnings.warn(f"decoder.vocab_size {self._cfg.decoder.vocab_size} exceeds the tokenizer's {tok.vocab_size} ids "
                              "(padded lm_head); ids beyond the tokenizer decode to nothing")
            return tok
        # synthetic configs only: one id per UTF-8 byte stands in for the BPE, and the tag range is padded to the
        # config's fixed width (a randomly initialised model can emit any id)
        return OCRTokenizer(self._special_tokens, ByteMathTokenizer(self._cfg.qwen_offset),
                            reserve_special=self._cfg.num_special_tokens)

    def model(self, device=None, dtype=None, **caps) -> HipRecModel:
        self._resolve()
        if device is None:
            device = settings.TORCH_DEVICE_MODEL
        if device == "cuda":
            device = "cuda:0"
        if dtype is None:
            dtype = torch.bfloat16          # recognition/loader.py:35-38 picks bf16 on GPUs with native bf16
        tok = self.tokenizer()
        sysm = tok.system_tokens
        caps.setdefault("max_slots", settings.RECOGNITION_BATCH_SIZE or RecognitionPredictor.default_batch_sizes["cuda"])
        caps.setdefault("max_kv_len", 1536 + 32)
        if settings.SURYA_AMD_BROADCAST_WEIGHTS:
            from .. import dist as sdist
            caps.setdefault("broadcast_weights", sdist.collectives_on())
        return HipRecModel(self._cfg, self._sd, image_token_id=sysm["<IMAGE>"], pad_token_id=sysm["<PAD>"],
                           eos_token_id=sysm["</S>"], dtype=dtype, device=device, **caps)

    def processor(self, device=None, dtype=None) -> SuryaOCRProcessor:
        self._resolve()
        e = self._cfg.encoder
        return SuryaOCRProcessor(self.tokenizer(), self._cfg.num_register_tokens, e.patch_size, e.spatial_merge_size)


def rec_config_from_reference_json(raw: dict) -> RecConfig:
    """Map a SuryaModelConfig config.json (surya/common/surya/config.py) onto RecConfig."""
    from ..config import DecoderConfig, EncoderConfig
    ve, de = raw.get("vision_encoder", {}), raw.get("decoder", {})
    enc = EncoderConfig(**{k: (tuple(v) if isinstance(v, list) else v) for k, v in ve.items()
                           if k in EncoderConfig.__dataclass_fields__})
    dkw = {k: v for k, v in de.items() if k in DecoderConfig.__dataclass_fields__}
    if "head_dim" not in dkw and "hidden_size" in dkw and "num_attention_heads" in dkw:
        dkw["head_dim"] = dkw["hidden_size"] // dkw["num_attention_heads"]
    dec = DecoderConfig(**dkw)
    return RecConfig(name="checkpoint", encoder=enc, decoder=dec, bbox_size=raw.get("bbox_size", 1025),
                     image_embed_encoding_size=raw.get("image_embed_encoding_size", 1024),
                     image_embed_encoding_multiplier=raw.get("image_embed_encoding_multiplier", 256),
                     num_register_tokens=raw.get("num_register_tokens", 4))


FEED_END = object()          # what a `generate(feed=...)` callable returns once no further lines will come


@dataclass
class RecognitionPrompt:
    id: int
    task_name: str
    image: np.ndarray
    text: Optional[str]
    math_mode: bool


class RecognitionPredictor(BasePredictor):
    model_loader_cls = RecognitionModelLoader
    batch_size = settings.RECOGNITION_BATCH_SIZE
    torch_dtype = None
    default_batch_sizes = {"cpu": 32, "mps": 64, "cuda": 256, "xla": 128}
    encoder_chunk_size: int = 4096
    encoder_chunk_sizes = {"cpu": 4096, "mps": 4096, "cuda": 32768, "xla": 32768}
    min_prefill_ratio: float = settings.RECOGNITION_MIN_PREFILL_RATIO
    min_trim_length: int = 50        # kept for API compatibility; the slot cache has no left padding to trim
    tasks = {
        TaskNames.ocr_with_boxes: {"needs_bboxes": True, "img_size": (1024, 256), "max_tokens": 224},
        TaskNames.ocr_without_boxes: {"needs_bboxes": False, "img_size": (1024, 256), "max_tokens": 224},
        TaskNames.block_without_boxes: {"needs_bboxes": False, "img_size": (1024, 512), "max_tokens": 768},
    }

    # Multi-GPU: when set, ONE call's lines are dealt over the ranks of `process_group` (default group if
    # None) and the results all-gathered; every rank must pass the same inputs (checked). Off by default.
    shard_lines: bool = settings.SURYA_AMD_SHARD
    process_group = None
    # Line crops are cut, padded, resized, normalised and patchified on the device (surya_rec_preprocess): the host keeps page
    # pixels as uint8 and describes lines by reference. RECOGNITION_PREPROCESS_HOST=1 selects the host (numpy) chain instead.
    device_preprocess: bool = not settings.RECOGNITION_PREPROCESS_HOST

    def __init__(self, checkpoint=None, device=None, dtype=None):
        super().__init__(checkpoint, device, dtype)
        self.prompt_queue = deque()
        self.batch_prompt_mapping = None
        self.preprocess_workers = min(8, os.cpu_count() or 1)

    # ------------------------------------------------------------------------------------------ bookkeeping
    def setup_cache(self, batch_size: int):
        self.prompt_queue.clear()
        self.batch_prompt_mapping = {i: None for i in range(batch_size)}

    @property
    def num_empty_slots(self):
        return sum(v is None for v in self.batch_prompt_mapping.values())

    @property
    def num_active_slots(self):
        return len(self.batch_prompt_mapping) - self.num_empty_slots

    # --------------------------------------------------------------------------------------------- slicing
    def _page(self, flat: dict, image) -> int:
        """Device path: register a page (uint8 HWC) once, return its index for LineRefs."""
        flat.setdefault("pages", []).append(page_pixels(image))
        return len(flat["pages"]) - 1

    def detect_and_slice_bboxes(self, images, task_names, det_predictor, detection_batch_size=None, highres_images=None):
        det_predictions = det_predictor(images, batch_size=detection_batch_size)
        flat = {"slices": [], "slice_map": [], "polygons": [], "task_names": [], "input_text": [], "res_scales": []}
        for det_pred, image, highres, task in zip(det_predictions, images, highres_images, task_names):
            polygons = [p.polygon for p in det_pred.bboxes]
            if highres:
                ws, hs = highres.size[0] / image.size[0], highres.size[1] / image.size[1]
                scaled = [[[int(p[0] * ws), int(p[1] * hs)] for p in poly] for poly in polygons]
                src, polys_px = highres, scaled
                scales = [(ws, hs)] * len(polygons)
            else:
                src, polys_px = image, polygons
                scales = [(1, 1)] * len(polygons)
            if self.device_preprocess:
                pg = self._page(flat, src)
                slices = [poly_ref(pg, src.size[0], src.size[1], poly) for poly in polys_px]
            else:
                slices = slice_polys_from_image(self.processor.image_processor(src), polys_px)
            flat["slice_map"].append(len(slices))
            flat["slices"].extend(slices)
            flat["polygons"].extend(polygons)
            flat["task_names"].extend([task] * len(slices))
            flat["res_scales"].extend(scales)
        flat["input_text"] = [None] * len(flat["slices"])
        return flat

    def slice_bboxes(self, images, task_names, bboxes=None, polygons=None, input_text=None) -> dict:
        assert bboxes is not None or polygons is not None
        flat = {"slices": [], "slice_map": [], "polygons": [], "task_names": [], "input_text": [], "res_scales": []}
        # ONE decision for the whole call (prepare_lines dispatches on the type of the first slice): the device path takes 4-point
        # polygons only, so a single polygon with another vertex count anywhere sends every image of the call down the host path
        dev = self.device_preprocess and (polygons is None or all(len(pl) == 4 for page in polygons for pl in page))
        for idx, image in enumerate(images):
            arr = None if dev else self.processor.image_processor(image)
            pg = self._page(flat, image) if dev else -1
            if polygons is not None:
                polys = polygons[idx]
                slices = ([poly_ref(pg, image.size[0], image.size[1], pl) for pl in polys] if dev
                          else slice_polys_from_image(arr, polys))
            else:
                slices = ([bbox_ref(pg, image.size[0], image.size[1], b) for b in bboxes[idx]] if dev
                          else slice_bboxes_from_image(arr, bboxes[idx]))
                polys = [[[b[0], b[1]], [b[2], b[1]], [b[2], b[3]], [b[0], b[3]]] for b in bboxes[idx]]
            flat["slice_map"].append(len(slices))
            flat["slices"].extend(slices)
            flat["polygons"].extend(polys)
            flat["task_names"].extend([task_names[idx]] * len(slices))
            flat["input_text"].extend([None] * len(slices) if input_text is None else input_text[idx])
        n = len(flat["slices"])
        assert n == sum(flat["slice_map"]) == len(flat["polygons"]) == len(flat["input_text"]) == len(flat["task_names"])
        flat["res_scales"] = [(1, 1)] * n
        return flat

    def prepare_input(self, task_names, images, input_text, math_modes):
        """Area clamp + prompt assembly per crop (reference :259-292)."""
        batch = []
        for image, text, task, math_mode in zip(images, input_text, task_names, math_modes):
            size = self.tasks[task]["img_size"]
            if image.size == 0:     # the reference catches cv2.error here and substitutes a blank crop (:272-278)
                image = np.zeros((size[1], size[0], 3), dtype=np.float32)
            else:
                image = self.processor.scale_to_fit(image, size)
            text = text or ""
            if len(text) > self.tasks[task]["max_tokens"]:
                text = ""
            batch.append({"task": task, "inputs": [{"type": "image", "image": image, "rotated": False},
                                                   {"type": "text", "text": text.strip(), "math": math_mode}]})
        return batch

    def preprocess_prompts_device(self, prompts: List[RecognitionPrompt], pages):
        """Device path: prompts whose `image` is a LineRef -> one surya_rec_preprocess call; prompt ids on the host."""
        if getattr(self, "_prep", None) is None:
            self._prep = DevicePreprocessor(self.model.device, self.processor.patch_size, self.processor.merge_size,
                                            settings.RECOGNITION_PAD_VALUE, self.processor.image_mean, self.processor.image_std)
        pages = list(pages)
        refs = []
        for p in prompts:
            ln = p.image
            if ln.size == 0:          # the reference substitutes a blank crop of the task size (:272-278): a black page of that size
                size = self.tasks[p.task_name]["img_size"]
                pages.append(np.zeros((size[1], size[0], 3), np.uint8))
                ln = LineRef(len(pages) - 1, 0, 0, size[0], size[1])
            refs.append(ln)
        tiles, offs, grids = self._prep(pages, refs, [self.tasks[p.task_name]["img_size"] for p in prompts])
        m2 = self.processor.merge_size ** 2
        ids = []
        for p, (gh, gw) in zip(prompts, grids):
            text = p.text or ""
            if len(text) > self.tasks[p.task_name]["max_tokens"]:
                text = ""
            ids.append(self.processor.prompt_ids(gh * gw // m2, p.task_name, text.strip(), p.math_mode, False))
        return tiles, offs, grids, ids

    def preprocess_prompts(self, prompts: List[RecognitionPrompt]):
        """All prompts -> (device tiles [sum P, 588], per-prompt tile offsets, grids, prompt ids)."""
        def one(p):
            b = self.prepare_input([p.task_name], [p.image], [p.text], [p.math_mode])
            return self.processor(b)

        if len(prompts) > 4 and self.preprocess_workers > 1:
            with ThreadPoolExecutor(self.preprocess_workers) as ex:
                outs = list(ex.map(one, prompts))
        else:
            outs = [one(p) for p in prompts]
        counts = [o["image_tiles"].shape[0] for o in outs]
        offs = np.zeros(len(outs) + 1, np.int64)
        offs[1:] = np.cumsum(counts)
        host = torch.from_numpy(np.concatenate([o["image_tiles"] for o in outs], 0))
        tiles = host.pin_memory().to(self.model.device, non_blocking=True) if host.numel() else host.to(self.model.device)
        grids = [tuple(int(x) for x in o["grid_hw"][0]) for o in outs]
        ids = [o["input_ids"][0] for o in outs]
        return tiles, offs, grids, ids

    # ------------------------------------------------------------------------------------------- hot loop
    def prepare_lines(self, flat: dict, math_mode: bool = True) -> dict:
        """Host half of the loop: build prompts, pre-process every crop, upload the tiles (they stay in HBM)."""
        prompts, max_tokens = [], {}
        for idx, (img, txt, task) in enumerate(zip(flat["slices"], flat["input_text"], flat["task_names"])):
            prompts.append(RecognitionPrompt(id=idx, task_name=task, text=txt, image=img, math_mode=math_mode))
            max_tokens[idx] = settings.RECOGNITION_MAX_TOKENS or self.tasks[task]["max_tokens"]
        if prompts and isinstance(prompts[0].image, LineRef):
            tiles, tile_offs, grids, prompt_ids = self.preprocess_prompts_device(prompts, flat.get("pages", []))
        else:
            tiles, tile_offs, grids, prompt_ids = self.preprocess_prompts(prompts)
        return {"prompts": prompts, "max_tokens": max_tokens, "tiles": tiles, "tile_offs": tile_offs, "grids": grids,
                "prompt_ids": prompt_ids}

    def generate(self, prep: dict | None, recognition_batch_size: int | None = None, on_done=None, on_flush=None, feed=None) -> tuple:
        """Device half: continuous batching over KV slots until every line stopped (reference :501-607).
        on_done(line, tokens, scores, bbox_rows[T, 6]) is called once per line, as soon as its stream is final; on_flush() after
        every host synchronisation point that finished at least one line (so a caller can hand the lines over in batches).

        `feed` (optional) makes the line list open-ended: `feed(block)` returns the next `prepare_lines` dict whose prompts carry
        the ids that continue the ones already admitted (queue order == id order), None when nothing is ready (only for block =
        False) or FEED_END. The loop polls it between decode calls and blocks on it only when it has nothing left to run, so lines
        can be admitted while their producer (the detector of a streamed detect -> recognise call) still works on later pages.
        Scheduling decisions never change a line's stream (slot / batch-composition invariance), so the result per line is the
        one the closed list gives. A streamed call states its token budget up front in the first dict (`overall_max_tokens`)."""
        prompts, grids, prompt_ids = [], [], []
        batch_max_tokens: dict = {}
        predicted_tokens, scores = [], []
        chunk_tiles, chunk_offs, chunk_base = [], [], []      # per admitted dict: its tile tensor, local tile offsets, first id
        line_chunk = np.zeros(0, np.int64)                    # id -> index into the three lists above
        if recognition_batch_size is None:
            recognition_batch_size = self.get_batch_size()
        recognition_batch_size = min(recognition_batch_size, self.model.max_slots)
        self.setup_cache(recognition_batch_size)
        if callable(getattr(self.model, "discard_ahead", None)):
            self.model.discard_ahead()                         # a previous loop that ended early (exception) must not poison this one
        first = prep if prep is not None else {}
        overall_max_tokens = int(first.get("overall_max_tokens") or max(first["max_tokens"].values()))
        batch_bboxes = np.zeros((0, overall_max_tokens, 6), np.float32)
        eos, pad, nop = self.processor.eos_token_id, self.processor.pad_token_id, self.processor.no_output_token
        steps_per_sync = max(1, min(settings.RECOGNITION_STEPS_PER_SYNC, 8))
        max_prefill = self.model.c.max_prefill_tokens
        # Token bookkeeping in array form: one row per line, written for all active slots of a step at once (the per-token Python
        # loop cost 1.2-1.6 us per token, a third of the device's own time per decode call, and fought the assembly thread for
        # the GIL). batch_prompt_mapping stays the slot table the scheduling decisions read; slot_line mirrors it as an array.
        cap = max(1, overall_max_tokens) + 1
        tok_mat = np.zeros((0, cap), np.int64)
        sc_mat = np.zeros((0, cap), np.float32)
        line_len = np.zeros(0, np.int64)
        max_tok = np.zeros(0, np.int64)
        slot_line = np.full(recognition_batch_size, -1, np.int64)
        REP = 40                                               # detect_repeat_token's window
        rep_cols = np.arange(-REP, 0)

        def admit(d):
            """Append the lines of one prepare_lines dict (ids continue the admitted ones)."""
            nonlocal batch_bboxes, tok_mat, sc_mat, line_len, max_tok, line_chunk
            new = d["prompts"]
            base, m = len(prompts), len(new)
            if m == 0:
                return
            assert [p.id for p in new] == list(range(base, base + m)), "fed prompts must continue the admitted ids"
            mt = np.asarray([d["max_tokens"][p.id] for p in new], np.int64)
            assert int(mt.max()) <= overall_max_tokens, "a fed line's token budget exceeds the call's overall_max_tokens"
            prompts.extend(new)
            grids.extend(d["grids"])
            prompt_ids.extend(d["prompt_ids"])
            batch_max_tokens.update(d["max_tokens"])
            predicted_tokens.extend([] for _ in range(m))
            scores.extend([] for _ in range(m))
            chunk_tiles.append(d["tiles"]); chunk_offs.append(d["tile_offs"]); chunk_base.append(base)
            line_chunk = np.concatenate([line_chunk, np.full(m, len(chunk_base) - 1, np.int64)])
            batch_bboxes = np.concatenate([batch_bboxes, np.zeros((m, overall_max_tokens, 6), np.float32)])
            tok_mat = np.concatenate([tok_mat, np.zeros((m, cap), np.int64)])
            sc_mat = np.concatenate([sc_mat, np.zeros((m, cap), np.float32)])
            line_len = np.concatenate([line_len, np.zeros(m, np.int64)])
            max_tok = np.concatenate([max_tok, mt])
            self.prompt_queue.extend(new)

        def tiles_of(first_id, last_id):
            """Tile rows of the consecutive lines first_id..last_id (all of one admitted dict)."""
            c = int(line_chunk[first_id])
            assert c == int(line_chunk[last_id])
            o, b0 = chunk_offs[c], chunk_base[c]
            return chunk_tiles[c][int(o[first_id - b0]): int(o[last_id - b0 + 1])]

        feed_done = feed is None

        def poll(block):
            """Admit what the feed has ready; with `block`, wait for the next dict (or the end)."""
            nonlocal feed_done
            got = False
            while not feed_done:
                nxt = feed(block and not got)
                if nxt is None:
                    break
                if nxt is FEED_END:
                    feed_done = True
                    break
                admit(nxt)
                got = got or bool(nxt["prompts"])

        if prep is not None:
            admit(prep)

        def finished(p_idx):
            L_ = int(line_len[p_idx])
            predicted_tokens[p_idx] = tok_mat[p_idx, :L_].tolist()
            scores[p_idx] = sc_mat[p_idx, :L_].tolist()
            if on_done is not None:
                on_done(p_idx, predicted_tokens[p_idx], scores[p_idx], batch_bboxes[p_idx, :max(min(L_, overall_max_tokens), 1)])

        def put(p, pos, t, s_, b_):
            """Token t / score s_ / box b_ of lines p at positions pos (arrays over the lines of one step)."""
            tok_mat[p, pos] = t
            sc_mat[p, pos] = s_
            m = pos < overall_max_tokens
            if m.all():
                batch_bboxes[p, pos] = b_
            elif m.any():
                batch_bboxes[p[m], pos[m]] = b_[m]
            line_len[p] = pos + 1

        def absorb(call):
            """Host half of one decode call: append its tokens, apply the stop rules (reference :583-595)."""
            k, ring = call
            tok, sc, bb = self.model.wait_outputs(k, ring)
            changed = False
            for step in range(k):
                s_idx = np.flatnonzero(slot_line >= 0)
                if s_idx.size == 0:
                    break
                p = slot_line[s_idx]
                pos = line_len[p]
                t = tok[step, s_idx]
                put(p, pos, t, sc[step, s_idx], bb[step, s_idx])
                new_len = pos + 1
                stop = (t == eos) | (t == pad) | (new_len >= max_tok[p])
                # repeat rule: <= 5 distinct ids in the last 40 and the last u ids equal to the u before; the distinct count is
                # screened in array form, only the few candidate lines run the exact rule
                c = np.flatnonzero(~stop & (new_len >= REP))
                if c.size:
                    win = np.sort(tok_mat[p[c, None], new_len[c, None] + rep_cols], axis=1)
                    few = c[(np.diff(win, axis=1) != 0).sum(axis=1) + 1 <= 5]
                    for ci in few.tolist():
                        if detect_repeat_token(tok_mat[p[ci], :new_len[ci]].tolist()):
                            stop[ci] = True
                if stop.any():
                    changed = True
                    for ci in np.flatnonzero(stop).tolist():
                        s_ = int(s_idx[ci])
                        slot_line[s_] = -1
                        self.batch_prompt_mapping[s_] = None
                        finished(int(p[ci]))
            if changed:
                self.model.set_active([k_ for k_, v in self.batch_prompt_mapping.items() if v is not None])
                if on_flush is not None:
                    on_flush()

        # Look-ahead encoding (RECOGNITION_ENCODE_AHEAD, default on): the vision encoder of the next up-to-batch-size queued
        # lines runs on the model's second stream while the current lines decode; prefill then only scatters the finished
        # embeddings and runs the decoder over the prompts. Scheduling decisions (which lines, which slots, when) are unchanged.
        look_ahead = settings.RECOGNITION_ENCODE_AHEAD
        ahead = deque()
        ahead_cap = max(self.model.c.max_prefill_tokens, self.model.c.max_slots)
        merge2 = self.model.cfg.encoder.spatial_merge_size ** 2

        def encode_ahead():
            cand, ntok_img = [], 0
            for p in self.prompt_queue:
                t = int(grids[p.id][0]) * int(grids[p.id][1]) // merge2
                if len(cand) >= recognition_batch_size or (cand and ntok_img + t > ahead_cap):
                    break
                if cand and line_chunk[p.id] != line_chunk[cand[0]]:
                    break                              # one tile tensor per encoder pass: the next admitted dict waits its turn
                cand.append(p.id)
                ntok_img += t
            self.model.encode_ahead(tiles_of(cand[0], cand[-1]), [grids[i] for i in cand])
            ahead.extend(cand)

        # The device runs one decode call ahead of the host: call n + 1 is enqueued before call n's tokens are looked at,
        # so the bookkeeping above overlaps with GPU work. A line that stops inside call n rides along in call n + 1
        # (its outputs are dropped: the slot is unmapped by then); new lines are admitted only with nothing in flight.
        inflight, ring = None, 0
        while True:
            if not feed_done:
                poll(block=not (self.prompt_queue or self.num_active_slots > 0 or inflight))
            if not (self.prompt_queue or self.num_active_slots > 0 or inflight):
                if feed_done:
                    break
                continue
            if (self.num_empty_slots / recognition_batch_size) > self.min_prefill_ratio and self.prompt_queue:
                if inflight:
                    absorb(inflight)
                    inflight = None
                    continue
                empty = [k for k, v in self.batch_prompt_mapping.items() if v is None]
                if look_ahead and not ahead:
                    encode_ahead()                     # nothing encoded yet (first batch): the prefill below waits for it
                take, ntok = [], 0
                while self.prompt_queue and len(take) < len(empty) and (not look_ahead or len(take) < len(ahead)):
                    L_ = len(prompt_ids[self.prompt_queue[0].id])
                    if take and (ntok + L_ > max_prefill or line_chunk[self.prompt_queue[0].id] != line_chunk[take[0].id]):
                        break
                    take.append(self.prompt_queue.popleft())
                    ntok += L_
                slots = empty[: len(take)]
                if look_ahead:
                    for p in take:
                        assert ahead.popleft() == p.id
                    self.model.prefill(None, [grids[p.id] for p in take], [prompt_ids[p.id] for p in take], slots)
                    if not ahead and self.prompt_queue:
                        encode_ahead()                 # the next lines' encoder pass runs beside the decode steps below
                else:
                    self.model.prefill(tiles_of(take[0].id, take[-1].id), [grids[p.id] for p in take],   # queue order == id order
                                       [prompt_ids[p.id] for p in take], slots)
                tok, sc, bb = self.model.read_outputs(1)
                ids_, sl_ = np.asarray([p.id for p in take], np.int64), np.asarray(slots, np.int64)
                first = tok[0, sl_]
                put(ids_, np.zeros(len(take), np.int64), first, sc[0, sl_], bb[0, sl_])
                for p_id, s, go in zip(ids_.tolist(), slots, ((first != eos) & (first != nop)).tolist()):
                    if go:                                                  # prefill stop rule (reference :559-563)
                        self.batch_prompt_mapping[s] = p_id
                        slot_line[s] = p_id
                    else:
                        finished(p_id)
                self.model.set_active([k for k, v in self.batch_prompt_mapping.items() if v is not None])
                if on_flush is not None:
                    on_flush()
            else:
                # steps some active line can still need once the call in flight is done (token budgets are known up front)
                act = slot_line[slot_line >= 0]
                budget = (int((max_tok[act] - line_len[act]).max()) if act.size else 0) - (inflight[0] if inflight else 0)
                if budget <= 0 and not inflight and self.num_active_slots > 0:
                    budget = 1                         # a line admitted with a one-token budget still gets its stop-rule step
                nxt = None
                if budget > 0:
                    nxt = (min(steps_per_sync, budget), ring)
                    self.model.decode_async(*nxt)
                    ring ^= 1
                if inflight:
                    absorb(inflight)
                inflight = nxt
        # the loop's own dense bookkeeping (ids [n, cap], scores [n, cap], lengths): what the sharded loop packs for its all_gather
        # without walking the per-line lists again
        from ..dist import PackedLines
        self.last_packed = (PackedLines(tok_mat, line_len), PackedLines(sc_mat, line_len))
        return predicted_tokens, torch.from_numpy(batch_bboxes), scores

    def prediction_loop(self, flat: dict, recognition_batch_size: int | None = None, math_mode: bool = True) -> tuple:
        return self.generate(self.prepare_lines(flat, math_mode), recognition_batch_size)

    def sharded_prediction_loop(self, flat: dict, recognition_batch_size: int | None = None, math_mode: bool = True) -> tuple:
        """One process per GPU (torch.distributed initialised by the caller, e.g. torchrun), used when `shard_lines` is set:
        every rank holds the same width-sorted line list (verified with a fingerprint all_gather), pre-processes and
        recognises only the lines dealt to it round-robin, and all ranks get all results back through ONE all_gather
        (surya_amd/dist.py). No collective touches the per-step data path. Single process: plain loop."""
        from .. import dist as sdist
        import zlib
        group = self.process_group
        rank, world = sdist.world_info(group)
        n = len(flat["slices"])
        if not sdist.collectives_on(group):                  # one rank (unless a forced 1-rank group, dist.force_collectives)
            return self.prediction_loop(flat, recognition_batch_size, math_mode)
        dev = sdist.collective_device(self.model.device, group)
        shapes = np.asarray([s.shape[:2] for s in flat["slices"]], np.int64).reshape(-1, 2)
        if n and isinstance(flat["slices"][0], LineRef):      # device path: lines are references into the uploaded pages
            probe = b"".join(np.asarray([(l.page, l.x0, l.y0, l.x1, l.y1) for l in flat["slices"]], np.int64).tobytes()
                             for _ in (0,)) + b"".join(pg.reshape(-1)[:4096].tobytes() for pg in flat.get("pages", [])[:16])
        else:
            probe = b"".join(np.ascontiguousarray(flat["slices"][i]).tobytes()[:4096] for i in range(0, n, max(1, n // 16)))
        sdist.assert_same_inputs([n, zlib.crc32(shapes.tobytes()), zlib.crc32(probe)], group, dev)
        if n == 0:
            return self.prediction_loop(flat, recognition_batch_size, math_mode)
        mine = sdist.shard_indices(n, world, rank)
        local = {k: [flat[k][i] for i in mine] for k in ("slices", "input_text", "task_names")}
        if "pages" in flat:
            local["pages"] = flat["pages"]
        max_tokens = max(settings.RECOGNITION_MAX_TOKENS or self.tasks[t]["max_tokens"] for t in flat["task_names"])
        if mine:
            self.last_packed = None                          # only what generate() sets DURING this call counts (a stand-in prediction_loop
            toks, boxes, scores = self.prediction_loop(local, recognition_batch_size, math_mode)      # must not gather an earlier call's arrays)
            packed = getattr(self, "last_packed", None)
            if packed is not None and len(packed[0]) == len(toks):      # generate()'s dense arrays: no per-token Python in the pack
                toks, scores = packed
            self.last_packed = None                          # (and the [n, cap] matrices are not pinned on the predictor between calls)
            boxes = boxes.numpy()
            if boxes.shape[1] < max_tokens:
                boxes = np.pad(boxes, ((0, 0), (0, max_tokens - boxes.shape[1]), (0, 0)))
        else:
            toks, scores, boxes = [], [], np.zeros((0, max_tokens, 6), np.float32)
        toks, scores, boxes = sdist.gather_line_outputs(toks, scores, boxes, mine, n, max_tokens, device=dev, group=group)
        return toks, torch.from_numpy(boxes), scores

    # ------------------------------------------------------------------------------------- output assembly
    def get_bboxes_text(self, flat, predicted_tokens, scores, predicted_polygons, drop_repeated_text=False) -> list:
        """Token stream -> per line (texts, confidences, bbox_valid, polygons [n, 4, 2]) (reference :609-771): the stream is cut
        into runs of math-BPE ids, single special tags and UTF-16 ids; only the last kind carries per-character boxes.
        Array form of the reference's per-token loop: run boundaries, close-polygon filtering and the
        char -> box index map are numpy expressions per line; Python only walks the (few) runs of a line. Lines come back as
        None (<NOP>), or a tuple that `_chars_of` turns into TextChars after the geometry has been applied in bulk."""
        eos, pad, nop = self.processor.eos_token_id, self.processor.pad_token_id, self.processor.no_output_token
        blank = np.array([[0, 0], [0, 1], [1, 1], [1, 0]], np.float64)
        out = []
        for tokens, polys, sc in zip(predicted_tokens, predicted_polygons, scores):
            if nop in tokens:
                out.append(None)
                continue
            if drop_repeated_text and detect_repeat_token(tokens):
                out.append(([""], np.zeros(1), np.zeros(1, bool), blank[None].copy()))
                continue
            tid = np.asarray(tokens, np.int64)
            stop = np.nonzero((tid == eos) | (tid == pad))[0]
            n = int(stop[0]) if len(stop) else len(tid)
            n = min(n, len(polys), len(sc))              # zip() of the reference stops at the shortest of the three
            if n == 0:
                out.append(([], np.zeros(0), np.zeros(0, bool), np.zeros((0, 4, 2))))
                continue
            tid = tid[:n]
            P = np.asarray(polys[:n], np.float64)
            conf = np.asarray(sc[:n], np.float64)
            # clean_close_polygons: a box is dropped when all 4 corners sit within 0.1 of the PREVIOUS box of its run (util.py:100-120)
            far = (np.abs(P[1:] - P[:-1]).reshape(n - 1, 8).max(axis=1) > 0.1).tolist() if n > 1 else []
            texts, src, csrc, valid = self._line_runs(tid.tolist(), far)
            if not texts:
                out.append(([], np.zeros(0), np.zeros(0, bool), np.zeros((0, 4, 2))))
            else:
                v = np.asarray(valid, bool)
                pp = P[src]
                pp[~v] = blank
                out.append((texts, conf[csrc], v, pp))
        return out

    def _line_runs(self, ids: list, far: list):
        """The runs of one token stream (already cut at eos / pad): per output char its text, the token whose BOX it takes, the
        token whose CONFIDENCE it takes, bbox_valid. `far[j]`: box j + 1 differs from box j by more than 0.1 in some corner. The
        reference indexes a run's unfiltered confidences with the index into its FILTERED boxes (:700-712): kept as is."""
        tk = self.processor.ocr_tokenizer
        q_off, s_off = tk.qwen_offset, tk.special_token_offset
        n = len(ids)
        texts, src, csrc, valid = [], [], [], []
        if n and min(ids) >= s_off:
            kind = None                                  # one UTF-16 run (the usual line of text)
        else:
            kind = [0 if t < q_off else (1 if t < s_off else 2) for t in ids]
        a_ = 0
        while a_ < n:
            if kind is None:
                k, b_ = 2, n
            else:
                k = kind[a_]
                b_ = a_ + 1
                if k != 1:
                    while b_ < n and kind[b_] == k:
                        b_ += 1
            if k == 2:
                # a run of UTF-16 code units decodes in one piece (tokenizer._decode_ocr's flush of a non-math buffer)
                # (ids above the tokenizer's range -- a checkpoint with a padded lm_head -- wrap into 16 bits like the byte masking
                # of tokenizer._decode_ocr instead of raising OverflowError)
                text = array("H", [(t - s_off) & 0xFFFF for t in ids[a_:b_]]).tobytes().decode("utf-16le", errors="ignore")
                if text:
                    boxes = [a_] + [j for j in range(a_ + 1, b_) if far[j - 1]]
                    L, nb = len(text), len(boxes)
                    texts.extend(text)
                    src.extend(boxes[:L] if L <= nb else boxes + [boxes[-1]] * (L - nb))      # char i -> box min(i, nb - 1)
                    csrc.extend(range(a_, a_ + L) if L <= nb else list(range(a_, a_ + nb)) + [a_ + nb - 1] * (L - nb))
                    valid.extend([True] * L)
            else:
                text = tk.decode(ids[a_:b_], task=TaskNames.ocr_without_boxes if k == 1 else TaskNames.block_without_boxes)
                if not (k == 1 and (text == NOMATH_TOKEN or _SCRIPT_TAG.match(text))):
                    texts.append(text); src.append(a_); csrc.append(a_); valid.append(False)
            a_ = b_
        return texts, src, csrc, valid

    @staticmethod
    def _chars_of(line, res_scale, line_bbox) -> List[TextChar]:
        """TextChars of one line with the reference's per-char geometry (:905-909: rescale by the high-res factor with int()
        truncation, shift to the line's corner, clamp into the line's bbox) applied to all of the line's polygons at once;
        objects are built without re-validating fields that were just computed (pydantic model_construct)."""
        texts, conf, valid, P = line
        if not texts:
            return []
        P = P.copy()
        P[..., 0] = np.trunc(P[..., 0] * (1.0 / res_scale[0])) + line_bbox[0]
        P[..., 1] = np.trunc(P[..., 1] * (1.0 / res_scale[1])) + line_bbox[1]
        np.clip(P[..., 0], line_bbox[0], line_bbox[2], out=P[..., 0])
        np.clip(P[..., 1], line_bbox[1], line_bbox[3], out=P[..., 1])
        polys = P.tolist()
        conf = [0.0 if c != c else c for c in conf.tolist()]               # BaseChar: NaN -> 0, stored as a float
        return [_text_char(pg, c, t, v) for pg, c, t, v in zip(polys, conf, texts, valid.tolist())]

    def _assemble_line(self, flat, sorted_pos, orig, tokens, sc, bbox_rows, drop_repeated_text, return_words, bbox_size) -> TextLine:
        """One line's TextLine from its finished token stream (reference :609-771 + :886-925)."""
        polygon, res_scale = flat["polygons"][orig], flat["res_scales"][orig]
        polys = prediction_to_polygon_batch(bbox_rows[None], [flat["slices"][sorted_pos].shape], bbox_size, bbox_size // 2)
        chars = self.get_bboxes_text(flat, [tokens], [sc], polys, drop_repeated_text)[0]
        if chars is None or not chars[0]:      # <NOP> (input text was good) or nothing decoded (reference :889-899)
            return TextLine(text="", polygon=polygon, chars=[], confidence=1, original_text_good=True)
        # mean of the characters' confidences as the TextChar objects hold them (NaN -> 0, schema.py), reference :899-903
        confidence = float(np.mean(np.where(np.isnan(chars[1]), 0.0, chars[1])))
        box = PolygonBox(polygon=polygon)
        chars = self._chars_of(chars, res_scale, box.bbox)
        chars = fix_unbalanced_tags(chars, self.processor.ocr_tokenizer.special_tokens)
        text = clean_math_tags(unwrap_math("".join(c.text for c in chars)))
        return TextLine(text=text, polygon=polygon, chars=chars, confidence=confidence,
                        words=words_from_chars(chars, box) if return_words else [])

    def _assemble_batch(self, flat, items, drop_repeated_text, return_words, bbox_size) -> List[TextLine]:
        """TextLines of several finished lines at once; items = [(sorted_pos, orig, tokens, scores, bbox_rows[T, 6])]. The same
        result as `_assemble_line` per item (tests/test_assemble_cpu.py compares the two), but the numpy work -- box tokens ->
        polygons, close-box filter, per-char rescale / shift / clamp -- is done ONCE for the whole batch instead of ~25 small
        array calls per line, and TextLine is built from values that are already in validated form. Python walks only the token
        runs (`_line_runs`) and creates the character objects. ~430 -> 80-90 us per 45-character line (tools/hostbench/assemble_cost.py)."""
        eos, pad, nop = self.processor.eos_token_id, self.processor.pad_token_id, self.processor.no_output_token
        out: List[Optional[TextLine]] = [None] * len(items)
        work, t_max = [], 0
        for i, (sp, orig, tokens, sc, rows) in enumerate(items):
            if nop in tokens or (drop_repeated_text and detect_repeat_token(tokens)):
                out[i] = self._assemble_line(flat, sp, orig, tokens, sc, rows, drop_repeated_text, return_words, bbox_size)
                continue
            n = len(tokens)
            for j, t in enumerate(tokens):
                if t == eos or t == pad:
                    n = j
                    break
            n = min(n, len(rows), len(sc))
            if n == 0:
                out[i] = TextLine(text="", polygon=flat["polygons"][orig], chars=[], confidence=1, original_text_good=True)
                continue
            work.append((i, n))
            t_max = max(t_max, n)
        if not work:
            return out
        W = len(work)
        R = np.zeros((W, t_max, 6), np.float32)
        for w, (i, n) in enumerate(work):
            R[w, :n] = items[i][4][:n]
        P = prediction_to_polygon_batch(R, [flat["slices"][items[i][0]].shape for i, _ in work], bbox_size,
                                        bbox_size // 2).astype(np.float64)                       # [W, t_max, 4, 2]
        far = (np.abs(P[:, 1:] - P[:, :-1]).reshape(W, t_max - 1, 8).max(axis=2) > 0.1).tolist() if t_max > 1 else [[]] * W
        keep, all_w, all_src, all_conf, all_valid, counts, geo = [], [], [], [], [], [], []
        for w, (i, n) in enumerate(work):
            sp, orig, tokens, sc, rows = items[i]
            texts, src, csrc, valid = self._line_runs(tokens[:n], far[w])
            if not texts:                      # nothing decoded (reference :889-899)
                out[i] = TextLine(text="", polygon=flat["polygons"][orig], chars=[], confidence=1, original_text_good=True)
                continue
            polygon = coerce_polygon(flat["polygons"][orig])
            xs, ys = [p[0] for p in polygon], [p[1] for p in polygon]
            bbox = [min(xs), min(ys), max(xs), max(ys)]
            rs = flat["res_scales"][orig]
            keep.append((i, texts, valid, polygon, bbox))
            all_w.extend([w] * len(src)); all_src.extend(src); all_valid.extend(valid)
            all_conf.extend([0.0 if sc[j] != sc[j] else sc[j] for j in csrc])          # TextChar's NaN -> 0 rule
            counts.append(len(src))
            geo.append((1.0 / rs[0], 1.0 / rs[1], bbox[0], bbox[1], bbox[2], bbox[3]))
        if not keep:
            return out
        PP = P[all_w, all_src]                                                                    # [C, 4, 2]
        v_all = np.asarray(all_valid, bool)
        PP[~v_all] = _BLANK_POLY
        g = np.repeat(np.asarray(geo, np.float64), counts, axis=0)[:, :, None]                    # [C, 6, 1]
        PP[..., 0] = np.minimum(np.maximum(np.trunc(PP[..., 0] * g[:, 0]) + g[:, 2], g[:, 2]), g[:, 4])
        PP[..., 1] = np.minimum(np.maximum(np.trunc(PP[..., 1] * g[:, 1]) + g[:, 3], g[:, 3]), g[:, 5])
        polys = PP.tolist()
        conf_arr = np.asarray(all_conf, np.float64)
        special = self.processor.ocr_tokenizer.special_tokens
        a = 0
        for (i, texts, valid, polygon, bbox), c in zip(keep, counts):
            b = a + c
            confidence = float(np.mean(conf_arr[a:b]))
            chars = [_text_char(pg, cf, t, v) for pg, cf, t, v in zip(polys[a:b], all_conf[a:b], texts, valid)]
            a = b
            if not all(valid):                                   # tags only come from special / math runs (bbox_valid False)
                chars = fix_unbalanced_tags(chars, special)
                text = "".join(ch.text for ch in chars)
            else:
                text = "".join(texts)
            if "<" in text:
                text = clean_math_tags(unwrap_math(text))
            words = words_from_chars(chars, PolygonBox(polygon=polygon)) if return_words else []
            out[i] = _text_line(polygon, confidence, text, chars, words)
        return out

    def __call__(self, images: List[Image.Image], task_names: List[str] | None = None, det_predictor=None,
                 detection_batch_size: int | None = None, recognition_batch_size: int | None = None,
                 highres_images: List[Image.Image] | None = None, bboxes: List[List[List[int]]] | None = None,
                 polygons: List[List[List[List[int]]]] | None = None, input_text: List[List[str | None]] | None = None,
                 sort_lines: bool = False, math_mode: bool = True, return_words: bool = False,
                 drop_repeated_text: bool = False) -> List[OCRResult]:
        # the whole call runs with the cyclic GC paused (gc_paused): slicing, scheduling and assembly allocate ~10^6 acyclic objects
        with gc_paused():
            return self._call(images, task_names, det_predictor, detection_batch_size, recognition_batch_size, highres_images,
                              bboxes, polygons, input_text, sort_lines, math_mode, return_words, drop_repeated_text)

    def _call(self, images, task_names, det_predictor, detection_batch_size, recognition_batch_size, highres_images, bboxes,
              polygons, input_text, sort_lines, math_mode, return_words, drop_repeated_text) -> List[OCRResult]:
        if getattr(self, "_poisoned", None):
            raise RuntimeError(self._poisoned)
        allowed = self.tasks.keys()
        t_call = time.perf_counter()
        stamps = self.last_timing = {}                    # wall-clock phases of this call in ms (bench.py's e2e leg reports them)
        if task_names is None:
            task_names = [TaskNames.ocr_with_boxes] * len(images)
        assert all(t in allowed for t in task_names), (
            f"One or more tasks in {task_names} is not supported. Supported tasks are {allowed}")
        assert len(images) == len(task_names), "You need to pass in one task name for each image"
        images = convert_if_not_rgb(images)
        if highres_images is not None:
            assert len(images) == len(highres_images), "You need to pass in one highres image for each image"
        highres_images = convert_if_not_rgb(highres_images) if highres_images is not None else [None] * len(images)

        if bboxes is None and polygons is None:
            assert det_predictor is not None, (
                "You need to pass in a detection predictor if you don't provide bboxes or polygons")
            if self.shard_pages:
                from .. import dist as sdist
                if sdist.collectives_on(self.process_group):
                    return self._call_page_sharded(images, task_names, det_predictor, detection_batch_size, recognition_batch_size,
                                                   highres_images, sort_lines, math_mode, return_words, drop_repeated_text)
            if self._can_stream(det_predictor):
                return self._call_streamed(images, task_names, det_predictor, detection_batch_size, recognition_batch_size,
                                           highres_images, sort_lines, math_mode, return_words, drop_repeated_text, stamps, t_call)
            flat = self.detect_and_slice_bboxes(images, task_names, det_predictor, detection_batch_size, highres_images)
        else:
            if bboxes is not None:
                assert len(images) == len(bboxes), "You need to pass in one list of bboxes for each image"
            if polygons is not None:
                assert len(images) == len(polygons), "You need to pass in one list of polygons for each image"
            flat = self.slice_bboxes(images, bboxes=bboxes, polygons=polygons, input_text=input_text, task_names=task_names)
        if len(flat["slices"]) == 0:
            return []
        stamps["slice_ms"] = (time.perf_counter() - t_call) * 1e3

        # widest first: the length bucketing that keeps prefill batches homogeneous (reference :847-854)
        order = sorted(range(len(flat["slices"])), key=lambda i: -flat["slices"][i].shape[1])
        for key in ("slices", "input_text", "task_names"):
            flat[key] = [flat[key][i] for i in order]

        # original position of every sorted line, so a finished line can be assembled against its own polygon / scale
        orig_of = order
        bbox_size = self.model.cfg.bbox_size

        def assemble(batch):
            # batch = [(sorted_pos, tokens, scores, bbox_rows)]: the lines that stopped at one synchronisation point
            return self._assemble_batch(flat, [(k, orig_of[k], t, sc, bb) for k, t, sc, bb in batch], drop_repeated_text,
                                        return_words, bbox_size)

        text_lines = [None] * len(order)                      # by ORIGINAL position
        if self.shard_lines:
            predicted_tokens, batch_bboxes, scores = self.sharded_prediction_loop(flat, recognition_batch_size, math_mode)
            bb = batch_bboxes.numpy()
            for a in range(0, len(order), 256):
                ks = range(a, min(a + 256, len(order)))
                for k, line in zip(ks, assemble([(k, predicted_tokens[k], scores[k], bb[k]) for k in ks])):
                    text_lines[orig_of[k]] = line
        else:
            # Output assembly is host work of the same order as the device loop itself; it runs on one worker thread WHILE the
            # device decodes the next lines: the lines that stopped at a synchronisation point are handed over together as soon
            # as it is over (batched numpy work, `_assemble_batch`). The scheduler thread spends most of its time blocked in
            # hipEventSynchronize (GIL released), which is when the worker runs.
            futures, pending = [], []
            with ThreadPoolExecutor(1) as pool:
                def on_done(k, tokens, sc, bbox_rows):
                    pending.append((k, list(tokens), list(sc), bbox_rows.copy()))

                def on_flush():
                    if pending:
                        batch = pending[:]
                        pending.clear()
                        futures.append(([b[0] for b in batch], pool.submit(assemble, batch)))
                t0 = time.perf_counter()
                prep = self.prepare_lines(flat, math_mode)
                t1 = time.perf_counter()
                self.generate(prep, recognition_batch_size, on_done=on_done, on_flush=on_flush)
                on_flush()
                t2 = time.perf_counter()
                for ks, f in futures:
                    for k, line in zip(ks, f.result()):
                        text_lines[orig_of[k]] = line
                stamps.update(prepare_ms=(t1 - t0) * 1e3, device_loop_ms=(t2 - t1) * 1e3,
                              assemble_tail_ms=(time.perf_counter() - t2) * 1e3)
            assert all(t is not None for t in text_lines)

        results, start = [], 0
        for idx, image in enumerate(images):
            end = start + flat["slice_map"][idx]
            lines = text_lines[start:end]
            start = end
            if sort_lines:
                lines = sort_text_lines(lines)
            results.append(OCRResult(text_lines=lines, image_bbox=[0, 0, image.size[0], image.size[1]]))
        stamps["total_ms"] = (time.perf_counter() - t_call) * 1e3
        return results

    # ---------------------------------------------------------------------- N > 1: whole pages per rank
    # `shard_lines` deals the LINES of every page over the ranks: every rank detects its share of pages, but then slices, sorts and
    # -- after the all_gather of the token outputs -- assembles ALL lines of ALL pages, so the host work of a call does not shrink with
    # the rank count (128 pages over 8 ranks: ~355 lines of device work per rank against ~2800 lines of assembly on each; VERDICT r04
    # weak #7b), and the streamed detect -> recognise schedule is off. `shard_pages` deals whole PAGES instead: rank r takes pages
    # r, r + world, ... and runs the complete single-rank call on them -- streamed, its own detector batches feeding its own
    # continuous-batching loop, its own assembly -- with no collective on the data path at all. One gather at the end:
    #   gather_page_results = True   every rank gets every page's OCRResult (the drop-in contract; pickled result objects, host cost
    #                                grows with the page count)
    #   gather_page_results = False  results stay partitioned: a rank returns OCRResults for its own pages and None elsewhere; the
    #                                gather carries one small record per page (line count, character count, CRC of the text) so every
    #                                rank can check that each page was processed exactly once (`last_page_summary`)
    shard_pages: bool = False
    gather_page_results: bool = True

    def _call_page_sharded(self, images, task_names, det_predictor, detection_batch_size, recognition_batch_size, highres_images,
                           sort_lines, math_mode, return_words, drop_repeated_text) -> list:
        from .. import dist as sdist
        import zlib
        group = self.process_group
        rank, world = sdist.world_info(group)
        n = len(images)
        dev = sdist.collective_device(self.model.device, group)
        sizes = np.asarray([im.size for im in images], np.int64).reshape(-1, 2)
        # (a 64 x 16 pixel corner of up to 16 pages: `tobytes()` of whole pages copied several MB per page on every rank, inside the call)
        probe = b"".join(images[i].crop((0, 0, min(images[i].size[0], 64), min(images[i].size[1], 16))).tobytes()
                         for i in range(0, n, max(1, n // 16))) if n else b""
        sdist.assert_same_inputs([n, zlib.crc32(sizes.tobytes()), zlib.crc32(probe)], group, dev)
        mine = sdist.shard_indices(n, world, rank)
        saved = (self.shard_pages, self.shard_lines, getattr(det_predictor, "shard_pages", False))
        self.shard_pages = self.shard_lines = False
        if hasattr(det_predictor, "shard_pages"):
            det_predictor.shard_pages = False
        try:
            hr = [highres_images[i] for i in mine]
            local = self._call([images[i] for i in mine], [task_names[i] for i in mine], det_predictor, detection_batch_size,
                               recognition_batch_size, hr if any(h is not None for h in hr) else None,
                               None, None, None, sort_lines, math_mode, return_words, drop_repeated_text) if mine else []
        finally:
            self.shard_pages, self.shard_lines = saved[0], saved[1]
            if hasattr(det_predictor, "shard_pages"):
                det_predictor.shard_pages = saved[2]
        if len(local) != len(mine):
            # the single-rank call returns [] when its pages hold no line at all (reference contract, :928-929); this rank's pages then
            # exist all the same: one empty OCRResult per page, so the gather below has an entry for every page of the whole call
            # (a rank dealt only blank pages made those pages None on every rank / tripped the processed-once assert)
            assert len(local) == 0, (len(local), len(mine))
            local = [OCRResult(text_lines=[], image_bbox=[0, 0, images[i].size[0], images[i].size[1]]) for i in mine]
        if self.gather_page_results:
            out = sdist.gather_objects(local, mine, n, group)
            # ... and the call as a whole keeps the single-rank contract: no line anywhere -> []
            return out if any(r is not None and len(r.text_lines) for r in out) else []
        rec = [(len(r.text_lines), sum(len(l.text) for l in r.text_lines), zlib.crc32("\n".join(l.text for l in r.text_lines).encode()))
               for r in local]
        self.last_page_summary = sdist.gather_objects(rec, mine, n, group)
        assert all(x is not None for x in self.last_page_summary), "a page was processed by no rank"
        if not any(x[0] for x in self.last_page_summary):
            return []                                         # no line on any page of the call: the single-rank contract
        out = [None] * n
        for i, r in zip(mine, local):
            out[i] = r
        return out

    # ---------------------------------------------------------------------- streamed detect -> recognise
    # One call, two threads: a producer runs the detector batch by batch (DetectionPredictor.iter_detect) and turns each
    # batch's boxes into a prepared chunk of lines (LineRefs, width sort within the chunk, device pre-processing, prompt ids);
    # the calling thread runs the continuous-batching loop and admits a chunk as soon as it exists, so the recogniser decodes the
    # first pages' lines while the detector still works on the later pages. Both threads launch on the same HIP stream (device
    # work is ordered by submission, no cross-stream hand-off), each waits only on its own events. The reference detects every
    # page before the first crop is recognised (recognition/__init__.py:371-399, 836-845); the order of the work does not enter
    # any result: a line's tokens do not depend on its slot or batch mates, and results are re-assembled by original position.
    stream_detection: bool = settings.RECOGNITION_STREAM_DETECTION

    def _can_stream(self, det_predictor) -> bool:
        """This package's detector on its device post-processing path, one process, and a `__call__` nobody overrode (a subclass that
        filters or edits the results in `__call__` must keep seeing every page before the first crop is cut: serial path)."""
        from ..detection.predictor import DetectionPredictor
        return (self.stream_detection and self.device_preprocess and not self.shard_lines
                and isinstance(det_predictor, DetectionPredictor)
                and type(det_predictor).__call__ is DetectionPredictor.__call__ and type(det_predictor)._call is DetectionPredictor._call
                and type(det_predictor).iter_detect is DetectionPredictor.iter_detect
                # ... nor the pieces iter_detect is made of: a subclass that customises them for __call__ would be silently bypassed
                and type(det_predictor)._detect is DetectionPredictor._detect
                and type(det_predictor)._detect_device is DetectionPredictor._detect_device
                and type(det_predictor)._iter_detect_device is DetectionPredictor._iter_detect_device
                and det_predictor.device_postprocess and not det_predictor.shard_pages)

    def _call_streamed(self, images, task_names, det_predictor, detection_batch_size, recognition_batch_size, highres_images,
                       sort_lines, math_mode, return_words, drop_repeated_text, stamps, t_call) -> List[OCRResult]:
        import queue
        import sys
        import threading
        # by line id (admission order): slices / task_names / input_text; by ORIGINAL position (page order): the rest
        flat = {"slices": [], "slice_map": [], "polygons": [], "task_names": [], "input_text": [], "res_scales": []}
        orig_of: List[int] = []                               # line id -> original position
        q: "queue.Queue" = queue.Queue()
        overall_max_tokens = max([settings.RECOGNITION_MAX_TOKENS or self.tasks[t]["max_tokens"] for t in task_names] or [1])
        device = self.model.device
        det_wall = [0.0]
        stop = threading.Event()

        def produce():
            try:
                if torch.cuda.is_available():
                    torch.cuda.set_device(device)             # the current device is per thread
                page, t_p = 0, time.perf_counter()
                for dets in det_predictor.iter_detect(images, batch_size=detection_batch_size):
                    if stop.is_set():
                        break
                    pages, refs, polys_all, scales_all, tasks_all = [], [], [], [], []
                    for det_pred in dets:
                        image, highres, task = images[page], highres_images[page], task_names[page]
                        polygons = [b.polygon for b in det_pred.bboxes]
                        if highres:
                            ws, hs = highres.size[0] / image.size[0], highres.size[1] / image.size[1]
                            src = highres
                            polys_px = [[[int(pt[0] * ws), int(pt[1] * hs)] for pt in poly] for poly in polygons]
                            scale = (ws, hs)
                        else:
                            src, polys_px, scale = image, polygons, (1, 1)
                        pages.append(page_pixels(src))
                        refs.extend(poly_ref(len(pages) - 1, src.size[0], src.size[1], poly) for poly in polys_px)
                        polys_all.extend(polygons)
                        scales_all.extend([scale] * len(polygons))
                        tasks_all.extend([task] * len(polygons))
                        flat["slice_map"].append(len(polygons))
                        page += 1
                    if not refs:
                        continue
                    base_orig, base_id = len(flat["polygons"]), len(flat["slices"])
                    # widest first inside the chunk: the length bucketing of the reference's global sort (:847-854), per arrival
                    order = sorted(range(len(refs)), key=lambda i: -refs[i].shape[1])
                    prompts = [RecognitionPrompt(id=base_id + k, task_name=tasks_all[i], text=None, image=refs[i], math_mode=math_mode)
                               for k, i in enumerate(order)]
                    tiles, tile_offs, grids, prompt_ids = self.preprocess_prompts_device(prompts, pages)
                    # everything the assembly thread reads about these lines is in place BEFORE the chunk can be admitted
                    flat["polygons"].extend(polys_all)
                    flat["res_scales"].extend(scales_all)
                    flat["slices"].extend(refs[i] for i in order)
                    flat["task_names"].extend(tasks_all[i] for i in order)
                    flat["input_text"].extend([None] * len(order))
                    orig_of.extend(base_orig + i for i in order)
                    q.put({"prompts": prompts, "tiles": tiles, "tile_offs": tile_offs, "grids": grids, "prompt_ids": prompt_ids,
                           "max_tokens": {p.id: settings.RECOGNITION_MAX_TOKENS or self.tasks[p.task_name]["max_tokens"]
                                          for p in prompts}})
                det_wall[0] = (time.perf_counter() - t_p) * 1e3
                q.put(FEED_END)
            except BaseException as e:                        # surfaces in the calling thread
                q.put(e)

        def feed(block):
            try:
                item = q.get(block)
            except queue.Empty:
                return None
            if isinstance(item, BaseException):
                raise item
            return item

        bbox_size = self.model.cfg.bbox_size

        def assemble(batch):
            return self._assemble_batch(flat, [(k, orig_of[k], t, sc, bb) for k, t, sc, bb in batch], drop_repeated_text,
                                        return_words, bbox_size)

        futures, pending = [], []
        # the producer's Python work (box lists, line descriptors) must not keep the scheduler from its next launch for a
        # whole default switch interval (5 ms = a decode call of 4 steps)
        old_switch = sys.getswitchinterval()
        sys.setswitchinterval(min(old_switch, 5e-4))
        if os.environ.get("SURYA_AMD_PROFILE_PRODUCER"):     # debugging aid: cProfile of the detector thread's Python work, to stderr
            import cProfile, pstats
            _inner = produce

            def produce():                                   # noqa: F811
                pr = cProfile.Profile()
                pr.enable()
                try:
                    _inner()
                finally:
                    pr.disable()
                    pstats.Stats(pr, stream=sys.stderr).sort_stats("tottime").print_stats(22)
        producer = threading.Thread(target=produce, name="surya-amd-detect", daemon=True)
        try:
            with ThreadPoolExecutor(1) as pool:
                def on_done(k, tokens, sc, bbox_rows):
                    pending.append((k, list(tokens), list(sc), bbox_rows.copy()))

                def on_flush():
                    if pending:
                        batch = pending[:]
                        pending.clear()
                        futures.append(([b[0] for b in batch], pool.submit(assemble, batch)))
                t0 = time.perf_counter()
                producer.start()
                try:
                    self.generate({"prompts": [], "max_tokens": {}, "overall_max_tokens": overall_max_tokens}, recognition_batch_size,
                                  on_done=on_done, on_flush=on_flush, feed=feed)
                except BaseException:
                    stop.set()                         # the producer ends after the batch it is working on ...
                    producer.join(timeout=30.0)        # ... and is waited for: it launches detector and pre-processing work on this predictor's
                    if producer.is_alive():            # objects and stream, and a retry of the call must not run beside it
                        self._poisoned = "a streamed call failed and its detector thread did not stop within 30 s: create a new predictor"
                    raise
                producer.join()                        # it has put FEED_END: nothing left to run
                on_flush()
                t2 = time.perf_counter()
                n = len(orig_of)
                text_lines = [None] * n
                for ks, f in futures:
                    for k, line in zip(ks, f.result()):
                        text_lines[orig_of[k]] = line
                stamps.update(streamed=1.0, detect_thread_ms=det_wall[0], device_loop_ms=(t2 - t0) * 1e3,
                              assemble_tail_ms=(time.perf_counter() - t2) * 1e3)
        finally:
            sys.setswitchinterval(old_switch)
        if n == 0:
            return []
        assert all(t is not None for t in text_lines) and len(flat["slice_map"]) == len(images)
        results, start = [], 0
        for idx, image in enumerate(images):
            end = start + flat["slice_map"][idx]
            lines = text_lines[start:end]
            start = end
            if sort_lines:
                lines = sort_text_lines(lines)
            results.append(OCRResult(text_lines=lines, image_bbox=[0, 0, image.size[0], image.size[1]]))
        stamps["total_ms"] = (time.perf_counter() - t_call) * 1e3
        return results
