"""LayoutPredictor: drop-in for surya.layout.LayoutPredictor (surya/layout/__init__.py:18-225) on the HIP layout model.

Same call signature, batching by slice count, greedy box loop with the reference's stop and re-labelling rules, same LayoutResult
schema. The model calls are `HipLayoutModel.encode` (once per batch) and `.decode_step` (once per emitted box; the reference
synchronises with the host after every step as well). There is no CPU fallback for the model."""
from __future__ import annotations

import os
from typing import List, Optional

import numpy as np
import torch
from PIL import Image

from ..common import imageops
from ..common.predictor import BasePredictor, ModelLoader
from ..detection.heatmap import clean_boxes
from ..settings import settings
from .config import ID_TO_LABEL, LayoutConfig, layout_config
from .model import HipLayoutModel
from .schema import LayoutBox, LayoutResult
from .slicer import ImageSlicer

LAYOUT_SLICE_MIN = {"height": 1500, "width": 1500}        # surya/settings.py:101-105
LAYOUT_SLICE_SIZE = {"height": 1200, "width": 1200}
LAYOUT_MAX_BOXES = 100                                     # surya/settings.py:108


def _round_bf16(a: np.ndarray) -> np.ndarray:
    """fp32 values rounded to the nearest bfloat16 (ties to even), kept as fp32 -- what a bf16 tensor op leaves behind."""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32)
    return ((u + (((u >> 16) & 1) + 0x7FFF)) & 0xFFFF0000).view(np.float32)


def polygons_of_predictions(preds: np.ndarray, sizes: np.ndarray, bbox_scaler, skew_scaler, skew_min=0.001, dtype="float32") -> np.ndarray:
    """surya/layout/util.py:4-40 for a batch: preds [n, >= 6] (the fed-back token as floats), sizes [n, 2] = (width, height) ->
    float64 [n, 4, 2]. The reference does the corner arithmetic with TENSOR ops, i.e. in the model dtype (every intermediate rounded to
    fp32, or to bf16 for a bf16 model), and only the final `.item() * scale` in Python floats; `dtype` names that dtype."""
    R = _round_bf16 if dtype in ("bfloat16", "bf16") else (lambda a: a)
    p = R(np.asarray(preds, np.float32))
    sz = np.asarray(sizes, np.float64).reshape(-1, 2)
    w_scale, h_scale = sz[:, 0] / bbox_scaler, sz[:, 1] / bbox_scaler
    two = np.float32(2)
    cx, cy = p[:, 0], p[:, 1]
    hw, hh = R(p[:, 2] / two), R(p[:, 3] / two)
    x1, y1, x2, y2 = R(cx - hw), R(cy - hh), R(cx + hw), R(cy + hh)
    skew_x = np.floor(R(R(p[:, 4] - np.float32(skew_scaler)) / two))
    skew_y = np.floor(R(R(p[:, 5] - np.float32(skew_scaler)) / two))
    skew_x = np.where(np.abs(skew_x) < skew_min, np.float32(0), skew_x)
    skew_y = np.where(np.abs(skew_y) < skew_min, np.float32(0), skew_y)
    xs = np.stack([R(x1 - skew_x), R(x2 - skew_x), R(x2 + skew_x), R(x1 + skew_x)], -1).astype(np.float64) * w_scale[:, None]
    ys = np.stack([R(y1 - skew_y), R(y1 + skew_y), R(y2 + skew_y), R(y2 - skew_y)], -1).astype(np.float64) * h_scale[:, None]
    return np.stack([xs, ys], -1)


def prediction_to_polygon(pred, img_size, bbox_scaler, skew_scaler, skew_min=0.001, dtype="float32"):
    """surya/layout/util.py:4-40 on one length-7 vector (cx, cy, w, h, xskew, yskew, label)."""
    v = np.asarray([float(pred[i]) for i in range(6)], np.float32)[None]
    return polygons_of_predictions(v, np.asarray(img_size)[None], bbox_scaler, skew_scaler, skew_min, dtype)[0].tolist()


class LayoutImageProcessor:
    """SuryaEncoderImageProcessor (surya/common/donut/processor.py:24-126) for the layout model: every slice is resized straight to
    max_size (no aspect preservation) with cv2 interpolation flag 2 -- the reference passes PIL's BILINEAR constant to cv2.resize,
    which reads it as INTER_CUBIC --, rescaled by 1/255 in fp64 and normalised with mean = std = 0.5. cv2 is absent from the image:
    the cubic resample is common/imageops' restatement rounded to uint8 (cv2 returns uint8 for uint8 input)."""
    image_mean = np.array((0.5, 0.5, 0.5), np.float32)
    image_std = np.array((0.5, 0.5, 0.5), np.float32)

    def __init__(self, max_size, image_mean=None, image_std=None):
        self.max_size = max_size
        if image_mean is not None:
            self.image_mean = np.asarray(image_mean, np.float32)
        if image_std is not None:
            self.image_std = np.asarray(image_std, np.float32)

    def __call__(self, images: List[Image.Image]):
        out = []
        W, H = self.max_size["width"], self.max_size["height"]
        for img in images:
            a = np.asarray(img, dtype=np.uint8)
            assert a.ndim == 3 and a.shape[2] == 3
            r = imageops.resize(a.astype(np.float32), W, H, "cubic") if a.shape[:2] != (H, W) else a.astype(np.float32)
            r = np.clip(np.rint(r), 0, 255).astype(np.float32).transpose(2, 0, 1)
            r = (r.astype(np.float64) * (1 / 255)).astype(np.float32)
            out.append(((r - self.image_mean[:, None, None]) / self.image_std[:, None, None]).astype(np.float32))
        return {"pixel_values": out}


class LayoutModelLoader(ModelLoader):
    """checkpoint: None / config name (synthetic weights), {"config": LayoutConfig, "state_dict": {...}}, or a directory in the
    reference's on-disk format (surya/layout/loader.py:25-64): config.json with `encoder` / `decoder` sub-configs and *.safetensors
    with the reference's parameter names (`encoder.*`, `decoder.*`). The image processor needs no file (loader.py:66-70)."""

    def __init__(self, checkpoint=None):
        super().__init__(checkpoint)
        ck = checkpoint
        if isinstance(ck, dict):
            self._cfg, self._sd = ck["config"], ck["state_dict"]
        elif isinstance(ck, str) and os.path.isdir(ck):
            from .config import layout_config_from_reference_json, read_checkpoint_dir
            raw, self._sd, _ = read_checkpoint_dir(ck)
            self._cfg = layout_config_from_reference_json(raw)
        else:
            from ..synth import make_layout_weights
            self._cfg = layout_config(ck if isinstance(ck, str) else "LAYOUT-DEFAULT")
            self._sd = make_layout_weights(self._cfg, 0)

    def model(self, device=None, dtype=None, max_batch=None) -> HipLayoutModel:
        if device is None or device == "cuda":
            device = "cuda:0"
        return HipLayoutModel(self._cfg, self._sd, dtype=dtype or torch.bfloat16, device=device,
                              max_batch=max_batch or LayoutPredictor.default_batch_sizes["cuda"], max_boxes=LAYOUT_MAX_BOXES)

    def processor(self, device=None, dtype=None) -> LayoutImageProcessor:
        h, w = self._cfg.encoder.image_size
        return LayoutImageProcessor({"height": h, "width": w})


class LayoutPredictor(BasePredictor):
    model_loader_cls = LayoutModelLoader
    batch_size = None
    # "cuda": the reference's 32 (surya/layout/__init__.py:21-26) is sized for consumer cards; on MI355X the engine call is launch-bound at 32
    # pages (547 us per decode step = 88 launches at M = 32) and 128 pages per call run 2x the pages/s (bench.py layout leg), so a caller who
    # hands >= 128 pages and no batch_size gets 128-page engine calls; fewer pages -> one call of that many
    default_batch_sizes = {"cpu": 4, "mps": 4, "cuda": 128, "xla": 16}

    # Multi-GPU: when set, ONE call's pages are dealt over the ranks of the initialised process group and the per-page
    # results all-gathered (common/predictor.sharded_over_ranks). Off by default, like DetectionPredictor.shard_pages.
    shard_pages: bool = settings.SURYA_AMD_SHARD
    process_group = None

    def __call__(self, images: List[Image.Image], batch_size: Optional[int] = None, top_k: int = 5) -> List[LayoutResult]:
        if self.shard_pages:
            from ..common.predictor import sharded_over_ranks
            out = sharded_over_ranks(images, lambda mine: self.batch_layout_detection(mine, top_k=top_k, batch_size=batch_size),
                                     self.model.device, self.process_group)
            if out is not None:
                return out
        return self.batch_layout_detection(images, top_k=top_k, batch_size=batch_size)

    def batch_layout_detection(self, images: List[Image.Image], batch_size: Optional[int] = None, top_k: int = 5) -> List[LayoutResult]:
        assert all(isinstance(image, Image.Image) for image in images)
        if batch_size is None:
            batch_size = self.get_batch_size()
        batch_size = min(batch_size, self.model.max_batch)
        slicer = ImageSlicer(LAYOUT_SLICE_MIN, LAYOUT_SLICE_SIZE)
        counts = [slicer.slice_count(image) for image in images]
        batches, start, end = [], 0, 1                                   # the reference's batching by slice count (:53-66)
        while end < len(counts):
            if sum(counts[start:end]) >= batch_size or sum(counts[start:end + 1]) > batch_size:
                batches.append((start, end))
                start = end
            end += 1
        if start < len(counts):
            batches.append((start, len(counts)))
        dcfg = self.model.config.decoder
        results: List[LayoutResult] = []
        for start, end in batches:
            batch_images = [image.convert("RGB") for image in images[start:end]]
            batch_images, tile_positions = slicer.slice(batch_images)
            orig_sizes = [image.size for image in batch_images]
            batch_results = []
            for s0 in range(0, len(batch_images), self.model.max_batch):     # a page's slices may exceed the model's batch
                chunk = batch_images[s0:s0 + self.model.max_batch]
                batch_results.extend(self._detect_chunk(chunk, orig_sizes[s0:s0 + self.model.max_batch], dcfg, top_k))
            assert len(batch_results) == len(tile_positions)
            results.extend(slicer.join(batch_results, tile_positions))
        assert len(results) == len(images)
        return results

    def _detect_chunk(self, chunk, orig_sizes, dcfg, top_k) -> List[LayoutResult]:
        """The greedy box loop of surya/layout/__init__.py:110-177 for one encoder batch. One model call per box as there, served from
        device-fed runs (model.FedRuns): the host derives every fed-back token itself and the run records are checked against it. The
        per-step host work covers all unfinished pages at once (numpy / one batched softmax + top-k) instead of a Python loop per page."""
        from .model import FedRuns
        n = len(chunk)
        px = torch.from_numpy(np.stack(self.processor(chunk)["pixel_values"]))
        if torch.device(self.model.device).type == "cuda":          # (a host stand-in model of the CPU tests takes the tensor as it is)
            px = px.pin_memory().to(self.model.device, non_blocking=True)
        self.model.encode(px.contiguous())
        assert dcfg.pause_token_count == 0, "pause tokens in the decoder prompt are not built"
        sizes = np.asarray(orig_sizes, np.int64).reshape(n, 2)
        self.model.set_feedback(sizes)
        runs = FedRuns(self.model, 0, LAYOUT_MAX_BOXES, settings.LAYOUT_STEPS_PER_SYNC)
        mdtype = "bfloat16" if getattr(self.model, "dtype", torch.float32) == torch.bfloat16 else "float32"
        boxes = np.full((n, 7), dcfg.bos_token_id, np.int32)
        preds = [[] for _ in range(n)]
        all_done = np.zeros(n, bool)
        sp = dcfg.special_token_count
        header_footer = [k + sp for k, v in ID_TO_LABEL.items() if v in ("PageHeader", "PageFooter")]
        for position in range(LAYOUT_MAX_BOXES):
            cls, box = runs.step(boxes, ~all_done)           # finished pages: the device's rule keeps running, the host's does not
            class_preds = cls.argmax(-1)
            box_preds = box * dcfg.bbox_size
            if mdtype == "bfloat16":
                box_preds = _round_bf16(box_preds)                   # a tensor op in the model dtype (exact for bbox_size = 1024)
            all_done |= (class_preds == dcfg.eos_token_id) | (class_preds == dcfg.pad_token_id)
            if all_done.all():
                break
            nxt = np.concatenate([box_preds, class_preds[:, None].astype(np.float32)], -1)      # float tokens, truncated below (:131)
            act = np.nonzero(~all_done)[0]
            polys = polygons_of_predictions(nxt[act], sizes[act], dcfg.bbox_size, dcfg.skew_scaler, dtype=mdtype)
            logits = cls[act].copy()
            w, h = sizes[act, 0], sizes[act, 1]
            # page headers / footers in the middle of a page take their next-best label (:158-169)
            mid = (np.isin(class_preds[act], header_footer) & (polys[:, 0, 1] < h * .8) & (polys[:, 2, 1] > h * .2) & (polys[:, 0, 0] < w * .8)
                   & (polys[:, 2, 0] > w * .2))
            if mid.any():
                rows = np.nonzero(mid)[0]
                logits[rows, class_preds[act][rows]] = 0
                nxt[act[rows], 6] = logits[rows].argmax(-1)
            probs, idx = torch.topk(torch.softmax(torch.from_numpy(logits), dim=-1), k=top_k, dim=-1)
            for r, j in enumerate(act):
                p = nxt[j].copy()
                preds[j].append({"token": p, "polygon": polys[r].tolist(), "label": int(p[6]) - sp, "top_k_probs": probs[r], "top_k_indices": idx[r]})
            boxes = nxt.astype(np.int64).astype(np.int32)
        out = []
        for j in range(n):
            lb = []
            keep = [p for p in preds[j] if p["token"][6] > sp]                # special tokens (pause ...) carry no box (:187)
            for z, p in enumerate(keep):
                top_k_dict = {ID_TO_LABEL.get(int(l)): float(pr) for l, pr in zip(p["top_k_indices"] - sp, p["top_k_probs"]) if l > 0}
                name = ID_TO_LABEL[int(p["label"])]
                lb.append(LayoutBox(polygon=p["polygon"], label=name, position=z, top_k=top_k_dict, confidence=top_k_dict[name]))
            out.append(LayoutResult(bboxes=clean_boxes(lb), image_bbox=[0, 0, orig_sizes[j][0], orig_sizes[j][1]]))
        return out
