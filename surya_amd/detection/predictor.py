"""DetectionPredictor: drop-in for surya.detection.DetectionPredictor on MI355X.

Same call signature / result schema (surya/detection/__init__.py:22-48, detection/schema.py:12-17). The model forward,
sigmoid and x4 upsample run in libsurya_amd.so (HipDetModel); tall-page splitting, PIL resizing and heat-map -> box
post-processing stay on the host as in the reference. No CPU fallback for the model.
"""
from __future__ import annotations

import math
import os
from concurrent.futures import ThreadPoolExecutor
from typing import Generator, List, Tuple

import numpy as np
import torch
from PIL import Image, ImageOps

from ..common.imageops import copy_pool, page_pixels, parallel_copy
from ..common.predictor import BasePredictor, ModelLoader, gc_paused
from ..config import DetConfig, det_config
from ..settings import settings
from .heatmap import TextDetectionResult, parallel_get_boxes, result_from_device_boxes
from ..common.pil_resample import plan as plan_resize
from .model import DeviceResampler, HipDetModel, HipDetPost


class SegformerImageProcessor:
    """rescale 1/255 + ImageNet normalise -> CHW float32 (surya/detection/processor.py:126-146); `size` from the checkpoint.
    The reference's rescale step multiplies in float64 and rounds to float32 once (transformers' image_transforms.rescale), so the
    scaled pixel is the correctly rounded px / 255; a float32 product px * float32(1/255) is one ulp off for some pixel values
    (found by the live cross-check against the reference's own processor, tests/test_oracle_vs_reference.py)."""
    image_mean = np.array([0.485, 0.456, 0.406], np.float32)
    image_std = np.array([0.229, 0.224, 0.225], np.float32)

    def __init__(self, size=None):
        self.size = size or {"height": 512, "width": 512}

    def __call__(self, image: np.ndarray):
        a = ((image.astype(np.float64) * (1 / 255)).astype(np.float32) - self.image_mean) / self.image_std
        return {"pixel_values": [np.ascontiguousarray(a.transpose(2, 0, 1))]}


def get_total_splits(image_size, height):              # surya/detection/util.py:7-13
    return math.ceil(image_size[1] / height) if image_size[1] > settings.DETECTOR_IMAGE_CHUNK_HEIGHT else 1


def split_image(img: Image.Image, height: int, copy: bool = True):
    """Tall pages (> DETECTOR_IMAGE_CHUNK_HEIGHT px) are cut into `height`-px strips, the last one white-padded
    (surya/detection/util.py:16-36). copy=False hands a page that needs no cutting back as it is (for callers that only read
    its pixels: the reference copies because its prepare_image resizes in place)."""
    ih = img.size[1]
    if ih <= settings.DETECTOR_IMAGE_CHUNK_HEIGHT:
        return [img.copy() if copy else img], [ih]
    parts, heights = [], []
    for i in range(math.ceil(ih / height)):
        top, bottom = i * height, min((i + 1) * height, ih)
        crop = img.crop((0, top, img.size[0], bottom))
        if bottom - top < height:
            crop = ImageOps.pad(crop, (img.size[0], height), color=255, centering=(0, 0))
        parts.append(crop)
        heights.append(bottom - top)
    return parts, heights


def det_config_from_reference_json(raw: dict):
    """Map an EfficientViTConfig config.json (surya/detection/model/config.py:12-52) onto DetConfig."""
    from ..config import DetConfig
    kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in raw.items() if k in DetConfig.__dataclass_fields__}
    kw.setdefault("num_labels", raw.get("num_labels", len(raw.get("id2label", {})) or 2))
    return DetConfig(**{**kw, "name": "checkpoint"})


class DetectionModelLoader(ModelLoader):
    """checkpoint: None / config name (synthetic weights), {"config": DetConfig, "state_dict": ..., "size": int}, or a directory in
    the reference's on-disk format (surya/detection/loader.py:23-63): config.json (EfficientViTConfig), *.safetensors (the
    reference's parameter names) and preprocessor_config.json (SegformerImageProcessor: size, image_mean, image_std)."""

    def __init__(self, checkpoint=None):
        super().__init__(checkpoint)
        ck = checkpoint
        self._mean = self._std = None
        if isinstance(ck, dict):
            self._cfg, self._sd, self._size = ck["config"], ck["state_dict"], int(ck.get("size", 1024))
        elif isinstance(ck, str) and os.path.isdir(ck):
            import json
            from safetensors.torch import load_file
            with open(os.path.join(ck, "config.json")) as f:
                self._cfg = det_config_from_reference_json(json.load(f))
            self._sd = {}
            for fn in sorted(os.listdir(ck)):
                if fn.endswith(".safetensors"):
                    self._sd.update(load_file(os.path.join(ck, fn)))
            if not self._sd:
                raise FileNotFoundError(f"{ck}: no *.safetensors file")
            self._size = 1024
            pp = os.path.join(ck, "preprocessor_config.json")
            if os.path.exists(pp):
                with open(pp) as f:
                    raw = json.load(f)
                size = raw.get("size") or {}
                if size.get("height") != size.get("width"):
                    raise ValueError(f"{pp}: the detector runs on square processor sizes, got {size}")
                self._size = int(size.get("height", 1024))
                self._mean, self._std = raw.get("image_mean"), raw.get("image_std")
        else:
            from ..synth import make_det_weights
            self._cfg = det_config(ck if isinstance(ck, str) else settings.SURYA_AMD_DET_CONFIG)
            self._sd = make_det_weights(self._cfg, 0)
            self._size = int(os.environ.get("SURYA_AMD_DET_SIZE", "1024"))

    def model(self, device=None, dtype=None, max_batch=None) -> HipDetModel:
        if device is None or device == "cuda":
            device = "cuda:0"
        if dtype is None:
            dtype = torch.bfloat16        # the reference picks fp16 on GPUs (detection/loader.py:31); BASELINE asks bf16
        mb = max_batch or settings.DETECTOR_BATCH_SIZE or DetectionPredictor.default_batch_sizes["cuda"]
        bw = False
        if settings.SURYA_AMD_BROADCAST_WEIGHTS:
            from .. import dist as sdist
            bw = sdist.collectives_on()
        return HipDetModel(self._cfg, self._sd, height=self._size, width=self._size, dtype=dtype, device=device, max_batch=mb,
                           broadcast_weights=bw)

    def processor(self, device=None, dtype=None) -> SegformerImageProcessor:
        p = SegformerImageProcessor({"height": self._size, "width": self._size})
        if self._mean is not None:
            p.image_mean = np.asarray(self._mean, np.float32)
        if self._std is not None:
            p.image_std = np.asarray(self._std, np.float32)
        return p


class DetectionPredictor(BasePredictor):
    model_loader_cls = DetectionModelLoader
    batch_size = settings.DETECTOR_BATCH_SIZE
    default_batch_sizes = {"cpu": 8, "mps": 8, "cuda": 36, "xla": 18}

    # Multi-GPU: when set, ONE call's pages are dealt over the ranks (a page's strips stay on one rank), each
    # rank detects + post-processes its pages and the per-page results (boxes only, a few KB) are all-gathered as objects.
    # Every rank must pass the same pages (checked). Off by default, like RecognitionPredictor.shard_lines.
    shard_pages: bool = settings.SURYA_AMD_SHARD
    process_group = None

    def __call__(self, images: List[Image.Image], batch_size=None, include_maps=False) -> List[TextDetectionResult]:
        with gc_paused():                                  # result objects are acyclic; see common/predictor.py
            return self._call(images, batch_size, include_maps)

    def _call(self, images, batch_size=None, include_maps=False) -> List[TextDetectionResult]:
        if self.shard_pages:
            from ..common.predictor import sharded_over_ranks
            out = sharded_over_ranks(images, lambda mine: self._detect(mine, batch_size, include_maps), self.model.device, self.process_group)
            if out is not None:
                return out
        return self._detect(images, batch_size, include_maps)

    # Heat map -> boxes runs on the device (surya_det_boxes): what leaves the GPU per page is a few hundred 4-point boxes instead
    # of two fp32 maps (8 MB at 1024^2). DETECTOR_POSTPROCESS_HOST=1 keeps the host post-processing of the reference layout
    # (heat maps D2H + surya_amd/detection/heatmap.py on a thread pool): the checker of tests/test_gpu_det.py, not a fallback.
    device_postprocess: bool = not settings.DETECTOR_POSTPROCESS_HOST
    # the double LANCZOS resize of every page on the device; DETECTOR_RESIZE_HOST=1 keeps Pillow on a thread pool (the checker of
    # tests/test_gpu_resample.py; also what pages take whose resize needs Pillow's reduce() pre-pass)
    device_resize: bool = not settings.DETECTOR_RESIZE_HOST

    def _detect(self, images: List[Image.Image], batch_size=None, include_maps=False) -> List[TextDetectionResult]:
        if self.device_postprocess:
            return self._detect_device(images, batch_size, include_maps)
        gen = self.batch_detection(images, batch_size=batch_size)
        futures = []
        workers = max(1, min(settings.DETECTOR_POSTPROCESSING_CPU_WORKERS, len(images)))
        if len(images) >= settings.DETECTOR_MIN_PARALLEL_THRESH:
            with ThreadPoolExecutor(max_workers=workers) as ex:
                for preds, sizes in gen:
                    futures.extend(ex.submit(parallel_get_boxes, p, s, include_maps) for p, s in zip(preds, sizes))
            return [f.result() for f in futures]
        out = []
        for preds, sizes in gen:
            out.extend(parallel_get_boxes(p, s, include_maps) for p, s in zip(preds, sizes))
        return out

    def _detect_device(self, images, batch_size=None, include_maps=False) -> List[TextDetectionResult]:
        return [r for batch in self._iter_detect_device(images, batch_size, include_maps) for r in batch]

    def iter_detect(self, images: List[Image.Image], batch_size=None) -> Generator[List[TextDetectionResult], None, None]:
        """The results of `__call__(images, batch_size)` handed over batch by batch, in page order, each as soon as ITS boxes have
        left the device (the next batch is already launched by then). What RecognitionPredictor's streamed detect -> recognise
        call consumes (recognition/predictor.py `_call_streamed`): lines of the first pages are admitted to the continuous-batching
        loop while the detector still works on the later ones. Single process, device post-processing only."""
        if not self.device_postprocess or self.shard_pages:
            raise RuntimeError("iter_detect needs the device post-processing path of one process (no DETECTOR_POSTPROCESS_HOST, no page sharding)")
        return self._iter_detect_device(images, batch_size, False, eager_first=True)

    def _iter_detect_device(self, images, batch_size=None, include_maps=False, eager_first=False):
        """eager_first: hand the FIRST batch over as soon as its boxes exist instead of after the second batch has been prepared and
        launched -- a consumer that starts other device work from it (the streamed recognise call) gets going one host preparation
        (~10 ms per 16 pages) earlier, at the price of that much overlap inside the detector."""
        if getattr(self, "_post", None) is None:
            self._post = HipDetPost(self.model.device)
        tt, lt = settings.DETECTOR_TEXT_THRESHOLD, settings.DETECTOR_BLANK_THRESHOLD

        def finish(job):
            """Results of one launched batch (waits for ITS event only)."""
            out: List[TextDetectionResult] = []
            jobs, n_pages, tiles_of, split_heights, sizes, heat, pw = job
            res = {}
            for pages_, pending, psizes in jobs:
                for pg, (bx, cf), psize in zip(pages_, self._post.collect(pending), psizes):
                    res[pg] = (bx, cf, psize)
            maps = heat.cpu().numpy() if include_maps else None      # callers that want the maps pay for their D2H
            for pg in range(n_pages):
                bx, cf, psize = res[pg]
                hi = ai = None
                if include_maps:
                    # the reference keeps a page's FIRST tile whole -- an unsplit page shorter than the processor height still gets
                    # its full map -- and trims only the later strips of a split page to their valid rows (detection/__init__.py:134-151)
                    rows = [maps.shape[2] if k == 0 else split_heights[t] for k, t in enumerate(tiles_of[pg])]
                    hm = np.vstack([maps[t, 0, :r] for t, r in zip(tiles_of[pg], rows)])
                    am = np.vstack([maps[t, 1, :r] for t, r in zip(tiles_of[pg], rows)])
                    hi, ai = Image.fromarray((hm * 255).astype(np.uint8)), Image.fromarray((am * 255).astype(np.uint8))
                out.append(result_from_device_boxes(bx, cf, list(psize), sizes[pg], hi, ai))
            return out

        # One batch in flight ahead of the one being read: the generator prepares (PIL convert / resize, pinned staging) and
        # launches batch i + 1 while the GPU still works on batch i, whose boxes are only then waited for.
        prev = None
        for heat, split_index, split_heights, sizes in self.batch_heatmaps(images, batch_size):
            ph, pw = heat.shape[2], heat.shape[3]
            n_pages = split_index[-1] + 1
            tiles_of = [[i for i, k in enumerate(split_index) if k == page] for page in range(n_pages)]
            whole = [page for page in range(n_pages) if len(tiles_of[page]) == 1]
            jobs = []
            if whole:
                sel = heat if len(whole) == heat.shape[0] else heat[[tiles_of[pg][0] for pg in whole]].contiguous()
                jobs.append((whole, self._post.launch(sel, tt, lt), [(pw, ph)] * len(whole)))
            # tall pages: strips re-assembled on the device (:134-151); pages of equal height share ONE post-processing call (a call per
            # page was 17 launches + a D2H each: 16 letter pages spent 40 of their 77 ms there, tools/hostbench/det_letter_profile.py)
            by_height = {}
            for pg in range(n_pages):
                if len(tiles_of[pg]) > 1:
                    by_height.setdefault(sum(split_heights[t] for t in tiles_of[pg]), []).append(pg)
            for hf, pgs in by_height.items():
                full = torch.empty((len(pgs), hf, pw), dtype=heat.dtype, device=heat.device)
                for i, pg in enumerate(pgs):
                    r0 = 0
                    for t in tiles_of[pg]:
                        full[i, r0: r0 + split_heights[t]] = heat[t, 0, : split_heights[t]]
                        r0 += split_heights[t]
                jobs.append((pgs, self._post.launch(full, tt, lt), [(pw, hf)] * len(pgs)))
            job = (jobs, n_pages, tiles_of, split_heights, sizes, heat, pw)
            if prev is not None:
                yield finish(prev)
            prev = job
            if eager_first:
                eager_first = False
                yield finish(prev)
                prev = None
        if prev is not None:
            yield finish(prev)

    def resize_image(self, img: Image.Image) -> np.ndarray:
        """The reference's double LANCZOS resize to the processor size (surya/detection/__init__.py:50-57), uint8 HWC."""
        new_size = (self.processor.size["width"], self.processor.size["height"])
        if img.size != new_size:
            img = img.copy()                                          # thumbnail() works in place; the page may be the caller's
            img.thumbnail(new_size, Image.Resampling.LANCZOS)
            img = img.resize(new_size, Image.Resampling.LANCZOS)
        return page_pixels(img)

    def prepare_image(self, img: Image.Image) -> torch.Tensor:
        new_size = (self.processor.size["width"], self.processor.size["height"])
        img.thumbnail(new_size, Image.Resampling.LANCZOS)          # the reference's double resize (:50-57)
        img = img.resize(new_size, Image.Resampling.LANCZOS)
        arr = np.asarray(img, dtype=np.uint8)
        return torch.from_numpy(self.processor(arr)["pixel_values"][0])

    def batch_heatmaps(self, images: List, batch_size=None):
        """Split -> prepare_image -> model, like batch_detection (surya/detection/__init__.py:64-132), but the heat maps stay
        on the device: yields (heat cuda fp32 [tiles, labels, H, W], page index of each tile, valid rows of each tile, sizes)."""
        assert all(isinstance(im, Image.Image) for im in images)
        if batch_size is None:
            batch_size = self.get_batch_size()
        batch_size = min(batch_size, self.model.max_batch)
        ph = self.processor.size["height"]
        orig_sizes = [im.size for im in images]
        splits = [get_total_splits(s, ph) for s in orig_sizes]
        batches, cur, cur_n = [], [], 0
        for i in range(len(images)):                               # greedy packing by tile count (:77-90)
            if cur_n + splits[i] > batch_size:
                if cur:
                    batches.append(cur)
                cur, cur_n = [], 0
            cur.append(i)
            cur_n += splits[i]
        if cur:
            batches.append(cur)
        for idxs in batches:
            # the device path only READS the pages (zero-copy pixel views, resize on the device): no defensive copies of pages that
            # already are RGB (convert() and split_image() each copied 4 MB per 1024^2 page under the GIL; the Pillow resize of
            # the host path, which works in place, copies for itself in resize_image)
            batch_images = [images[j] if images[j].mode == "RGB" else images[j].convert("RGB") for j in idxs]
            split_index, split_heights, parts = [], [], []
            tall = sum(im.size[1] > settings.DETECTOR_IMAGE_CHUNK_HEIGHT for im in batch_images)
            cut = (list(copy_pool().map(lambda im: split_image(im, ph, copy=False), batch_images)) if tall > 1     # PIL crop / pad release the GIL
                   else [split_image(im, ph, copy=False) for im in batch_images])
            for k, (ps, hs) in enumerate(cut):
                parts.extend(ps)
                split_index.extend([k] * len(ps))
                split_heights.extend(hs)
            # pages go to the device as uint8; the reference's double LANCZOS resize to the processor size runs there too
            # (surya_resample_lanczos_u8, bit-identical to Pillow) unless Pillow would take a path the kernels do not restate
            # (a reduce() pre-pass for >= 4x shrinks: common/pil_resample.plan) or DETECTOR_RESIZE_HOST=1 asks for the host path;
            # rescale + normalise happen in the model's first kernel
            pw = self.processor.size["width"]
            plans = [None if not self.device_resize else plan_resize(p_.size[0], p_.size[1], (pw, ph)) for p_ in parts]
            pool = copy_pool()
            # pixels of every part: Pillow's own memory where it can export it (page_pixels), the Pillow double resize for the parts the
            # device path does not take; both release the GIL in their C loops, so the parts of a batch are prepared side by side
            if len(parts) > 1:
                px = list(pool.map(lambda k: self.resize_image(parts[k]) if plans[k] is None else page_pixels(parts[k]), range(len(parts))))
            else:
                px = [self.resize_image(parts[k]) if plans[k] is None else page_pixels(parts[k]) for k in range(len(parts))]
            # RGBX views of PIL's own memory where it can export them (page_pixels), else repacked RGB; one stride per batch
            ready = [k for k, pl in enumerate(plans) if pl is None or not pl]          # already at the processor size
            pix = 4 if all(px[k].shape[2] == 4 for k in ready) else 3
            host = torch.empty((len(parts), ph, pw, pix), dtype=torch.uint8, pin_memory=True)   # caching host allocator
            hv = host.numpy()
            parallel_copy([hv[k] for k in ready], [px[k] if px[k].shape[2] == pix else px[k][..., :3] for k in ready])
            dev_batch = host.to(self.model.device, non_blocking=True)
            todo = [k for k, pl in enumerate(plans) if pl]
            if todo:
                # the parts that need the double LANCZOS resize: ONE pinned staging buffer and ONE H2D copy for the batch (a pinned
                # allocation + copy + transfer per page cost ~4.7 ms of host time each: 210 pages/s on letter pages), then the resize
                # passes part by part on the device
                if getattr(self, "_resampler", None) is None:
                    self._resampler = DeviceResampler(self.model.device)
                sizes_b = [int(px[k].nbytes) for k in todo]
                offs_b = np.concatenate([[0], np.cumsum([(b + 255) & ~255 for b in sizes_b])]).astype(np.int64)
                stage = torch.empty(int(offs_b[-1]), dtype=torch.uint8, pin_memory=True)
                sv = stage.numpy()
                parallel_copy([sv[int(o): int(o) + b].reshape(px[k].shape) for k, o, b in zip(todo, offs_b[:-1], sizes_b)],
                              [px[k] for k in todo])
                dev_stage = stage.to(self.model.device, non_blocking=True)
                for k, o, b in zip(todo, offs_b[:-1], sizes_b):
                    cur = dev_stage[int(o): int(o) + b].view(px[k].shape)
                    pl = plans[k]
                    for i, tgt in enumerate(pl):                            # thumbnail's size, then the processor size
                        cur = self._resampler.resize(cur, tgt, out=dev_batch[k] if i == len(pl) - 1 else None)
            heat_parts = []
            for s in range(0, len(parts), self.model.max_batch):            # a single page may exceed max_batch tiles
                heat_parts.append(self.model.forward_u8(dev_batch[s: s + self.model.max_batch], self.processor.image_mean,
                                                        self.processor.image_std))
            heat = heat_parts[0] if len(heat_parts) == 1 else torch.cat(heat_parts, 0)
            yield heat, split_index, [min(h, ph) for h in split_heights], [orig_sizes[j] for j in idxs]

    def batch_detection(self, images: List, batch_size=None) -> Generator[Tuple[List[List[np.ndarray]], List[Tuple[int, int]]], None, None]:
        assert all(isinstance(im, Image.Image) for im in images)
        if batch_size is None:
            batch_size = self.get_batch_size()
        batch_size = min(batch_size, self.model.max_batch)
        nlab = self.model.cfg.num_labels
        ph = self.processor.size["height"]
        orig_sizes = [im.size for im in images]
        splits = [get_total_splits(s, ph) for s in orig_sizes]
        batches, cur, cur_n = [], [], 0
        for i in range(len(images)):                               # greedy packing by tile count (:77-90)
            if cur_n + splits[i] > batch_size:
                if cur:
                    batches.append(cur)
                cur, cur_n = [], 0
            cur.append(i)
            cur_n += splits[i]
        if cur:
            batches.append(cur)
        for idxs in batches:
            batch_images = [images[j].convert("RGB") for j in idxs]
            split_index, split_heights, parts = [], [], []
            for k, im in enumerate(batch_images):
                ps, hs = split_image(im, ph)
                parts.extend(ps)
                split_index.extend([k] * len(ps))
                split_heights.extend(hs)
            tiles = torch.stack([self.prepare_image(p) for p in parts], 0).contiguous()
            heat_parts = []
            for s in range(0, tiles.shape[0], self.model.max_batch):        # a single page may exceed max_batch tiles
                chunk = tiles[s: s + self.model.max_batch]
                if torch.device(self.model.device).type == "cuda":           # (a stand-in model on the CPU in the live cross-check)
                    chunk = chunk.pin_memory().to(self.model.device, non_blocking=True)
                heat_parts.append(self.model.forward(chunk))
            logits = torch.cat(heat_parts, 0).cpu().numpy()                 # fp32, one D2H per batch (:132)
            preds: List[List[np.ndarray]] = []
            for i, (idx, height) in enumerate(zip(split_index, split_heights)):
                maps = [logits[i][k] for k in range(nlab)]
                if len(preds) <= idx:
                    preds.append(maps)
                else:
                    if height < ph:
                        maps = [m[:height, :] for m in maps]               # cut the white padding of the last strip
                    preds[idx] = [np.vstack([preds[idx][k], maps[k]]) for k in range(nlab)]
            yield preds, [orig_sizes[j] for j in idxs]
