"""Device-side line pre-processing: page pixels -> image_tiles in one library call (csrc/rec_prep.h).

The host computes only integers here -- which page, which rectangle / polygon, the sizes scale_to_fit and the x28 round-up
produce (surya/common/surya/processor/__init__.py:141-230 arithmetic) and where each line's patch rows go; cropping, the
outside-polygon pad, both resizes, normalisation and patchify run on the GPU. Pages cross PCIe once as uint8.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from .. import _lib as L
from ..common.imageops import page_pixels, parallel_copy  # noqa: F401  (page_pixels is re-exported: the predictor and tests import it from here)

# must match sa::prep::LineDesc (csrc/rec_prep.h); align=True reproduces the C layout (8-byte longs after the int block)
LINE_DESC = np.dtype([("page_off", np.int64), ("page_w", np.int32), ("page_h", np.int32), ("x0", np.int32), ("y0", np.int32),
                      ("cw", np.int32), ("ch", np.int32), ("has_poly", np.int32), ("poly", np.float32, (8,)), ("mid_w", np.int32),
                      ("mid_h", np.int32), ("out_w", np.int32), ("out_h", np.int32), ("mask_off", np.int64), ("mid_off", np.int64),
                      ("tile_row", np.int64)], align=True)
assert LINE_DESC.itemsize == 112, LINE_DESC.itemsize


@dataclass
class LineRef:
    """A line of a page by reference (no pixels): what RecognitionPredictor's `flat["slices"]` holds on the device path.
    `shape` is the shape the cropped array would have (slice_bboxes_from_image / slice_and_pad_poly), which is all the
    predictor reads from a slice besides its pixels (width sort, polygon scaling)."""
    page: int
    x0: int
    y0: int
    x1: int
    y1: int
    poly: Optional[Tuple[Tuple[int, int], ...]] = None      # absolute page coordinates, 4 vertices

    @property
    def shape(self):
        return (max(self.y1 - self.y0, 0), max(self.x1 - self.x0, 0), 3)

    @property
    def size(self):
        return self.shape[0] * self.shape[1] * 3


def bbox_ref(page: int, page_w: int, page_h: int, bbox) -> LineRef:
    """The rectangle slice_bboxes_from_image cuts (surya/input/processing.py:35-54: clip at 0, at least 1 px, clip to the page)."""
    b = [max(int(v), 0) for v in bbox]
    if b[3] <= b[1]:
        b[3] = b[1] + 1
    if b[2] <= b[0]:
        b[2] = b[0] + 1
    b[2], b[3] = min(b[2], page_w), min(b[3], page_h)
    return LineRef(page, b[0], b[1], b[2], b[3])


def poly_ref(page: int, page_w: int, page_h: int, coords) -> LineRef:
    """The bounding rectangle slice_and_pad_poly cuts (:64-101) + the polygon for the pad mask. numpy slicing clips at the
    page's far edges (and a negative start would wrap, which no caller produces: detection boxes are fitted to the page)."""
    pts = tuple((int(c[0]), int(c[1])) for c in coords)
    x0, y0 = min(p[0] for p in pts), min(p[1] for p in pts)
    x1, y1 = max(p[0] for p in pts), max(p[1] for p in pts)
    return LineRef(page, max(x0, 0), max(y0, 0), min(x1, page_w), min(y1, page_h), pts)


def fit_sizes(h: int, w: int, max_size, min_size=(168, 168), factor: int = 28):
    """(mid_h, mid_w) after scale_to_fit and (out_h, out_w) after the round-up to multiples of `factor`."""
    cur, mx, mn = w * h, max_size[0] * max_size[1], min_size[0] * min_size[1]
    mw, mh = w, h
    if cur > mx:
        s = (mx / cur) ** 0.5
        mw, mh = math.floor(w * s), math.floor(h * s)
    elif cur < mn:
        s = (mn / cur) ** 0.5
        mw, mh = math.ceil(w * s), math.ceil(h * s)
    return (mh, mw), (math.ceil(mh / factor) * factor, math.ceil(mw / factor) * factor)


class DevicePreprocessor:
    def __init__(self, device, patch_size: int = 14, merge_size: int = 2, pad_value: float = 255.0,
                 mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
        if not torch.cuda.is_available():
            raise L.SuryaAmdError("DevicePreprocessor needs a GPU (MI355X)")
        self.lib = L.lib()
        self.device = torch.device(device)
        self.ps, self.merge, self.pad = patch_size, merge_size, float(pad_value)
        self.mean = (C.c_float * 3)(*[float(np.float32(m)) for m in mean])
        self.std = (C.c_float * 3)(*[float(np.float32(s)) for s in std])

    def __call__(self, pages: Sequence[np.ndarray], lines: Sequence[LineRef], max_sizes: Sequence[Tuple[int, int]]):
        """pages: uint8 [H, W, 3] (RGB) or [H, W, 4] (RGBX) arrays; lines: LineRefs into them; max_sizes: the task's img_size per line.
        Returns (tiles cuda fp32 [sum P, 3 ps^2], tile_offs int64 [n + 1], grids [(gh, gw)])."""
        n = len(lines)
        f = self.ps * self.merge
        offs, total = [], 0
        pix = 4 if pages and all(pg.shape[2] == 4 for pg in pages) else 3
        if pix == 3:                                   # mixed strides: repack the RGBX views
            pages = [pg if pg.shape[2] == 3 else np.ascontiguousarray(pg[..., :3]) for pg in pages]
        for pg in pages:
            assert pg.dtype == np.uint8 and pg.ndim == 3 and pg.shape[2] == pix
            offs.append(total)
            total += pg.size
        desc = np.zeros(n, LINE_DESC)
        tile_offs = np.zeros(n + 1, np.int64)
        grids, mask_bytes, mid_floats, max_mid_w = [], 0, 0, 0
        for i, (ln, mx) in enumerate(zip(lines, max_sizes)):
            ph, pw = pages[ln.page].shape[:2]
            h, w = ln.y1 - ln.y0, ln.x1 - ln.x0
            if h <= 0 or w <= 0:
                raise ValueError("empty line crop on the device pre-processing path (caller substitutes a blank page)")
            (mh, mw), (oh, ow) = fit_sizes(h, w, mx, factor=f)
            d = desc[i]
            d["page_off"], d["page_w"], d["page_h"] = offs[ln.page], pw, ph
            d["x0"], d["y0"], d["cw"], d["ch"] = ln.x0, ln.y0, w, h
            if ln.poly is not None and len(ln.poly) >= 3:
                if len(ln.poly) != 4:
                    raise ValueError("device pre-processing handles 4-point polygons")
                d["has_poly"] = 1
                d["poly"] = np.asarray([(x - ln.x0, y - ln.y0) for x, y in ln.poly], np.float32).reshape(-1)
                d["mask_off"] = mask_bytes
                mask_bytes += h * w
            d["mid_w"], d["mid_h"], d["out_w"], d["out_h"] = mw, mh, ow, oh
            if (mh, mw) != (h, w):
                d["mid_off"] = mid_floats
                mid_floats += mh * mw * 3
                max_mid_w = max(max_mid_w, mw)
            d["tile_row"] = tile_offs[i]
            gh, gw = oh // self.ps, ow // self.ps
            grids.append((gh, gw))
            tile_offs[i + 1] = tile_offs[i] + gh * gw
        torch.cuda.set_device(self.device)
        # pinned staging straight from torch's caching host allocator (a pageable buffer + .pin_memory() would allocate, fault in and
        # copy the pages' ~3 MB each a second time)
        host = torch.empty(max(total, 1), dtype=torch.uint8, pin_memory=True)
        hv = host.numpy()
        parallel_copy([hv[o: o + pg.size].reshape(pg.shape) for pg, o in zip(pages, offs)], list(pages))   # 3-8 MB each: side by side
        d_pages = host.to(self.device, non_blocking=True)
        d_desc = torch.from_numpy(desc.view(np.uint8).reshape(-1)).to(self.device)
        d_mask = torch.empty(max(mask_bytes, 1), dtype=torch.uint8, device=self.device)
        d_mid = torch.empty(max(mid_floats, 1), dtype=torch.float32, device=self.device)
        tiles = torch.empty((int(tile_offs[-1]), 3 * self.ps * self.ps), dtype=torch.float32, device=self.device)
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        L.check(self.lib.surya_rec_preprocess(L.ptr(d_pages), L.ptr(d_desc), C.c_int(n), L.ptr(d_mask), L.ptr(d_mid), L.ptr(tiles),
                                              C.c_int(self.ps), C.c_int(self.merge), C.c_float(self.pad), self.mean, self.std,
                                              C.c_int(int(mask_bytes > 0)), C.c_int(max_mid_w), C.c_int(pix), stream),
                "surya_rec_preprocess")
        self._keep = (d_pages, d_desc, d_mask, d_mid, host)          # alive until the stream has consumed them
        return tiles, tile_offs, grids
