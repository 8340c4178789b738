"""Large pages are cut into slices along their long axis before layout detection and the per-slice results merged afterwards
(host logic of surya/layout/slicer.py:11-139: same thresholds, same merge rule)."""
from __future__ import annotations

import math
from typing import List, Tuple

from PIL import Image

from .schema import LayoutResult


class ImageSlicer:
    merge_tolerance = .05
    merge_margin = .05

    def __init__(self, slice_min_dims, slice_sizes, max_slices=4):
        self.slice_min_dims, self.slice_sizes, self.max_slices = slice_min_dims, slice_sizes, max_slices

    def _slice_size(self, dimension: int, dim_type: str) -> int:
        return max(self.slice_sizes[dim_type], dimension // self.max_slices + 1)

    def slice_count(self, image: Image.Image) -> int:
        w, h = image.size
        return math.ceil(w / self._slice_size(w, "width")) if w > h else math.ceil(h / self._slice_size(h, "height"))

    def slice(self, images: List[Image.Image]):
        slices, positions = [], []
        for idx, image in enumerate(images):
            w, h = image.size
            if w > self.slice_min_dims["width"] or h > self.slice_min_dims["height"]:
                if w > h:
                    size = self._slice_size(w, "width")
                    for i, x in enumerate(range(0, w, size)):
                        slices.append(image.crop((x, 0, min(x + size, w), h)))
                        positions.append((idx, i, 0))
                else:
                    size = self._slice_size(h, "height")
                    for i, y in enumerate(range(0, h, size)):
                        slices.append(image.crop((0, y, w, min(y + size, h))))
                        positions.append((idx, 0, i))
            else:
                slices.append(image)
                positions.append((idx, 0, 0))
        return slices, positions

    def join(self, results: List[LayoutResult], tile_positions: List[Tuple[int, int, int]]) -> List[LayoutResult]:
        out, cur = [], None
        for idx, (result, (image_idx, tile_x, tile_y)) in enumerate(zip(results, tile_positions)):
            if idx == 0 or image_idx != tile_positions[idx - 1][0]:
                if cur is not None:
                    out.append(cur)
                cur = result
            else:
                cur = self.merge_results(cur, result, "width" if tile_x > 0 else "height")
        if cur is not None:
            out.append(cur)
        return out

    def merge_results(self, res1: LayoutResult, res2: LayoutResult, merge_dir="width") -> LayoutResult:
        bbox = res1.image_bbox.copy()
        remove = set()
        horizontal = merge_dir == "width"
        if horizontal:
            bbox[2] += res2.image_bbox[2]
        else:
            bbox[3] += res2.image_bbox[3]
        max_position = max([box.position for box in res1.bboxes]) + 1       # (raises on an empty first slice, as the reference does)
        for i, box2 in enumerate(res2.bboxes):
            if horizontal:
                box2.shift(x_shift=res1.image_bbox[2])
            else:
                box2.shift(y_shift=res1.image_bbox[3])
            box2.position += max_position
            for box1 in res1.bboxes:
                margin = {"x_margin": self.merge_margin} if horizontal else {"y_margin": self.merge_margin}
                touch = (box1.intersection_pct(box2, **margin) > self.merge_tolerance or
                         box2.intersection_pct(box1, **margin) > self.merge_tolerance)
                if horizontal:
                    aligned = box1.y_overlap(box2) > box1.height // 2 or box2.y_overlap(box1) > box2.height // 2
                else:
                    aligned = box1.x_overlap(box2) > box1.width // 2 or box2.x_overlap(box1) > box2.width // 2
                same = box1.label == box2.label or (box1.label in ["Picture", "Figure"] and box2.label in ["Picture", "Figure"])
                if touch and aligned and same:
                    box1.merge(box2)
                    remove.add(i)
        return LayoutResult(image_bbox=bbox, bboxes=res1.bboxes + [b for i, b in enumerate(res2.bboxes) if i not in remove], sliced=True)
