"""ImageSlicer: pages that are too large for one pass of the layout model are cut into strips along their long side and the per-strip
LayoutResults stitched back together (drop-in for surya/layout/slicer.py:11-139: same thresholds, same strip geometry, same merge rule
for boxes that a cut went through).

The geometry lives in one place (`_strips`): a page is either left whole or cut into equal steps of max(slice size, side // max_slices
+ 1) pixels along the longer side, the last strip taking what is left."""
from __future__ import annotations

from typing import List, Tuple

from PIL import Image

from .schema import LayoutResult

_AXES = {"width": 0, "height": 1}


class ImageSlicer:
    merge_tolerance = .05            # share of a box that must lie inside its neighbour (after the margin) to count as touching
    merge_margin = .05               # boxes are widened by this share of the smaller one, across the cut, before that test

    def __init__(self, slice_min_dims, slice_sizes, max_slices=4):
        self.slice_min_dims, self.slice_sizes, self.max_slices = slice_min_dims, slice_sizes, max_slices

    # ------------------------------------------------------------------------------------------------------------- geometry
    def _step(self, size: Tuple[int, int]):
        """(name of the long side, strip length along it)."""
        side = "width" if size[0] > size[1] else "height"
        extent = size[_AXES[side]]
        return side, max(self.slice_sizes[side], extent // self.max_slices + 1)

    def _strips(self, size: Tuple[int, int]):
        """[(crop box, (tile_x, tile_y))] of a page; one whole-page entry when the page is within the minimum dimensions."""
        w, h = size
        if w <= self.slice_min_dims["width"] and h <= self.slice_min_dims["height"]:
            return [((0, 0, w, h), (0, 0))]
        side, step = self._step(size)
        out = []
        for k, start in enumerate(range(0, size[_AXES[side]], step)):
            stop = min(start + step, size[_AXES[side]])
            out.append(((start, 0, stop, h), (k, 0)) if side == "width" else ((0, start, w, stop), (0, k)))
        return out

    def slice_count(self, image: Image.Image) -> int:
        """Strips a page WOULD be cut into (the batching of LayoutPredictor counts them for every page, small ones included)."""
        side, step = self._step(image.size)
        return -(-image.size[_AXES[side]] // step)

    def slice(self, images: List[Image.Image]):
        pieces, positions = [], []
        for index, image in enumerate(images):
            strips = self._strips(image.size)
            for box, (tx, ty) in strips:
                pieces.append(image if len(strips) == 1 and box == (0, 0) + image.size else image.crop(box))
                positions.append((index, tx, ty))
        return pieces, positions

    # -------------------------------------------------------------------------------------------------------------- stitching
    def join(self, results: List[LayoutResult], tile_positions: List[Tuple[int, int, int]]) -> List[LayoutResult]:
        pages: List[LayoutResult] = []
        last_page = None
        for result, (page, tile_x, _tile_y) in zip(results, tile_positions):
            if page != last_page:
                pages.append(result)
            else:
                pages[-1] = self.merge_results(pages[-1], result, "width" if tile_x > 0 else "height")
            last_page = page
        return pages

    def _continues(self, first, second, across: str) -> bool:
        """`second` (already shifted into page coordinates) is the part of `first` beyond a cut made across `across`."""
        margin = {"x_margin" if across == "width" else "y_margin": self.merge_margin}
        touching = max(first.intersection_pct(second, **margin), second.intersection_pct(first, **margin)) > self.merge_tolerance
        if across == "width":
            lined_up = first.y_overlap(second) > first.height // 2 or second.y_overlap(first) > second.height // 2
        else:
            lined_up = first.x_overlap(second) > first.width // 2 or second.x_overlap(first) > second.width // 2
        pictures = ("Picture", "Figure")
        alike = first.label == second.label or (first.label in pictures and second.label in pictures)
        return touching and lined_up and alike

    def merge_results(self, res1: LayoutResult, res2: LayoutResult, merge_dir="width") -> LayoutResult:
        """res2 = the strip after res1 along `merge_dir`: its boxes move by res1's extent and continue res1's reading order; a box that
        continues one of res1's boxes is absorbed by it (the absorbing box grows at once, so it may swallow further parts)."""
        axis = _AXES[merge_dir]
        page_box = list(res1.image_bbox)
        page_box[2 + axis] += res2.image_bbox[2 + axis]
        offset = res1.image_bbox[2 + axis]
        next_position = max(box.position for box in res1.bboxes) + 1        # (an empty first strip raises, as in the reference)
        kept = []
        for part in res2.bboxes:
            part.shift(**({"x_shift": offset} if axis == 0 else {"y_shift": offset}))
            part.position += next_position
            absorbed = False
            for whole in res1.bboxes:
                if self._continues(whole, part, merge_dir):
                    whole.merge(part)
                    absorbed = True
            if not absorbed:
                kept.append(part)
        return LayoutResult(image_bbox=page_box, bboxes=res1.bboxes + kept, sliced=True)
