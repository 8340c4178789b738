"""LabelShaper: the token layout of the table-recognition decoder (drop-in for surya/table_rec/shaper.py:8-145).

A decoder token is a row of numbers, one span per box property in BOX_PROPERTIES order:

    bbox (cx, cy, w, h, xskew + 512, yskew + 512) | category + 5 | merges + 5 | colspan | is_header + 5

regression properties take as many columns as they have values, classification properties one column holding the class index shifted
past the 5 special tokens. The conversions work on whole batches as float64 matrices (the reference walks dicts item by item; the
numbers it produces are the ones below, tests/test_oracle_vs_reference.py holds the two equal on random inputs)."""
from __future__ import annotations

import math
from typing import Dict, List

import numpy as np

from .config import BOX_DIM, BOX_PROPERTIES, SPECIAL_TOKENS

# corner k of a box = centre + SIGN[k] * half extent, then skewed by SKEW[k] * skew: TL, TR, BR, BL
_CORNER_SIGN = ((-1, -1), (1, -1), (1, 1), (-1, 1))
_SKEW_SIGN = ((-1, -1), (-1, 1), (1, 1), (1, -1))


class LabelShaper:
    def __init__(self):
        self.property_keys = [name for name, _, _ in BOX_PROPERTIES]
        self._kind = {name: kind for name, _, kind in BOX_PROPERTIES}
        self._values = {name: count for name, count, _ in BOX_PROPERTIES}
        self._span, column = {}, 0
        for name in self.property_keys:
            width = self._values[name] if self._kind[name] == "regression" else 1
            self._span[name] = (column, column + width)
            column += width
        self._columns = column

    # ---------------------------------------------------------------------------------------------------------- token layout
    def component_idx(self, key):
        if key not in self._span:
            raise ValueError(f"{key!r} is not a box property")
        return self._span[key]

    def component_idx_dict(self):
        return dict(self._span)

    def get_box_property(self, key, add_special_tokens=True):
        """(name, number of classes or values, kind); classification heads also cover the special tokens unless told otherwise."""
        if key not in self._kind:
            raise ValueError(f"{key!r} is not a box property")
        extra = SPECIAL_TOKENS if (add_special_tokens and self._kind[key] == "classification") else 0
        return key, self._values[key] + extra, self._kind[key]

    # ------------------------------------------------------------------------------------------------------- dicts -> tokens
    def dict_to_labels(self, label_components: List[dict]):
        """Property dicts -> token rows. Every bbox is clamped to [0, BOX_DIM] IN its dict first (the predictor keeps those dicts as its
        predictions, so they see the clamp, shaper.py:24-27)."""
        n = len(label_components)
        if n == 0:
            return []
        rows = np.zeros((n, self._columns), np.float64)
        for name, (lo, hi) in self._span.items():
            try:
                column = [item[name] for item in label_components]
            except KeyError:
                raise ValueError(f"a label component has no {name!r} entry") from None
            if self._kind[name] == "classification":
                assert all(isinstance(v, int) for v in column), f"{name}: class indices must be ints"
                rows[:, lo] = np.asarray(column, np.float64) + SPECIAL_TOKENS
                continue
            values = np.asarray(column, np.float64).reshape(n, -1)
            assert values.shape[1] == hi - lo, f"{name}: {hi - lo} values per box expected"
            if name == "bbox":
                values = np.clip(values, 0, BOX_DIM)
                for item, clamped in zip(label_components, values.tolist()):
                    item["bbox"][:] = clamped
            rows[:, lo:hi] = values
        return rows.tolist()

    def convert_polygons_to_bboxes(self, label_components: List[Dict]):
        """4 corners (clipped to the box space) -> bbox (cx, cy, w, h, xskew, yskew), the skews shifted by BOX_DIM // 2 into positive
        numbers (:82-111). Sums are written corner by corner: the order of the additions is part of the result."""
        if not label_components:
            return label_components
        p = np.clip(np.asarray([item["polygon"] for item in label_components], np.float64), 0, BOX_DIM)
        x, y = p[:, :, 0], p[:, :, 1]
        (x1, x2, x3, x4), (y1, y2, y3, y4) = x.T, y.T
        bbox = np.stack([(x1 + x2 + x3 + x4) / 4, (y1 + y2 + y3 + y4) / 4,
                         (x2 + x3) / 2 - (x1 + x4) / 2, (y3 + y4) / 2 - (y2 + y1) / 2,
                         (x3 + x4) / 2 - (x1 + x2) / 2 + BOX_DIM // 2, (y2 + y3) / 2 - (y1 + y4) / 2 + BOX_DIM // 2], -1)
        for item, row in zip(label_components, bbox.tolist()):
            item["bbox"] = row
        return label_components

    # ------------------------------------------------------------------------------------------------------- tokens -> shapes
    def convert_bbox_to_polygon(self, box, skew_scaler=BOX_DIM // 2, skew_min=.001):
        """bbox -> its 4 corners TL, TR, BR, BL (:113-145): whole-number skews (floor of half the shifted value), none below skew_min."""
        centre, half = (box[0], box[1]), (box[2] / 2, box[3] / 2)
        skew = [math.floor((box[4 + axis] - skew_scaler) / 2) for axis in (0, 1)]
        skew = [0 if abs(s) < skew_min else s for s in skew]
        return [[centre[axis] + sign[axis] * half[axis] + lean[axis] * skew[axis] for axis in (0, 1)]
                for sign, lean in zip(_CORNER_SIGN, _SKEW_SIGN)]
