"""LabelShaper: token vectors <-> box property dicts of the table-recognition decoder (surya/table_rec/shaper.py:8-145).

A token is BOX_PROPERTIES laid end to end: bbox (cx, cy, w, h, xskew + 512, yskew + 512) | category + 5 | merges + 5 | colspan |
is_header + 5 (classification values are shifted past the 5 special tokens on the way INTO the model)."""
from __future__ import annotations

import math
from typing import Dict, List

import numpy as np

from .config import BOX_DIM, BOX_PROPERTIES, SPECIAL_TOKENS


class LabelShaper:
    def __init__(self):
        self.property_keys = [k for k, _, _ in BOX_PROPERTIES]

    def dict_to_labels(self, label_components: List[dict]):
        """:12-51. Clamps each bbox to [0, BOX_DIM] IN PLACE (the predictor's stored predictions see the clamp), then flattens."""
        if not label_components:
            return []
        for k, kcount, mode in BOX_PROPERTIES:
            for lc in label_components:
                if k not in lc:
                    raise ValueError(f"Missing key {k} in label component {lc}")
                if mode == "classification":
                    assert isinstance(lc[k], int)
                else:
                    assert (isinstance(lc[k], (int, float)) and kcount == 1) or len(lc[k]) == kcount
        out = []
        for lc in label_components:
            bbox = lc["bbox"]
            for i in range(len(bbox)):
                bbox[i] = 0 if bbox[i] < 0 else (BOX_DIM if bbox[i] > BOX_DIM else bbox[i])
            vec = []
            for k, _, mode in BOX_PROPERTIES:
                item = lc[k]
                if isinstance(item, (list, tuple)):
                    vec += list(item)
                elif isinstance(item, (float, int)):
                    vec.append(item + SPECIAL_TOKENS if mode == "classification" else item)
                else:
                    raise ValueError(f"Invalid item {item} for key {k}")
            out.append(vec)
        return out

    def component_idx(self, key):
        idx = 0
        for k, kcount, mode in BOX_PROPERTIES:
            incr = kcount if mode == "regression" else 1
            if k == key:
                return idx, idx + incr
            idx += incr
        raise ValueError(f"Key {key} not found in properties")

    def get_box_property(self, key, add_special_tokens=True):
        for k, kcount, mode in BOX_PROPERTIES:
            if k == key:
                return k, kcount + (SPECIAL_TOKENS if mode == "classification" and add_special_tokens else 0), mode
        raise ValueError(f"Key {key} not found in properties")

    def component_idx_dict(self):
        return {k: self.component_idx(k) for k, _, _ in BOX_PROPERTIES}

    def convert_polygons_to_bboxes(self, label_components: List[Dict]):
        """:82-111: 4 corners -> (cx, cy, w, h, xskew, yskew) with the skews shifted by BOX_DIM // 2 into positive space."""
        for lc in label_components:
            poly = np.clip(lc["polygon"], 0, BOX_DIM)
            (x1, y1), (x2, y2), (x3, y3), (x4, y4) = poly
            cx, cy = (x1 + x2 + x3 + x4) / 4, (y1 + y2 + y3 + y4) / 4
            width, height = (x2 + x3) / 2 - (x1 + x4) / 2, (y3 + y4) / 2 - (y2 + y1) / 2
            x_skew = (x3 + x4) / 2 - (x1 + x2) / 2 + BOX_DIM // 2
            y_skew = (y2 + y3) / 2 - (y1 + y4) / 2 + BOX_DIM // 2
            lc["bbox"] = [cx, cy, width, height, x_skew, y_skew]
        return label_components

    def convert_bbox_to_polygon(self, box, skew_scaler=BOX_DIM // 2, skew_min=.001):
        """:113-145."""
        cx, cy, width, height = box[0], box[1], box[2], box[3]
        x1, y1, x2, y2 = cx - width / 2, cy - height / 2, cx + width / 2, cy + height / 2
        skew_x = math.floor((box[4] - skew_scaler) / 2)
        skew_y = math.floor((box[5] - skew_scaler) / 2)
        if abs(skew_x) < skew_min:
            skew_x = 0
        if abs(skew_y) < skew_min:
            skew_y = 0
        flat = [x1 - skew_x, y1 - skew_y, x2 - skew_x, y1 + skew_y, x2 + skew_x, y2 + skew_y, x1 + skew_x, y2 - skew_y]
        return [[flat[2 * i], flat[2 * i + 1]] for i in range(4)]
