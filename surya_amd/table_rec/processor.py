"""TableRecProcessor: decoder prompts + pixel values of the table-recognition model (surya/table_rec/processor.py:13-93).

A prompt is [bos x 10, query token, query_end x 10] (+ one token per detected column in the cell pass); the query is the table (first
pass) or one of its rows (second pass) as a LabelShaper token. Images go through the family's image processor (straight resize to the
model size, 1/255, mean = std = 0.5: layout/predictor.py LayoutImageProcessor)."""
from __future__ import annotations

from typing import List, Optional

import numpy as np

from ..layout.predictor import LayoutImageProcessor
from .config import BOX_DIM, SPECIAL_TOKENS
from .shaper import LabelShaper


class TableRecProcessor:
    token_pad_id = 0
    token_eos_id = 1
    token_bos_id = 1
    token_query_end_id = 4                                     # surya/table_rec/loader.py:70-75

    def __init__(self, max_size, image_mean=None, image_std=None):
        self.image_processor = LayoutImageProcessor(max_size, image_mean, image_std)
        self.box_size = (BOX_DIM, BOX_DIM)
        self.special_token_count = SPECIAL_TOKENS
        self.shaper = LabelShaper()

    def resize_polygon(self, polygon, orig_size, new_size):
        """:29-46: scales the corners IN PLACE and clamps them into [0, new_size]."""
        ws, hs = new_size[0] / orig_size[0], new_size[1] / orig_size[1]
        for corner in polygon:
            x, y = corner[0] * ws, corner[1] * hs
            corner[0] = 0 if x < 0 else (new_size[0] if x > new_size[0] else x)
            corner[1] = 0 if y < 0 else (new_size[1] if y > new_size[1] else y)
        return polygon

    def __call__(self, images: Optional[List], query_items: List[dict], columns: Optional[List[dict]] = None, convert_images: bool = True):
        if convert_images:
            assert len(images) == len(query_items) and len(images) > 0
            for image, q in zip(images, query_items):
                q["polygon"] = self.resize_polygon(q["polygon"], image.size, self.box_size)
        query_labels = self.shaper.dict_to_labels(self.shaper.convert_polygons_to_bboxes(query_items))
        width = len(query_labels[0])
        prompts = [[[self.token_bos_id] * width, label, [self.token_query_end_id] * width] for label in query_labels]
        if columns:                                            # every row's prompt ends with ALL the batch's columns (:77-81)
            column_labels = self.shaper.dict_to_labels(self.shaper.convert_polygons_to_bboxes(columns))
            for p in prompts:
                p += column_labels
        # torch.tensor(..., dtype=long) on mixed int / float lists truncates toward zero (:83)
        ids = np.array(prompts, dtype=np.float64).astype(np.int64)
        out = {"input_ids": ids, "attention_mask": np.ones_like(ids)}
        if convert_images:
            out["pixel_values"] = self.image_processor(images)["pixel_values"]
        return out
