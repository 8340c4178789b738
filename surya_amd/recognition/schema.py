"""Output schemas of RecognitionPredictor (fields of surya/recognition/schema.py:10-40)."""
from __future__ import annotations

import math
from typing import List, Optional

from pydantic import BaseModel, field_validator

from ..common.geometry import PolygonBox


class TaskNames:   # surya/common/surya/schema.py:1-11
    block_without_boxes = "block_without_boxes"
    ocr_with_boxes = "ocr_with_boxes"
    ocr_without_boxes = "ocr_without_boxes"


TASK_NAMES = [TaskNames.block_without_boxes, TaskNames.ocr_with_boxes, TaskNames.ocr_without_boxes]


class BaseChar(PolygonBox):
    text: str
    confidence: Optional[float] = 0

    @field_validator("confidence", mode="before")
    @classmethod
    def _nan_to_zero(cls, v):
        if v is None or (isinstance(v, float) and math.isnan(v)):
            return 0
        try:
            return 0 if math.isnan(float(v)) else v
        except (TypeError, ValueError):
            return v


class TextChar(BaseChar):
    bbox_valid: bool = True


class TextWord(BaseChar):
    bbox_valid: bool = True


class TextLine(BaseChar):
    chars: List[TextChar]
    original_text_good: bool = False
    words: List[TextWord] | None = None


class OCRResult(BaseModel):
    text_lines: List[TextLine]
    image_bbox: List[float]
