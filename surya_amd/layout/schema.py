"""Output schema of the layout predictor (surya/layout/schema.py:1-17), field for field."""
from typing import Dict, List, Optional

from pydantic import BaseModel

from ..common.geometry import PolygonBox


class LayoutBox(PolygonBox):
    label: str
    position: int
    top_k: Optional[Dict[str, float]] = None


class LayoutResult(BaseModel):
    bboxes: List[LayoutBox]
    image_bbox: List[float]
    sliced: bool = False  # Whether the image was sliced and reconstructed
