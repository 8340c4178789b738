"""Result objects of LayoutPredictor, field for field what the reference's callers (marker, the surya_layout CLI) read:

  LayoutBox / LayoutResult                      <- surya/layout/schema.py:8-17

(TableRecPredictor's are in table_rec/schema.py.) All boxes are common.geometry.PolygonBox (4 corners, derived bbox / width / height,
optional confidence)."""
from __future__ import annotations

from typing import Dict, List, Optional

from pydantic import BaseModel

from ..common.geometry import PolygonBox


# ------------------------------------------------------------------------------------------------------------------ layout
class LayoutBox(PolygonBox):
    """One region of a page: `label` is a name from layout.config.ID_TO_LABEL, `position` the reading-order index the decoder emitted it
    at, `top_k` the k most likely labels with their probabilities (the label's own probability is also `confidence`)."""
    label: str
    position: int
    top_k: Optional[Dict[str, float]] = None


class LayoutResult(BaseModel):
    bboxes: List[LayoutBox]
    image_bbox: List[float]                  # [0, 0, width, height] of the input page
    sliced: bool = False                     # the page was cut into slices (layout/slicer.py) and its boxes were shifted back
