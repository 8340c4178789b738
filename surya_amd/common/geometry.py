"""PolygonBox: the base of every output schema (behaviour of surya/common/polygon.py:9-201).

Host-side glue, not accelerated; semantics (int truncation in rescale/expand, clamp order) are kept exactly so
callers that post-process results see identical numbers.
"""
from __future__ import annotations

import copy
import numbers
from typing import List, Optional

import numpy as np
from pydantic import BaseModel, computed_field, field_validator


def coerce_polygon(value) -> List[List[float]]:
    """[x0, y0, x1, y1], 4 corner points, or a (4, 2) array -> 4 corner points as floats (polygon.py:13-38)."""
    if isinstance(value, np.ndarray) and value.shape == (4, 2):
        return value.tolist()
    if isinstance(value, (list, tuple)) and len(value) == 4:
        if all(isinstance(v, numbers.Number) for v in value):
            x0, y0, x1, y1 = (float(v) for v in value)
            return [[x0, y0], [x1, y0], [x1, y1], [x0, y1]]
        if all(isinstance(p, (list, tuple)) and len(p) == 2 for p in value):
            return [[float(p[0]), float(p[1])] for p in value]
    raise ValueError(f"expected a bbox [x0,y0,x1,y1] or 4 corner points, got {value!r}")


class PolygonBox(BaseModel):
    polygon: List[List[float]]
    confidence: Optional[float] = None

    @field_validator("polygon", mode="before")
    @classmethod
    def _coerce(cls, value):
        """Accept [x0, y0, x1, y1], 4 corner points, or a (4, 2) array (polygon.py:13-38)."""
        return coerce_polygon(value)

    @computed_field
    @property
    def bbox(self) -> List[float]:
        xs = [p[0] for p in self.polygon]
        ys = [p[1] for p in self.polygon]
        return [min(xs), min(ys), max(xs), max(ys)]

    @property
    def height(self):
        b = self.bbox
        return b[3] - b[1]

    @property
    def width(self):
        b = self.bbox
        return b[2] - b[0]

    @property
    def area(self):
        return self.width * self.height

    @property
    def center(self):
        b = self.bbox
        return [(b[0] + b[2]) / 2, (b[1] + b[3]) / 2]

    def rescale(self, processor_size, image_size):
        """(w, h) of the processed page -> (w, h) of the image, truncating to int (polygon.py:59-69)."""
        sx = image_size[0] / processor_size[0]
        sy = image_size[1] / processor_size[1]
        for c in self.polygon:
            c[0] = int(c[0] * sx)
            c[1] = int(c[1] * sy)

    def round(self, divisor):
        for c in self.polygon:
            c[0] = int(c[0] / divisor) * divisor
            c[1] = int(c[1] / divisor) * divisor

    def fit_to_bounds(self, bounds):
        # (a fresh list of fresh points, like the reference's deepcopy -- without the generic copier: 2 calls per detected box)
        x0, y0, x1, y1 = bounds[0], bounds[1], bounds[2], bounds[3]
        self.polygon = [[max(min(c[0], x1), x0), max(min(c[1], y1), y0)] for c in self.polygon]

    def clamp(self, bbox: List[float]):
        for c in self.polygon:
            c[0] = max(min(c[0], bbox[2]), bbox[0])
            c[1] = max(min(c[1], bbox[3]), bbox[1])

    def shift(self, x_shift: float | None = None, y_shift: float | None = None):
        for c in self.polygon:
            if x_shift is not None:
                c[0] += x_shift
            if y_shift is not None:
                c[1] += y_shift

    def expand(self, x_margin: float, y_margin: float):
        """Grow by a fraction of width / height; corner order TL, TR, BR, BL (polygon.py:98-112)."""
        dx, dy = x_margin * self.width, y_margin * self.height
        sx, sy = (-1, 1, 1, -1), (-1, -1, 1, 1)
        self.polygon = [[int(p[0] + sx[i] * dx), int(p[1] + sy[i] * dy)] for i, p in enumerate(self.polygon)]

    def merge(self, other):
        a, b = self.bbox, other.bbox
        x0, y0, x1, y1 = min(a[0], b[0]), min(a[1], b[1]), max(a[2], b[2]), max(a[3], b[3])
        self.polygon = [[x0, y0], [x1, y0], [x1, y1], [x0, y1]]

    def merge_left(self, other):
        x0 = min(self.bbox[0], other.bbox[0])
        self.polygon[0][0] = x0
        self.polygon[3][0] = x0

    def merge_right(self, other):
        x1 = max(self.bbox[2], other.bbox[2])
        self.polygon[1][0] = x1
        self.polygon[2][0] = x1

    def x_overlap(self, other, x_margin=0):
        a, b = self.bbox, other.bbox
        return max(0, min(a[2] + x_margin, b[2] + x_margin) - max(a[0] - x_margin, b[0] - x_margin))

    def y_overlap(self, other, y_margin=0):
        a, b = self.bbox, other.bbox
        return max(0, min(a[3] + y_margin, b[3] + y_margin) - max(a[1] - y_margin, b[1] - y_margin))

    def intersection_area(self, other, x_margin=0, y_margin=0):
        return self.x_overlap(other, x_margin) * self.y_overlap(other, y_margin)

    def intersection_pct(self, other, x_margin=0, y_margin=0):
        assert 0 <= x_margin <= 1 and 0 <= y_margin <= 1
        if self.area == 0:
            return 0
        if x_margin:
            x_margin = int(min(self.width, other.width) * x_margin)
        if y_margin:
            y_margin = int(min(self.height, other.height) * y_margin)
        return self.intersection_area(other, x_margin, y_margin) / self.area

    def intersection_polygon(self, other) -> List[List[float]]:
        p, q = self.polygon, other.polygon
        return [[max(p[0][0], q[0][0]), max(p[0][1], q[0][1])], [min(p[1][0], q[1][0]), max(p[1][1], q[1][1])],
                [min(p[2][0], q[2][0]), min(p[2][1], q[2][1])], [max(p[3][0], q[3][0]), min(p[3][1], q[3][1])]]

    def distance(self, other):
        a, b = self.center, other.center
        return ((a[0] - b[0]) ** 2 + (a[1] - b[1]) ** 2) ** 0.5

    def __hash__(self):
        return hash(tuple(self.bbox))
