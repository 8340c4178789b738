"""Result objects of TableRecPredictor, field for field what the reference's callers (marker, the surya_table CLI) read:

  TableCell / TableRow / TableCol / TableResult <- surya/table_rec/schema.py:8-48

All boxes are common.geometry.PolygonBox (4 corners, derived bbox / width / height, optional confidence)."""
from __future__ import annotations

from typing import List, Optional

from pydantic import BaseModel

from ..common.geometry import PolygonBox


class _TableBox(PolygonBox):
    is_header: bool


class TableRow(_TableBox):
    row_id: int
    label = property(lambda self: f"Row {self.row_id}")


class TableCol(_TableBox):
    col_id: int
    label = property(lambda self: f"Column {self.col_id}")


class TableCell(_TableBox):
    """A grid cell (row x column intersection) or a spanning cell from the second decoding pass; `merge_up` / `merge_down` are the
    decoder's vertical-merge votes, `rowspan` grows when TableRecPredictor.decode_batch_predictions acts on them; `text_lines` is
    filled by callers that run OCR on the table afterwards."""
    row_id: int
    cell_id: int
    within_row_id: int
    colspan: int
    rowspan: Optional[int] = None
    col_id: Optional[int] = None
    merge_up: bool = False
    merge_down: bool = False
    text_lines: Optional[List[dict]] = None
    label = property(lambda self: f"Cell {self.cell_id} {self.rowspan}/{self.colspan}")


class TableResult(BaseModel):
    cells: List[TableCell]                   # after vertical merges
    unmerged_cells: List[TableCell]
    rows: List[TableRow]
    cols: List[TableCol]
    image_bbox: List[float]
