"""Result schema of TableRecPredictor (surya/table_rec/schema.py:8-48)."""
from __future__ import annotations

from typing import List, Optional

from pydantic import BaseModel

from ..common.geometry import PolygonBox


class TableCell(PolygonBox):
    row_id: int
    colspan: int
    within_row_id: int
    cell_id: int
    is_header: bool
    rowspan: Optional[int] = None
    merge_up: bool = False
    merge_down: bool = False
    col_id: Optional[int] = None
    text_lines: Optional[List[dict]] = None

    @property
    def label(self):
        return f"Cell {self.cell_id} {self.rowspan}/{self.colspan}"


class TableRow(PolygonBox):
    row_id: int
    is_header: bool

    @property
    def label(self):
        return f"Row {self.row_id}"


class TableCol(PolygonBox):
    col_id: int
    is_header: bool

    @property
    def label(self):
        return f"Column {self.col_id}"


class TableResult(BaseModel):
    cells: List[TableCell]
    unmerged_cells: List[TableCell]
    rows: List[TableRow]
    cols: List[TableCol]
    image_bbox: List[float]
