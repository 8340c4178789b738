"""Result schema of TableRecPredictor: defined next to the layout results (one model family, one schema module)."""
from ..layout.schema import TableCell, TableCol, TableResult, TableRow  # noqa: F401
