"""TableRecPredictor: drop-in for surya.table_rec.TableRecPredictor (surya/table_rec/__init__.py:21-387) on the HIP Donut-Swin + ADETR
engine (csrc/layout_model.hip, SA_FAMILY_TABLE).

Two decoding passes per batch of table images, as in the reference: (1) prompt = the whole table -> rows and columns; (2) one prompt per
detected row (+ all the batch's columns as context) -> the row's spanning cells. The encoder runs once; the second pass re-batches the
decoder onto the encoder states by index (HipLayoutModel.select) where the reference stacks copies. The prompt's T tokens go through
HipLayoutModel.prefill in one pass (surya_layout_prefill), as the reference's first decoder call does. No CPU fallback."""
from __future__ import annotations

import os
from copy import deepcopy
from itertools import chain
from typing import List, Optional

import numpy as np
import torch
from PIL import Image

from ..common.geometry import PolygonBox
from ..common.predictor import BasePredictor, ModelLoader
from ..layout.model import FedRuns, HipLayoutModel
from ..settings import settings
from .config import BOX_DIM, BOX_PROPERTIES, CATEGORY_TO_ID, MAX_BOXES, MERGE_KEYS, MERGE_VALUES, SPECIAL_TOKENS, TableRecConfig, table_config
from .processor import TableRecProcessor
from .schema import TableCell, TableCol, TableResult, TableRow
from .shaper import LabelShaper

TABLE_REC_MAX_BOXES = MAX_BOXES                             # surya/settings.py:115
PROMPT_CAPACITY = 512                                       # decoder positions kept beyond TABLE_REC_MAX_BOXES for prompts with many columns


class TableRecModelLoader(ModelLoader):
    """checkpoint: None / config name (synthetic weights), {"config": TableRecConfig, "state_dict": {...}}, or a directory in the
    reference's on-disk format (surya/table_rec/loader.py:19-76): config.json with `encoder` / `decoder` sub-configs, *.safetensors
    with the reference's parameter names, preprocessor_config.json (SuryaEncoderImageProcessor: image_mean, image_std; the processor's
    size is overridden by TABLE_REC_IMAGE_SIZE = the encoder's image size, table_rec/processor.py:17-20)."""

    def __init__(self, checkpoint=None):
        super().__init__(checkpoint)
        ck = checkpoint
        self._mean = self._std = None
        if isinstance(ck, dict):
            self._cfg, self._sd = ck["config"], ck["state_dict"]
        elif isinstance(ck, str) and os.path.isdir(ck):
            from ..layout.config import read_checkpoint_dir
            from .config import table_config_from_reference_json
            raw, self._sd, pp = read_checkpoint_dir(ck)
            self._cfg = table_config_from_reference_json(raw)
            if pp:
                self._mean, self._std = pp.get("image_mean"), pp.get("image_std")
        else:
            from ..synth import make_table_weights
            self._cfg = table_config(ck if isinstance(ck, str) else "TABLE-DEFAULT")
            self._sd = make_table_weights(self._cfg, 0)

    def model(self, device=None, dtype=None, max_batch=None) -> HipLayoutModel:
        if device is None or device == "cuda":
            device = "cuda:0"
        return HipLayoutModel(self._cfg, self._sd, dtype=dtype or torch.bfloat16, device=device,
                              max_batch=max_batch or TableRecPredictor.default_batch_sizes["cuda"],
                              max_boxes=TABLE_REC_MAX_BOXES + PROMPT_CAPACITY)

    def processor(self, device=None, dtype=None) -> TableRecProcessor:
        h, w = self._cfg.encoder.image_size
        return TableRecProcessor({"height": h, "width": w}, image_mean=self._mean, image_std=self._std)


def split_property_logits(dcfg, cls: np.ndarray):
    """[B, sum of the non-bbox head widths] -> {property: [B, n]} (the head slot stacks category | merges | colspan | is_header)."""
    out, o = {}, 0
    for k, n in dcfg.head_widths():
        if k != "bbox":
            out[k] = cls[:, o:o + n]
            o += n
    return out


class TableRecPredictor(BasePredictor):
    model_loader_cls = TableRecModelLoader
    batch_size = None
    # "cuda": 128 rows per engine call instead of the reference's 32 (surya/table_rec/__init__.py:24-29): see LayoutPredictor
    default_batch_sizes = {"cpu": 8, "mps": 8, "cuda": 128, "xla": 16}

    # Multi-GPU: when set, ONE call's table crops are dealt over the ranks of the initialised process group in whole
    # batches (the reference's second pass makes a table's cells depend on the other tables of its batch) and the per-table results
    # all-gathered (common/predictor.sharded_over_ranks). Off by default, like DetectionPredictor.shard_pages.
    shard_pages: bool = settings.SURYA_AMD_SHARD
    process_group = None

    def __call__(self, images: List[Image.Image], batch_size: Optional[int] = None) -> List[TableResult]:
        if self.shard_pages:
            from ..common.predictor import sharded_over_ranks
            bs = min(batch_size or self.get_batch_size(), self.model.max_batch)       # whole batches per rank: a table's cells depend on its batch mates
            out = sharded_over_ranks(images, lambda mine: self.batch_table_recognition(mine, bs), self.model.device, self.process_group, chunk=bs)
            if out is not None:
                return out
        return self.batch_table_recognition(images, batch_size)

    # ------------------------------------------------------------------------------------------------------------ decoding
    def inference_loop(self, src_index: List[int], batch_input_ids: np.ndarray) -> List[List[dict]]:
        """surya/table_rec/__init__.py:35-131 for rows that cross-attend the encoded images src_index[i]. batch_input_ids: int
        [n, T, 10]. Returns per row the list of predicted box-property dicts (until the row's category is </S> / <PAD>)."""
        dcfg = self.model.config.decoder
        shaper = LabelShaper()
        n, T = batch_input_ids.shape[:2]
        assert n == len(src_index) and n <= self.model.max_batch
        if T > PROMPT_CAPACITY:
            raise ValueError(f"decoder prompt of {T} tokens exceeds the {PROMPT_CAPACITY} positions kept for prompts")
        self.model.select(src_index)
        self.model.set_feedback()
        predictions: List[List[dict]] = [[] for _ in range(n)]
        all_done = np.zeros(n, bool)
        position, token_count, step_tokens = 0, 0, T
        ids = batch_input_ids.astype(np.int32)
        runs = None
        while token_count < TABLE_REC_MAX_BOXES:
            if position == 0:                                # the prompt: one pass over its T tokens (the reference's prefill = True call)
                cls, box = self.model.prefill(ids)
                position = ids.shape[1]
                # the fed-back tokens: one model call per box as in the reference, served from device-fed runs (layout/model.FedRuns)
                runs = FedRuns(self.model, position, TABLE_REC_MAX_BOXES - T, settings.LAYOUT_STEPS_PER_SYNC)
            else:
                cls, box = runs.step(ids[:, 0])
                position += 1
            props = split_property_logits(dcfg, cls)
            category = props["category"].argmax(-1)
            done = (category == dcfg.eos_token_id) | (category == dcfg.pad_token_id)
            rows = []
            for j in range(n):
                bp = {}
                for k, _, mode in BOX_PROPERTIES:
                    if mode == "classification":
                        bp[k] = int(props[k][j].argmax(-1)) - SPECIAL_TOKENS
                    elif k == "bbox":
                        bp[k] = (box[j] * np.float32(BOX_DIM)).tolist()
                    else:                                    # colspan: round(clamp(x, min=1)) (:96-98), round half to even like torch.round
                        bp[k] = int(np.round(np.maximum(props[k][j], np.float32(1.0)))[0])
                rows.append(bp)
            all_done |= done
            if all_done.all():
                break
            nxt = np.array(shaper.dict_to_labels(rows), dtype=np.float64).astype(np.int64)       # clamps each bbox in place, then truncates
            for j in range(n):
                if not all_done[j]:
                    predictions[j].append(rows[j])
            token_count += step_tokens
            step_tokens = 1
            ids = nxt[:, None, :].astype(np.int32)
        return predictions

    def batch_table_recognition(self, images: List, batch_size=None) -> List[TableResult]:
        assert all(isinstance(image, Image.Image) for image in images)
        if batch_size is None:
            batch_size = self.get_batch_size()
        batch_size = min(batch_size, self.model.max_batch)
        if len(images) == 0:
            return []
        query_items = [{"polygon": [[0, 0], [im.width, 0], [im.width, im.height], [0, im.height]], "category": CATEGORY_TO_ID["Table"],
                        "colspan": 0, "merges": 0, "is_header": 0} for im in images]
        shaper = LabelShaper()
        results: List[TableResult] = []
        for i in range(0, len(images), batch_size):
            batch_images = [image.convert("RGB") for image in images[i:i + batch_size]]
            n = len(batch_images)
            orig_sizes = [image.size for image in batch_images]
            inputs = self.processor(images=batch_images, query_items=query_items[i:i + batch_size])
            self.model.encode_host(torch.from_numpy(np.stack(inputs["pixel_values"])))
            rowcol = self.inference_loop(list(range(n)), inputs["input_ids"])
            # second pass: one prompt per detected row, all the batch's columns as context (:190-230)
            row_queries, idx_map, columns = [], [], []
            for j, preds in enumerate(rowcol):
                for pr in preds:
                    item = {"polygon": shaper.convert_bbox_to_polygon(pr["bbox"]), "category": pr["category"], "colspan": 0, "merges": 0,
                            "is_header": int(pr["is_header"] == 1)}
                    if pr["category"] == CATEGORY_TO_ID["Table-row"]:
                        row_queries.append(item)
                        idx_map.append(j)
                    elif pr["category"] == CATEGORY_TO_ID["Table-column"]:
                        columns.append(item)
            cell_predictions: List[List[dict]] = []
            if row_queries:                                  # (the reference's torch.stack raises on a batch without rows)
                row_ids = self.processor(images=None, query_items=row_queries, columns=columns, convert_images=False)["input_ids"]
                for j in range(0, len(row_ids), batch_size):
                    cell_predictions.extend(self.inference_loop(idx_map[j:j + batch_size], row_ids[j:j + batch_size]))
            results.extend(self.decode_batch_predictions(rowcol, cell_predictions, orig_sizes, idx_map, shaper))
        return results

    # ------------------------------------------------------------------------------------------------------------ assembly
    def decode_batch_predictions(self, rowcol_predictions, cell_predictions, orig_sizes, idx_map, shaper) -> List[TableResult]:
        """:236-387: rows x columns -> a grid of cells, spanning cells from the second pass replace the grid cells they cover, vertical
        merges join cells of consecutive rows."""
        out = []
        for j, (preds, orig_size) in enumerate(zip(rowcol_predictions, orig_sizes)):
            cells_of_row = [c for i, c in enumerate(cell_predictions) if idx_map[i] == j]

            def to_image(bbox):
                return self.processor.resize_polygon(shaper.convert_bbox_to_polygon(bbox), (BOX_DIM, BOX_DIM), orig_size)

            columns = [TableCol(polygon=to_image(p["bbox"]), col_id=z, is_header=p["is_header"] == 1)
                       for z, p in enumerate(p for p in preds if p["category"] == CATEGORY_TO_ID["Table-column"])]
            rows, cells = [], []
            cell_id = 0
            for z, rp in enumerate(p for p in preds if p["category"] == CATEGORY_TO_ID["Table-row"]):
                row = TableRow(polygon=to_image(rp["bbox"]), row_id=z, is_header=rp["is_header"] == 1)
                rows.append(row)
                spanning = []
                for l, sc in enumerate(cells_of_row[z]):
                    polygon = to_image(sc["bbox"])
                    colspan = max(1, int(sc["colspan"]))
                    if colspan == 1 and sc["merges"] not in MERGE_VALUES:
                        continue                                 # a plain single cell: the grid provides it
                    if PolygonBox(polygon=polygon).height < row.height * .85:
                        continue                                 # must cover most of the row
                    spanning.append(TableCell(polygon=polygon, row_id=z, rowspan=1, cell_id=cell_id, within_row_id=l, colspan=colspan,
                                              merge_up=sc["merges"] in (MERGE_KEYS["merge_up"], MERGE_KEYS["merge_both"]),
                                              merge_down=sc["merges"] in (MERGE_KEYS["merge_down"], MERGE_KEYS["merge_both"]),
                                              is_header=row.is_header or z == 0))
                    cell_id += 1
                used, skip = set(), 0
                for l, col in enumerate(columns):
                    if skip:
                        skip -= 1
                        continue
                    cell_polygon = row.intersection_polygon(col)
                    added = False
                    for zz, sc in enumerate(spanning):
                        pct = PolygonBox(polygon=cell_polygon).intersection_pct(sc)
                        want_width = sum(c.width for c in columns[l:l + sc.colspan])
                        if pct > .9:
                            if sc.width > want_width * .85:
                                added = True
                                if zz not in used:
                                    used.add(zz)
                                    sc.col_id = l
                                    cells.append(sc)
                                    skip = sc.colspan - 1
                            else:
                                used.add(zz)
                    if not added:
                        cells.append(TableCell(polygon=cell_polygon, row_id=z, rowspan=1, cell_id=cell_id, within_row_id=l, colspan=1,
                                               merge_up=False, merge_down=False, col_id=l,
                                               is_header=row.is_header or col.is_header or z == 0))
                        cell_id += 1
            grid = deepcopy([[c for c in cells if c.row_id == row.row_id] for row in rows])
            for z, grid_row in enumerate(grid[1:]):
                above_row = grid[z]
                for l, cell in enumerate(grid_row):
                    if l >= len(above_row):
                        continue
                    above = above_row[l]
                    if above.merge_down and cell.merge_up and above.col_id == cell.col_id and above.colspan == cell.colspan:
                        above.merge(cell)
                        above.rowspan += cell.rowspan
                        grid_row[l] = above
            seen, merged = set(), []
            for cell in chain.from_iterable(grid):
                if cell.cell_id not in seen:
                    seen.add(cell.cell_id)
                    merged.append(cell)
            out.append(TableResult(cells=merged, unmerged_cells=cells, rows=rows, cols=columns, image_bbox=[0, 0, orig_size[0], orig_size[1]]))
        return out
