"""Golden vectors of the reference's HOST logic for the layout / table-recognition family, recorded from the real reference package in
the build container (oracle/ref_shim) so that they travel with the repo (tests/golden/family_host_reference.json, checked by
tests/test_oracle_golden.py anywhere):

  * table LabelShaper (surya/table_rec/shaper.py): polygons -> bbox tokens -> label vectors, bbox tokens -> polygons;
  * SuryaTableRecProcessor prompts (table_rec/processor.py:48-93) for row queries with and without columns;
  * TableRecPredictor.decode_batch_predictions (table_rec/__init__.py:236-387) on seeded synthetic tables (random grids, spanning
    cells that pass and fail the height / width rules, vertical merges, header rows);
  * layout prediction_to_polygon (surya/layout/util.py) and ImageSlicer geometry (surya/layout/slicer.py).

    python oracle/make_golden_family_host.py
"""
import copy
import importlib.util
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def query_items(rng, n):
    out = []
    for _ in range(n):
        x0, y0 = float(rng.uniform(-50, 900)), float(rng.uniform(-50, 900))
        w, h = float(rng.uniform(5, 500)), float(rng.uniform(5, 300))
        sk = rng.uniform(-8, 8, size=2).tolist()
        poly = [[x0 - sk[0], y0 - sk[1]], [x0 + w - sk[0], y0 + sk[1]], [x0 + w + sk[0], y0 + h + sk[1]], [x0 + sk[0], y0 + h - sk[1]]]
        out.append({"polygon": poly, "category": int(rng.integers(0, 5)), "colspan": int(rng.integers(0, 4)), "merges": int(rng.integers(0, 4)),
                    "is_header": int(rng.integers(0, 2))})
    return out


def synthetic_table(rng):
    """(rowcol predictions of one image, cell predictions per row, original size): a jittered grid + random second-pass cells."""
    nr, nc = int(rng.integers(2, 6)), int(rng.integers(2, 5))
    xs = np.sort(rng.choice(np.arange(40, 980, 20), size=nc + 1, replace=False)).tolist()
    ys = np.sort(rng.choice(np.arange(30, 990, 20), size=nr + 1, replace=False)).tolist()

    def bb(x0, y0, x1, y1, jitter=0.0):
        j = rng.uniform(-jitter, jitter, size=4) if jitter else np.zeros(4)
        return [float((x0 + x1) / 2 + j[0]), float((y0 + y1) / 2 + j[1]), float(x1 - x0 + j[2]), float(y1 - y0 + j[3]),
                float(512 + rng.integers(-3, 4)), float(512 + rng.integers(-3, 4))]

    rowcol = []
    for r in range(nr):
        rowcol.append({"bbox": bb(xs[0], ys[r], xs[-1], ys[r + 1], 3.0), "category": 1, "merges": 0, "colspan": 1, "is_header": int(r == 0 and rng.random() < 0.7)})
    for c in range(nc):
        rowcol.append({"bbox": bb(xs[c], ys[0], xs[c + 1], ys[-1], 3.0), "category": 2, "merges": 0, "colspan": 1, "is_header": int(rng.random() < 0.2)})
    rowcol.append({"bbox": bb(0, 0, 1024, 1024), "category": 4, "merges": 0, "colspan": 1, "is_header": 0})
    cells = []
    for r in range(nr):
        row_cells = []
        for _ in range(int(rng.integers(0, 4))):
            c0 = int(rng.integers(0, nc))
            span = int(rng.integers(1, min(3, nc - c0) + 1))
            tall = rng.random() < 0.8
            y1 = ys[r + 1] if tall else ys[r] + 0.5 * (ys[r + 1] - ys[r])
            narrow = rng.random() < 0.15
            x1 = xs[c0 + span] if not narrow else xs[c0] + 0.6 * (xs[c0 + span] - xs[c0])
            row_cells.append({"bbox": bb(xs[c0], ys[r], x1, y1, 2.0), "category": 3, "merges": int(rng.integers(0, 4)), "colspan": span,
                              "is_header": int(rng.integers(0, 2))})
        cells.append(row_cells)
    return rowcol, cells, (int(rng.integers(300, 2500)), int(rng.integers(300, 2500)))


def main():
    from types import SimpleNamespace
    import torch
    _, _, _, rshaper_mod = ref_shim.import_table_modules()
    spec = importlib.util.spec_from_file_location("ref_table_rec_pkg", os.path.join(ref_shim.REFERENCE_ROOT, "surya", "table_rec", "__init__.py"))
    rpkg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rpkg)
    rproc_mod = ref_shim.import_submodule("surya.table_rec.processor")
    rutil = ref_shim.import_submodule("surya.layout.util")
    rslicer = ref_shim.import_submodule("surya.layout.slicer")
    from PIL import Image
    rng = np.random.default_rng(20260921)
    ref = rshaper_mod.LabelShaper()
    out = {"shaper": {}, "processor": {}, "assembly": [], "layout": {}}
    items = query_items(rng, 40)
    conv = ref.convert_polygons_to_bboxes(copy.deepcopy(items))
    out["shaper"]["items"] = items
    out["shaper"]["bboxes"] = [[float(v) for v in c["bbox"]] for c in conv]
    out["shaper"]["labels"] = ref.dict_to_labels(copy.deepcopy(conv))
    boxes = rng.uniform(0, 1024, size=(60, 6)).tolist()
    out["shaper"]["box_to_polygon"] = [{"box": b, "polygon": ref.convert_bbox_to_polygon(list(b))} for b in boxes]
    out["shaper"]["component_idx"] = {k: list(v) for k, v in ref.component_idx_dict().items()}
    rp = object.__new__(rproc_mod.SuryaTableRecProcessor)
    rp.box_size, rp.special_token_count, rp.shaper = (1024, 1024), 5, ref
    rp.token_pad_id, rp.token_eos_id, rp.token_bos_id, rp.token_query_end_id = 0, 1, 1, 4
    rows_q, cols_q = query_items(rng, 6), query_items(rng, 4)
    out["processor"]["rows"] = rows_q
    out["processor"]["columns"] = cols_q
    out["processor"]["ids_with_columns"] = rp(images=None, query_items=copy.deepcopy(rows_q), columns=copy.deepcopy(cols_q), convert_images=False)["input_ids"].tolist()
    out["processor"]["ids_without_columns"] = rp(images=None, query_items=copy.deepcopy(rows_q), columns=None, convert_images=False)["input_ids"].tolist()
    sizes = [(300, 500), (128, 128), (700, 260)]
    q = [{"polygon": [[0, 0], [w, 0], [w, h], [0, h]], "category": 4, "colspan": 0, "merges": 0, "is_header": 0} for w, h in sizes]
    rp.image_processor = lambda images, *a, **k: {"pixel_values": []}
    imgs = [Image.new("RGB", s) for s in sizes]
    out["processor"]["image_sizes"] = sizes
    out["processor"]["ids_table_queries"] = rp(images=imgs, query_items=copy.deepcopy(q))["input_ids"].tolist()
    r_self = SimpleNamespace(processor=rp)
    for _ in range(12):
        rowcol, cells, size = synthetic_table(rng)
        res = rpkg.TableRecPredictor.decode_batch_predictions(r_self, [copy.deepcopy(rowcol)], copy.deepcopy(cells), [size], [0] * len(cells), ref)
        out["assembly"].append({"rowcol": rowcol, "cells": cells, "size": list(size), "expected": json.loads(json.dumps(res[0].model_dump()))})
    toks = rng.integers(0, 1025, size=(60, 7)).astype(np.float32)
    szs = [(int(rng.integers(50, 3000)), int(rng.integers(50, 3000))) for _ in range(60)]
    out["layout"]["prediction_to_polygon"] = [{"token": t.tolist(), "size": list(s), "polygon": rutil.prediction_to_polygon(torch.tensor(t), s, 1024, 512)}
                                              for t, s in zip(toks, szs)]
    sl = rslicer.ImageSlicer({"height": 1500, "width": 1500}, {"height": 1200, "width": 1200})
    geo = []
    for (w, h) in [(600, 800), (900, 1600), (5200, 1000), (1501, 1501), (1500, 1500), (1200, 4000), (3000, 3000)]:
        im = Image.new("RGB", (w, h))
        pieces, positions = sl.slice([im])
        geo.append({"size": [w, h], "count": sl.slice_count(im), "positions": [list(map(int, p)) if isinstance(p, (tuple, list)) else int(p) for p in positions],
                    "piece_sizes": [list(p.size) for p in pieces]})
    out["layout"]["slicer"] = geo
    path = os.path.join(GOLD, "family_host_reference.json")
    with open(path, "w", encoding="utf-8") as f:
        json.dump(out, f)
    print(path, os.path.getsize(path), "bytes;", sum(len(a["expected"]["cells"]) for a in out["assembly"]), "cells in the assembled tables")


if __name__ == "__main__":
    main()
