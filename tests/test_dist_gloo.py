"""CPU, world_size 2 over gloo: the N>1 path (sharding, output all_gather, weight broadcast) without GPUs.
The contract checked is the one of SURVEY 8(e): outputs of 2 ranks == outputs of 1 rank for any sharding -- first on the
collective helpers alone, then on the PRODUCT code path: RecognitionPredictor.sharded_prediction_loop (the real device
loop, stop rules, shard deal and gather) driven by the scripted fake model of tests/test_scheduler_cpu.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from surya_amd import dist as sd


def _fake_line(i, max_tokens):
    rng = np.random.default_rng(1000 + i)
    L = int(rng.integers(0, max_tokens + 1))                  # includes empty lines
    toks = rng.integers(0, 70000, size=L).tolist()
    scores = rng.random(L).astype(np.float32).tolist()
    bb = np.zeros((max_tokens, 6), np.float32)
    bb[:L] = rng.integers(0, 1025, size=(L, 6))
    return toks, scores, bb


NS = (7, 16, 1)       # odd split, even split, fewer lines than ranks


def _worker(rank, world, port, max_tokens, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = []
    for n in NS:
        idx = sd.shard_indices(n, world, rank)
        lines = [_fake_line(i, max_tokens) for i in idx]
        bbs = np.stack([l[2] for l in lines]) if lines else np.zeros((0, max_tokens, 6), np.float32)
        out.append(sd.gather_line_outputs([l[0] for l in lines], [l[1] for l in lines], bbs, idx, n, max_tokens))
    w = [torch.full((5, 3), float(rank + 1)), torch.arange(7, dtype=torch.int32) * (rank + 1), torch.full((2,), 9.0 * (rank + 1))]
    sd.broadcast_tensors(w, src=0, bucket_bytes=32)           # tiny buckets: exercises the bucketing
    # start-up weight distribution: only rank 0 holds the repacked list, the others learn shapes from the manifest
    mine = [torch.arange(12, dtype=torch.float32).reshape(3, 4), torch.ones(5, dtype=torch.bfloat16) * 3] if rank == 0 else None
    shared = sd.share_weights(mine, "cpu", src=0, bucket_bytes=16)
    assert [tuple(t.shape) for t in shared] == [(3, 4), (5,)] and shared[1].dtype == torch.bfloat16
    assert shared[0].flatten().tolist() == list(range(12)) and shared[1].float().tolist() == [3.0] * 5
    # per-page objects (detection results) come back in global order on every rank
    objs = sd.gather_objects([{"page": i} for i in sd.shard_indices(5, world, rank)], sd.shard_indices(5, world, rank), 5)
    assert objs == [{"page": i} for i in range(5)]
    # ranks that were handed different inputs must fail loudly on EVERY rank, not hang in a mis-sized gather
    try:
        sd.assert_same_inputs([7, 1234 + rank, 0])
        raise AssertionError("fingerprint mismatch went unnoticed")
    except RuntimeError as e:
        assert "different inputs" in str(e)
    sd.assert_same_inputs([7, 1234, 0])
    q.put((rank, out, [t.tolist() for t in w]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one_rank():
    max_tokens = 12
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, max_tokens, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, out, w in results:
        for n, (tok, sc, bb) in zip(NS, out):
            ref = [_fake_line(i, max_tokens) for i in range(n)]
            assert tok == [r[0] for r in ref]
            for a, b in zip(sc, [r[1] for r in ref]):
                assert a == b                                  # fp32 scores travel bit-exactly
            assert np.array_equal(bb, np.stack([r[2] for r in ref]))
        assert w[0] == [[1.0] * 3] * 5 and w[1] == list(range(7)) and w[2] == [9.0, 9.0]     # rank 0's weights everywhere


def _run_world(world, max_tokens=12):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, max_tokens, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return results


@pytest.mark.parametrize("world", [5, 8])
def test_many_ranks_equal_one_rank(world):
    """World sizes the driver's node will run (8) and one that divides nothing (5): per_rank rounding with padding lines on the last
    ranks (NS = 7, 16, 1 lines over 5 / 8 ranks -> ranks with 0 lines), the fingerprint gather with > 2 entries, the bucketed
    broadcast with 4 / 7 receivers, the manifest broadcast, the object gather."""
    max_tokens = 12
    results = _run_world(world, max_tokens)
    assert sorted(r[0] for r in results) == list(range(world))
    for rank, out, w in results:
        for n, (tok, sc, bb) in zip(NS, out):
            ref = [_fake_line(i, max_tokens) for i in range(n)]
            assert tok == [r[0] for r in ref]
            assert sc == [r[1] for r in ref]
            assert np.array_equal(bb, np.stack([r[2] for r in ref]))
        assert w[0] == [[1.0] * 3] * 5 and w[1] == list(range(7)) and w[2] == [9.0, 9.0]


def test_packed_lines_behave_like_lists_and_gather_cost_is_bounded():
    """The gather's result is a PackedLines: list-of-lists semantics (index, slice, iterate, ==) without eager conversion; packing
    from the loop's dense arrays and unpacking 2048 lines must stay whole-array work (the per-line unpack cost 22 ms)."""
    import time
    rng = np.random.default_rng(3)
    n, T = 2048, 48
    lens = rng.integers(0, T + 1, n)
    toks = [rng.integers(0, 70000, L).tolist() for L in lens]
    scs = [rng.random(L).astype(np.float32).tolist() for L in lens]
    bb = np.zeros((n, T, 6), np.float32)
    for i, L in enumerate(lens):
        bb[i, :L] = rng.integers(0, 1025, (L, 6))
    pt, ps = sd.PackedLines.from_lists(toks, T + 1, np.int64), sd.PackedLines.from_lists(scs, T + 1, np.float32)   # the loop's shapes
    assert pt == toks and ps == scs and pt[3] == toks[3] and pt[-1] == toks[-1] and pt[5:9] == toks[5:9] and len(pt) == n
    assert [len(t) for t in pt] == lens.tolist() and np.array_equal(pt.row(7), np.asarray(toks[7]))
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        a, b, c = sd.gather_line_outputs(pt, ps, bb, list(range(n)), n, T)
        best = min(best, time.perf_counter() - t0)
    assert a == toks and b == scs and np.array_equal(c, bb)
    assert best < 0.02, f"pack + unpack of 2048 lines took {best * 1e3:.1f} ms"      # ~3 ms here; the per-line version took 25-30


def test_shard_is_a_partition():
    for n in (0, 1, 5, 256):
        for world in (1, 2, 4, 8):
            parts = [sd.shard_indices(n, world, r) for r in range(world)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


# ------------------------------------------------------------------ the product's sharded loop on two ranks
def _loop_worker(rank, world, port, n_lines, max_tokens, slots, q):
    import sys
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import test_scheduler_cpu as ts
    from surya_amd.recognition.predictor import RecognitionPrompt
    from surya_amd.settings import settings
    if world > 1:
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    settings.RECOGNITION_MAX_TOKENS = max_tokens
    pred, _ = ts.make(1, max_tokens, slots)
    pred.model.device = torch.device("cpu")
    pred.process_group = None

    def prepare_lines(flat, math_mode=True):
        """Stand-in for the host pre-processing (needs no tokenizer / GPU upload): line id = pixel value of the crop."""
        ids = [int(s[0, 0, 0]) for s in flat["slices"]]
        grids = [(2, 2 + 2 * (i % 3)) for i in ids]
        offs = np.cumsum([0] + [h * w for h, w in grids])
        tiles = np.zeros((offs[-1], 3), np.float32)
        for k, i in enumerate(ids):
            tiles[offs[k]:offs[k + 1], 0] = i
        return {"prompts": [RecognitionPrompt(k, "ocr_with_boxes", None, None, True) for k in range(len(ids))],
                "max_tokens": {k: max_tokens for k in range(len(ids))}, "tiles": tiles, "tile_offs": offs, "grids": grids,
                "prompt_ids": [[1000 + i, 5, 6] for i in ids]}

    pred.prepare_lines = prepare_lines
    widths = [40 + (i * 7) % 23 for i in range(n_lines)]
    flat = {"slices": [np.full((8, w, 3), i, np.float32) for i, w in enumerate(widths)], "input_text": [None] * n_lines,
            "task_names": ["ocr_with_boxes"] * n_lines}
    toks, boxes, scores = pred.sharded_prediction_loop(flat, slots, True)
    q.put((rank, toks, boxes.numpy().copy(), scores))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("n_lines,slots", [(23, 4), (1, 3)])
def test_sharded_prediction_loop_two_ranks_equal_one(n_lines, slots):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import test_scheduler_cpu as ts
    max_tokens = 12
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_loop_worker, args=(r, 2, port, n_lines, max_tokens, slots, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, toks, boxes, scores in results:
        assert len(toks) == n_lines and boxes.shape[0] == n_lines
        for i in range(n_lines):
            exp = ts.expected(i, max_tokens)
            assert toks[i] == exp, (rank, i, toks[i], exp)
            assert len(scores[i]) == len(exp)
            assert (boxes[i, :len(exp), 0] == i).all()            # every gathered box row belongs to the right line
    assert results[0][1] == results[1][1] and np.array_equal(results[0][2], results[1][2])


def _table_worker(rank, world, port, q):
    """TableRecPredictor.__call__ with shard_pages on the PRODUCT host code (both decoding passes, grid assembly), the model behind it
    being the CPU oracle with HipLayoutModel's interface (tests/table_util.py)."""
    import sys
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from PIL import Image
    from surya_amd.synth import make_table_weights
    from surya_amd.table_rec import predictor as tp
    from surya_amd.table_rec.config import table_config
    from surya_amd.table_rec.processor import TableRecProcessor
    from table_util import OracleTableModel
    torch.set_num_threads(4)
    if world > 1:
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = table_config("TABLE-TINY")
    sd_ = make_table_weights(cfg, 0)
    sd_["decoder.box_property_heads.category.weight"][5 + 1] *= 3.0
    sd_["decoder.box_property_heads.category.weight"][5 + 2] *= 2.5
    tp.TABLE_REC_MAX_BOXES = 8
    pred = object.__new__(tp.TableRecPredictor)
    pred.model = OracleTableModel(cfg, sd_, max_batch=4)
    pred.processor = TableRecProcessor({"height": 128, "width": 128})
    pred.shard_pages = world > 1
    pred.process_group = None
    rng = np.random.default_rng(3)
    pages = [Image.fromarray(rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)) for h, w in ((200, 320), (128, 128), (90, 400), (150, 150), (64, 300))]
    out = pred(pages, batch_size=2)
    q.put((rank, [r.model_dump() for r in out]))
    if world > 1:
        try:
            pred([pages[0]] if rank == 0 else [pages[1]], batch_size=2)          # different inputs per rank must fail loudly
            q.put((rank, "no error"))
        except Exception as e:
            q.put((rank, "raised " + type(e).__name__))
        dist.barrier()
        dist.destroy_process_group()


def test_sharded_table_predictor_two_ranks_equal_one():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_table_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(4)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    q1 = ctx.Queue()
    p1 = ctx.Process(target=_table_worker, args=(0, 1, port, q1))
    p1.start()
    _, single = q1.get(timeout=300)
    p1.join(timeout=60)
    results = [g for g in got if isinstance(g[1], list)]
    errors = [g for g in got if isinstance(g[1], str)]
    assert len(results) == 2 and all(r[1] == single for r in results)       # every rank holds all 5 tables, equal to the 1-rank run
    assert len(single) == 5 and sum(len(t["rows"]) for t in single) > 0
    assert len(errors) == 2 and all(e[1].startswith("raised") for e in errors), errors


def _forced_worker(port, q):
    """One rank, gloo, dist.force_collectives(True): the helpers must take the collective branch (bench.py --force-dist rehearses the
    N > 1 path this way on a 1-GPU box) and still return this rank's own data."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    assert not sd.collectives_on()
    sd.force_collectives(True)
    assert sd.collectives_on()
    max_tokens = 9
    lines = [_fake_line(i, max_tokens) for i in range(6)]
    tok, sc, bb = sd.gather_line_outputs([l[0] for l in lines], [l[1] for l in lines], np.stack([l[2] for l in lines]), list(range(6)), 6, max_tokens)
    assert tok == [l[0] for l in lines] and np.array_equal(bb, np.stack([l[2] for l in lines]))
    assert all(a == b for a, b in zip(sc, [l[1] for l in lines]))
    w = [torch.arange(6, dtype=torch.float32)]
    sd.broadcast_tensors(w, src=0)
    shared = sd.share_weights([torch.ones(3)], "cpu")
    assert shared[0].tolist() == [1.0, 1.0, 1.0] and w[0].tolist() == list(range(6))
    assert sd.gather_objects(["a", "b"], [0, 1], 2) == ["a", "b"]
    sd.assert_same_inputs([1, 2, 3])
    sd.force_collectives(False)
    q.put("ok")
    dist.destroy_process_group()


def test_forced_one_rank_group_takes_the_collective_branch():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_forced_worker, args=(port, q))
    p.start()
    assert q.get(timeout=120) == "ok"
    p.join(timeout=60)
    assert p.exitcode == 0


# ------------------------------------------------------------------ whole pages per rank (RecognitionPredictor.shard_pages)
def _page_worker(rank, world, port, q, n_pages=7, vals=None):
    """_call_page_sharded on `world` ranks: the orchestration (fingerprint, page deal, the rank's own single-rank call, the gather) with
    the single-rank call replaced by a stand-in that derives a page's OCRResult from its pixels."""
    import torch.distributed as dist
    from PIL import Image
    from surya_amd.recognition.predictor import RecognitionPredictor
    from surya_amd.recognition.schema import OCRResult, TextLine
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class Model:
        device = torch.device("cpu")

    class Det:
        shard_pages = True                   # must be switched off inside the rank's own call and restored afterwards

    calls = []
    pred = object.__new__(RecognitionPredictor)
    pred.model, pred.process_group = Model(), None
    pred.shard_pages, pred.shard_lines = True, False

    def fake_call(images, task_names, det_predictor, dbs, rbs, highres, bboxes, polygons, input_text, sort_lines, math_mode, return_words, drop):
        assert not pred.shard_pages and not pred.shard_lines and not det_predictor.shard_pages      # a plain single-rank call
        calls.append(len(images))
        out = []
        for im in images:
            v = int(np.asarray(im)[0, 0, 0])
            lines = [TextLine(text=f"page {v} line {k}", polygon=[[0, 0], [9, 0], [9, 9], [0, 9]], chars=[], confidence=0.5, words=[])
                     for k in range(v % 4)]
            out.append(OCRResult(text_lines=lines, image_bbox=[0, 0, im.size[0], im.size[1]]))
        return out if any(len(r.text_lines) for r in out) else []      # the real _call's contract: no line on any page -> [] (reference :928-929)

    real_call = RecognitionPredictor._call
    pred._call = lambda *a: real_call(pred, *a) if pred.shard_pages else fake_call(*a)     # the outer call is the product's, the rank's own call the stand-in
    pred.tasks = {"ocr_with_boxes": {}}
    vals = vals or [10 + i for i in range(n_pages)]
    pages = [Image.fromarray(np.full((20 + i, 30, 3), vals[i], np.uint8)) for i in range(n_pages)]
    det = Det()
    full = pred(pages, det_predictor=det)
    assert det.shard_pages and pred.shard_pages
    pred.gather_page_results = False
    part = pred(pages, det_predictor=det)
    summary = list(pred.last_page_summary)
    if part == []:
        part = [None] * n_pages
    try:
        pred(pages[:-1] if rank == 0 else pages, det_predictor=det)          # rank 0 was handed one page fewer
        err = "no error"
    except Exception as e:
        err = "raised " + type(e).__name__
    q.put((rank, [r.model_dump() for r in full], [None if r is None else r.model_dump() for r in part], summary, calls, err))
    dist.barrier()
    dist.destroy_process_group()


def _run_pages(world, n_pages, vals=None):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_page_worker, args=(r, world, port, q, n_pages, vals)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=120) for _ in range(world)), key=lambda g: g[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return got


def test_page_sharded_call_with_a_rank_of_blank_pages():
    """a rank whose pages hold no line gets [] from its own single-rank call (the reference's contract); those pages must come
    back as EMPTY OCRResults on every rank, not as None -- and a call whose pages are all blank returns [] like the single-rank call."""
    vals = [12, 11, 16, 13]                                  # lines per page = v % 4: rank 0 (pages 0, 2) sees 0, 0; rank 1 sees 3, 1
    for rank, full, part, summary, calls, err in _run_pages(2, 4, vals):
        assert [len(r["text_lines"]) for r in full] == [0, 3, 0, 1]
        assert [r["image_bbox"] for r in full] == [[0, 0, 30, 20 + i] for i in range(4)]
        assert [s_[0] for s_ in summary] == [0, 3, 0, 1]
        mine = list(range(rank, 4, 2))
        assert [i for i, r in enumerate(part) if r is not None] == mine and all(part[i] == full[i] for i in mine)
    for rank, full, part, summary, calls, err in _run_pages(2, 3, [8, 12, 16]):      # every page blank
        assert full == [] and all(p is None for p in part) and [s_[0] for s_ in summary] == [0, 0, 0]


@pytest.mark.parametrize("world,n_pages", [(2, 7), (3, 7), (3, 2), (4, 1)])      # the last two: ranks that are dealt no page at all
def test_page_sharded_call(world, n_pages):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_page_worker, args=(r, world, port, q, n_pages)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=120) for _ in range(world)), key=lambda g: g[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect_lines = [(10 + i) % 4 for i in range(n_pages)]
    for rank, full, part, summary, calls, err in got:
        assert [len(r["text_lines"]) for r in full] == expect_lines            # every rank holds every page, in page order
        assert full == got[0][1]
        mine = list(range(rank, n_pages, world))
        assert [i for i, r in enumerate(part) if r is not None] == mine         # partitioned: own pages only ...
        assert all(part[i] == full[i] for i in mine)
        assert [s_[0] for s_ in summary] == expect_lines                        # ... and a record of every page on every rank
        assert calls == ([len(mine), len(mine)] if mine else []) and err.startswith("raised")      # a rank without pages makes no call of its own
