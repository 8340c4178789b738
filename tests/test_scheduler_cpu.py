"""CPU: the predictor's device loop (continuous batching + pipelined decode calls + look-ahead encoding) against a fake
model that implements the HipRecModel surface and ENFORCES the C-ABI contracts of include/surya_amd.h:

  * outputs of decode_async(n, ring) are only readable through wait_outputs(n, ring), each call exactly once, and a ring
    half is not reused before it was read;
  * at most two calls in flight; prefill only with nothing in flight;
  * encode_ahead refuses while earlier look-ahead images are unconsumed; prefill(tiles=None) consumes them in order;
  * a slot that is not active is never stepped; a finished line's extra steps are legal but their outputs must be ignored.

Every line has a scripted token stream (EOS at a scripted length, or endless), so the expected output is known exactly:
the reference's stop rules (recognition/__init__.py:583-595: eos / pad, max_tokens, detect_repeat_token) applied to it.
"""
from collections import deque
from types import SimpleNamespace

import numpy as np
import pytest

from surya_amd.recognition.postprocess import detect_repeat_token
from surya_amd.settings import settings

EOS, PAD, NOP = 1, 0, 3
H = 8                      # ring half = SA_MAX_STEPS / 2


def script(line, t):
    """Token t (0 = the prefill token) of line `line`: distinct values, EOS at a per-line length (some lines never stop)."""
    stop = [5, 0, 11, None, 2, 17, None, 9, 1, 30][line % 10]
    if stop is not None and t >= stop:
        return EOS
    if line % 10 == 6:
        return 100 + line + (t % 3)         # a 3-cycle: trips detect_repeat_token once 40 tokens are out
    return 100 + (line * 37 + t * 11) % 9000


class FakeModel:
    def __init__(self, max_slots, look_ahead_capacity=10 ** 9):
        self.max_slots = max_slots
        self.c = SimpleNamespace(max_prefill_tokens=10 ** 6, max_slots=max_slots)
        self.cfg = SimpleNamespace(encoder=SimpleNamespace(spatial_merge_size=2))
        self.slot_line, self.slot_pos = {}, {}
        self.active = []
        self.inflight = deque()            # (n, ring, outputs)
        self.ring_busy = [False, False]
        self.ahead = deque()
        self.grid_of = {}
        self.prefill_out = None
        self.stats = dict(decode_calls=0, steps=0, wasted_slot_steps=0, encode_ahead=0, prefills=0)

    # -- look-ahead encoder
    def encode_ahead(self, tiles, grid_hw):
        assert not self.ahead, "SA_ERR_STATE: previous look-ahead not consumed"
        assert tiles.shape[0] == sum(h * w for h, w in grid_hw)
        self.ahead.extend(int(tiles[off, 0]) for off in np.cumsum([0] + [h * w for h, w in grid_hw])[:-1])
        self.stats["encode_ahead"] += 1

    def discard_ahead(self):
        self.ahead.clear()
        self.stats["discards"] = self.stats.get("discards", 0) + 1

    def prefill(self, tiles, grid_hw, input_ids, slot_ids):
        assert not self.inflight, "prefill while decode calls are in flight"
        assert len(grid_hw) == len(input_ids) == len(slot_ids) > 0
        lines = [ids[0] - 1000 for ids in input_ids]                   # the test encodes the line id in the prompt
        if tiles is None:
            for ln in lines:
                assert self.ahead and self.ahead.popleft() == ln, "look-ahead images consumed out of order"
        else:
            assert tiles.shape[0] == sum(h * w for h, w in grid_hw)
        tok = np.zeros((1, self.max_slots), np.int32); sc = np.zeros((1, self.max_slots), np.float32)
        bb = np.zeros((1, self.max_slots, 6), np.int32)
        for ln, s in zip(lines, slot_ids):
            assert 0 <= s < self.max_slots and s not in self.active
            self.slot_line[s], self.slot_pos[s] = ln, 1
            tok[0, s], sc[0, s], bb[0, s] = script(ln, 0), 0.5, ln
        self.prefill_out = (tok, sc, bb)
        self.stats["prefills"] += 1

    def read_outputs(self, n):
        assert n == 1 and self.prefill_out is not None
        out, self.prefill_out = self.prefill_out, None
        return out

    def set_active(self, slots):
        assert len(set(slots)) == len(slots) and all(s in self.slot_line for s in slots)
        self.active = list(slots)

    def decode_async(self, n, ring):
        assert 1 <= n <= H and ring in (0, 1) and not self.ring_busy[ring], "ring half reused before it was read"
        assert len(self.inflight) < 2 and self.active
        tok = np.full((n, self.max_slots), -7, np.int32); sc = np.zeros((n, self.max_slots), np.float32)
        bb = np.zeros((n, self.max_slots, 6), np.int32)
        for k in range(n):
            for s in self.active:
                ln = self.slot_line[s]
                tok[k, s], sc[k, s], bb[k, s] = script(ln, self.slot_pos[s]), 0.25, ln
                self.slot_pos[s] += 1
        self.ring_busy[ring] = True
        self.inflight.append((n, ring, (tok, sc, bb)))
        self.stats["decode_calls"] += 1
        self.stats["steps"] += n

    def wait_outputs(self, n, ring):
        assert self.inflight and self.inflight[0][:2] == (n, ring), "outputs read out of order"
        _, _, out = self.inflight.popleft()
        self.ring_busy[ring] = False
        return out


def expected(line, max_tokens):
    toks = [script(line, 0)]
    if toks[0] in (EOS, NOP):
        return toks
    t = 1
    while True:
        toks.append(script(line, t)); t += 1
        if toks[-1] in (EOS, PAD) or len(toks) >= max_tokens or detect_repeat_token(toks):
            return toks


def make(n_lines, max_tokens, slots):
    from surya_amd.recognition.predictor import RecognitionPredictor, RecognitionPrompt
    pred = object.__new__(RecognitionPredictor)
    pred.prompt_queue, pred.batch_prompt_mapping = deque(), None
    pred.model = FakeModel(slots)
    pred.processor = SimpleNamespace(eos_token_id=EOS, pad_token_id=PAD, no_output_token=NOP)
    grids = [(2, 2 + 2 * (i % 3)) for i in range(n_lines)]
    offs = np.cumsum([0] + [h * w for h, w in grids])
    tiles = np.zeros((offs[-1], 3), np.float32)
    for i in range(n_lines):
        tiles[offs[i]:offs[i + 1], 0] = i                               # the fake reads the line id back from the tiles
    prep = {"prompts": [RecognitionPrompt(i, "ocr_with_boxes", None, None, True) for i in range(n_lines)],
            "max_tokens": {i: (max_tokens if i % 7 else max(1, max_tokens // 2)) for i in range(n_lines)},
            "tiles": tiles, "tile_offs": offs, "grids": grids, "prompt_ids": [[1000 + i, 5, 6] for i in range(n_lines)]}
    return pred, prep


@pytest.mark.parametrize("n_lines,max_tokens,slots,sps,ahead", [(23, 12, 4, 4, True), (23, 12, 4, 3, False), (9, 40, 16, 8, True),
                                                                 (40, 6, 7, 1, True), (5, 1, 2, 4, True), (64, 20, 8, 4, True),
                                                                 (30, 64, 8, 4, True), (27, 90, 5, 3, False)])
def test_device_loop_matches_scripted_streams(n_lines, max_tokens, slots, sps, ahead):
    old = (settings.RECOGNITION_STEPS_PER_SYNC, settings.RECOGNITION_ENCODE_AHEAD)
    settings.RECOGNITION_STEPS_PER_SYNC, settings.RECOGNITION_ENCODE_AHEAD = sps, ahead
    try:
        pred, prep = make(n_lines, max_tokens, slots)
        toks, boxes, scores = pred.generate(prep, slots)
    finally:
        settings.RECOGNITION_STEPS_PER_SYNC, settings.RECOGNITION_ENCODE_AHEAD = old
    m = pred.model
    assert not m.inflight and not m.ahead and m.prefill_out is None
    for i in range(n_lines):
        exp = expected(i, prep["max_tokens"][i])
        assert toks[i] == exp, (i, toks[i], exp)
        assert len(scores[i]) == len(exp)
        assert (boxes[i, :len(exp), 0].numpy() == i).all()             # every recorded box came from this line's slot
    assert (m.stats["encode_ahead"] > 0) == ahead


def test_no_speculative_call_when_budgets_are_known():
    """All lines run to max_tokens (no EOS): the loop issues exactly ceil((max_tokens - 1) / steps) calls per batch, the
    last one trimmed to the remaining budget -- no step beyond what some line needs."""
    old = (settings.RECOGNITION_STEPS_PER_SYNC, settings.RECOGNITION_ENCODE_AHEAD)
    settings.RECOGNITION_STEPS_PER_SYNC, settings.RECOGNITION_ENCODE_AHEAD = 4, True
    try:
        pred, prep = make(8, 11, 8)
        prep["prompt_ids"] = [[1000 + 3 + 10 * i, 5] for i in range(8)]      # lines 3, 13, 23, ...: scripted never to stop
        prep["max_tokens"] = {i: 11 for i in range(8)}
        offs = prep["tile_offs"]
        for i in range(8):
            prep["tiles"][offs[i]:offs[i + 1], 0] = 3 + 10 * i
        toks, _, _ = pred.generate(prep, 8)
    finally:
        settings.RECOGNITION_STEPS_PER_SYNC, settings.RECOGNITION_ENCODE_AHEAD = old
    assert all(len(t) == 11 for t in toks)
    assert pred.model.stats["steps"] == 10 and pred.model.stats["decode_calls"] == 3      # 4 + 4 + 2


def _chunks(prep, cuts):
    """Split one prepare_lines dict into consecutive dicts at the line ids in `cuts` (each with its OWN tile tensor and local
    tile offsets, the ids continuing), as the producer of a streamed detect -> recognise call hands them over."""
    offs, out = prep["tile_offs"], []
    edges = [0] + list(cuts) + [len(prep["prompts"])]
    for a, b in zip(edges[:-1], edges[1:]):
        out.append({"prompts": prep["prompts"][a:b], "max_tokens": {i: prep["max_tokens"][i] for i in range(a, b)},
                    "tiles": prep["tiles"][offs[a]:offs[b]].copy(), "tile_offs": offs[a:b + 1] - offs[a],
                    "grids": prep["grids"][a:b], "prompt_ids": prep["prompt_ids"][a:b]})
    return out


@pytest.mark.parametrize("n_lines,max_tokens,slots,sps,ahead,cuts,lag", [
    (23, 12, 4, 4, True, (5, 6, 17), 0), (23, 12, 4, 3, False, (5, 6, 17), 2), (40, 6, 7, 1, True, (1, 20, 39), 1),
    (64, 20, 8, 4, True, (16, 32, 48), 3), (30, 64, 8, 4, True, (29,), 50), (27, 90, 5, 3, False, (9, 9, 18), 4),
    (12, 9, 16, 4, True, (4, 8), 7)])
def test_fed_loop_equals_closed_list(n_lines, max_tokens, slots, sps, ahead, cuts, lag):
    """generate(feed=...): lines handed over in chunks WHILE the loop runs (a chunk becomes available `lag` polls after the
    previous one was taken; an empty chunk and a long drought -- the loop has to block -- are among the cases) give every line
    the stream of the closed list, under the fake model's C-ABI contract checks."""
    from surya_amd.recognition.predictor import FEED_END
    old = (settings.RECOGNITION_STEPS_PER_SYNC, settings.RECOGNITION_ENCODE_AHEAD)
    settings.RECOGNITION_STEPS_PER_SYNC, settings.RECOGNITION_ENCODE_AHEAD = sps, ahead
    try:
        pred, prep = make(n_lines, max_tokens, slots)
        pending = deque(_chunks(prep, cuts))
        state = {"wait": 0, "blocked": 0, "polls": 0}
        done, flushes = [], []

        def feed(block):
            state["polls"] += 1
            if not pending:
                return FEED_END
            if state["wait"] > 0 and not block:
                state["wait"] -= 1
                return None
            state["blocked"] += bool(block)
            state["wait"] = lag
            return pending.popleft()

        overall = max(prep["max_tokens"].values())
        toks, boxes, scores = pred.generate({"prompts": [], "max_tokens": {}, "overall_max_tokens": overall}, slots, feed=feed,
                                            on_done=lambda k, t, s, b: done.append((k, list(t), b.copy())),
                                            on_flush=lambda: flushes.append(len(done)))
    finally:
        settings.RECOGNITION_STEPS_PER_SYNC, settings.RECOGNITION_ENCODE_AHEAD = old
    m = pred.model
    assert not m.inflight and not m.ahead and m.prefill_out is None and not pending
    assert len(toks) == n_lines and boxes.shape[0] == n_lines
    for i in range(n_lines):
        exp = expected(i, prep["max_tokens"][i])
        assert toks[i] == exp, (i, toks[i], exp)
        assert len(scores[i]) == len(exp)
        assert (boxes[i, :len(exp), 0].numpy() == i).all()
    assert sorted(k for k, _, _ in done) == list(range(n_lines))           # on_done exactly once per line ...
    for k, t, b in done:
        assert t == toks[k] and (b[:len(t), 0] == k).all()                 # ... with the line's final stream and its own boxes
    if lag >= 50:
        assert state["blocked"] >= 1                                       # nothing left to run: the loop waited for the feed


def test_fed_loop_propagates_a_producer_failure():
    from surya_amd.recognition.predictor import FEED_END
    pred, prep = make(10, 8, 4)
    first, second = _chunks(prep, (4,))
    calls = {"n": 0}

    def feed(block):
        calls["n"] += 1
        if calls["n"] == 1:
            return first
        if not block:
            return None
        raise RuntimeError("detector failed")

    with pytest.raises(RuntimeError, match="detector failed"):
        pred.generate({"prompts": [], "max_tokens": {}, "overall_max_tokens": 8}, 4, feed=feed)


def test_fed_loop_rejects_ids_out_of_sequence_and_oversized_budgets():
    pred, prep = make(6, 8, 4)
    a, b = _chunks(prep, (3,))
    it = iter([b])
    with pytest.raises(AssertionError, match="continue the admitted ids"):
        pred.generate({"prompts": [], "max_tokens": {}, "overall_max_tokens": 8}, 4, feed=lambda block: next(it))
    pred, prep = make(6, 8, 4)
    it = iter([prep])
    with pytest.raises(AssertionError, match="overall_max_tokens"):
        pred.generate({"prompts": [], "max_tokens": {}, "overall_max_tokens": 4}, 4, feed=lambda block: next(it))


def test_streamed_call_is_only_taken_with_this_packages_unmodified_detector():
    """_can_stream: the detector must be this package's DetectionPredictor on its device post-processing path in one process, and
    nobody may have overridden its __call__ / _call / iter_detect (a subclass that edits the results there must keep seeing every
    page before the first crop is cut). Anything else -- incl. any callable with the reference's contract -- takes the serial path."""
    from surya_amd.detection.predictor import DetectionPredictor
    from surya_amd.recognition.predictor import RecognitionPredictor
    rec = object.__new__(RecognitionPredictor)
    rec.stream_detection, rec.device_preprocess, rec.shard_lines = True, True, False

    def det(cls=DetectionPredictor, post=True, shard=False):
        d = object.__new__(cls)
        d.device_postprocess, d.shard_pages = post, shard
        return d

    class OnlyModelHook(DetectionPredictor):
        def batch_heatmaps(self, images, batch_size=None):
            return super().batch_heatmaps(images, batch_size)

    class EditsResults(DetectionPredictor):
        def __call__(self, images, batch_size=None, include_maps=False):
            return super().__call__(images, batch_size, include_maps)[:1]

    assert rec._can_stream(det()) and rec._can_stream(det(OnlyModelHook))
    assert not rec._can_stream(det(EditsResults))
    assert not rec._can_stream(det(post=False)) and not rec._can_stream(det(shard=True))
    assert not rec._can_stream(lambda images, batch_size=None: [])
    for attr in ("stream_detection", "device_preprocess"):
        setattr(rec, attr, False)
        assert not rec._can_stream(det())
        setattr(rec, attr, True)
    rec.shard_lines = True
    assert not rec._can_stream(det())


def test_a_loop_that_ended_early_does_not_poison_the_next_one():
    """An exception between encode_ahead and the prefill that would have consumed it (here: the feed fails while look-ahead
    embeddings of queued lines are outstanding) leaves unconsumed images in the handle; the next generate() discards them first
    (surya_rec_encode_ahead with n_images = 0) instead of dying on SA_ERR_STATE."""
    old = (settings.RECOGNITION_STEPS_PER_SYNC, settings.RECOGNITION_ENCODE_AHEAD)
    settings.RECOGNITION_STEPS_PER_SYNC, settings.RECOGNITION_ENCODE_AHEAD = 4, True
    try:
        pred, prep = make(20, 12, 4)
        first, _ = _chunks(prep, (12,))
        state = {"n": 0}

        def feed(block):
            state["n"] += 1
            if state["n"] == 1:
                return first
            if state["n"] < 4:
                return None
            raise RuntimeError("producer died")

        with pytest.raises(RuntimeError, match="producer died"):
            pred.generate({"prompts": [], "max_tokens": {}, "overall_max_tokens": 12}, 4, feed=feed)
        m = pred.model
        assert m.ahead, "the scenario needs outstanding look-ahead images"
        m.inflight.clear(); m.ring_busy = [False, False]; m.active = []; m.prefill_out = None      # what surya_rec_* keep per call, not per loop
        pred2, prep2 = make(9, 10, 4)
        pred.prompt_queue.extend(prep2["prompts"])                           # stale queue content must not survive either
        toks, _, _ = pred.generate(prep2, 4)
        assert m.stats["discards"] >= 2 and not m.ahead
        for i in range(9):
            assert toks[i] == expected(i, prep2["max_tokens"][i])
    finally:
        settings.RECOGNITION_STEPS_PER_SYNC, settings.RECOGNITION_ENCODE_AHEAD = old


# ---------------------------------------------------------------------------------------------- the streamed call, without a GPU
def _key(poly):
    """A line's identity for the fake model: taken from its polygon, so the expected stream does not depend on admission order."""
    return (int(poly[0][1]) * 131 + int(poly[0][0])) % 997


class StreamFakeModel(FakeModel):
    """FakeModel whose scripted stream is chosen by the line KEY carried in the tiles / the prompt (not by the loop's line id)."""
    def __init__(self, max_slots):
        super().__init__(max_slots)
        self.cfg = SimpleNamespace(encoder=SimpleNamespace(spatial_merge_size=2), bbox_size=1024)
        self.device = "cpu"


def _streamed_predictor(slots, fail_after=None):
    from PIL import Image
    from surya_amd.recognition.predictor import RecognitionPredictor, TaskNames
    from surya_amd.recognition.schema import TextLine
    pred = object.__new__(RecognitionPredictor)
    pred.prompt_queue, pred.batch_prompt_mapping = deque(), None
    pred.model = StreamFakeModel(slots)
    pred.processor = SimpleNamespace(eos_token_id=EOS, pad_token_id=PAD, no_output_token=NOP)
    pred.last_timing = {}
    calls = {"prep": 0}

    def fake_preprocess(prompts, pages):
        calls["prep"] += 1
        if fail_after is not None and calls["prep"] > fail_after:
            raise RuntimeError("pre-processing failed in the producer")
        grids = [(2, 2 + 2 * (_key(p.image.poly) % 3)) for p in prompts]
        offs = np.cumsum([0] + [h * w for h, w in grids])
        tiles = np.zeros((offs[-1], 3), np.float32)
        for i, p in enumerate(prompts):
            assert 0 <= p.image.page < len(pages)                      # references are local to the chunk's own pages
            tiles[offs[i]:offs[i + 1], 0] = _key(p.image.poly)
        return tiles, offs, grids, [[1000 + _key(p.image.poly), 5, 6] for p in prompts]

    def fake_assemble(flat, items, drop_repeated_text, return_words, bbox_size):
        out = []
        for sorted_pos, orig, tokens, sc, rows in items:
            assert flat["slices"][sorted_pos].poly == tuple(tuple(pt) for pt in flat["polygons"][orig])   # id <-> original position
            out.append(TextLine(text=",".join(map(str, tokens)), polygon=flat["polygons"][orig], confidence=1.0, chars=[]))
        return out

    pred.preprocess_prompts_device = fake_preprocess
    pred._assemble_batch = fake_assemble
    return pred, calls, Image, TaskNames


def _fake_detector(boxes_per_page, batch):
    class Det:
        def iter_detect(self, images, batch_size=None):
            for a in range(0, len(images), batch):
                yield [SimpleNamespace(bboxes=[SimpleNamespace(polygon=pl) for pl in boxes_per_page[i]])
                       for i in range(a, min(a + batch, len(images)))]
    return Det()


def _page_polys(n_pages, seed):
    rng = np.random.default_rng(seed)
    out = []
    for pg in range(n_pages):
        n = int(rng.integers(0, 7)) if pg % 4 != 2 else 0               # some pages have no lines at all
        polys = []
        for k in range(n):
            x0, y0 = int(rng.integers(0, 40)), 10 + 17 * k + pg
            w = int(rng.integers(8, 60))
            polys.append([[x0, y0], [x0 + w, y0], [x0 + w, y0 + 9], [x0, y0 + 9]])
        out.append(polys)
    return out


@pytest.mark.parametrize("n_pages,det_batch,slots", [(9, 2, 4), (5, 5, 3), (12, 4, 32)])
def test_streamed_call_orchestration_on_fakes(n_pages, det_batch, slots):
    """_call_streamed with a fake detector and the contract-checking fake model: producer thread, queue, chunk-local page indices,
    width sort inside a chunk, admission while decoding, results put back by ORIGINAL position (pages without lines included)."""
    old = (settings.RECOGNITION_STEPS_PER_SYNC, settings.RECOGNITION_ENCODE_AHEAD, settings.RECOGNITION_MAX_TOKENS)
    settings.RECOGNITION_STEPS_PER_SYNC, settings.RECOGNITION_ENCODE_AHEAD, settings.RECOGNITION_MAX_TOKENS = 4, True, 12
    try:
        pred, calls, Image, TaskNames = _streamed_predictor(slots)
        polys = _page_polys(n_pages, seed=n_pages)
        images = [Image.new("RGB", (64, 160)) for _ in range(n_pages)]
        tasks = [TaskNames.ocr_with_boxes] * n_pages
        res = pred._call_streamed(images, tasks, _fake_detector(polys, det_batch), None, slots, [None] * n_pages, False, True, False,
                                  False, pred.last_timing, 0.0)
        if sum(len(p) for p in polys) == 0:
            assert res == []
            return
        assert len(res) == n_pages and pred.last_timing["streamed"] == 1.0
        for page, r in zip(polys, res):
            assert [ln.polygon for ln in r.text_lines] == [[[float(v) for v in pt] for pt in pl] for pl in page]
            for ln, pl in zip(r.text_lines, page):
                assert ln.text == ",".join(map(str, expected(_key(pl), 12)))
        assert calls["prep"] == sum(1 for a in range(0, n_pages, det_batch) if any(polys[a:a + det_batch]))
        assert not pred.model.inflight and not pred.model.ahead
    finally:
        settings.RECOGNITION_STEPS_PER_SYNC, settings.RECOGNITION_ENCODE_AHEAD, settings.RECOGNITION_MAX_TOKENS = old


def test_streamed_call_surfaces_a_producer_failure():
    old = (settings.RECOGNITION_ENCODE_AHEAD, settings.RECOGNITION_MAX_TOKENS)
    settings.RECOGNITION_ENCODE_AHEAD, settings.RECOGNITION_MAX_TOKENS = True, 12
    try:
        pred, calls, Image, TaskNames = _streamed_predictor(4, fail_after=1)
        polys = [[[[1, 10 + 20 * k], [30, 10 + 20 * k], [30, 19 + 20 * k], [1, 19 + 20 * k]] for k in range(3)] for _ in range(6)]
        images = [Image.new("RGB", (64, 160)) for _ in range(6)]
        with pytest.raises(RuntimeError, match="pre-processing failed in the producer"):
            pred._call_streamed(images, [TaskNames.ocr_with_boxes] * 6, _fake_detector(polys, 2), None, 4, [None] * 6, False, True, False,
                                False, pred.last_timing, 0.0)
    finally:
        settings.RECOGNITION_ENCODE_AHEAD, settings.RECOGNITION_MAX_TOKENS = old
