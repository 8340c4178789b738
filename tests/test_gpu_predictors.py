"""GPU: the drop-in predictors end to end (host logic + HIP model) on synthetic pages / crops.

Checks call signatures, output schemas and -- for recognition -- that the predictor's continuous-batching
scheduler yields exactly the oracle's token stream for every line, for several batch sizes and steps-per-sync settings
(results must not depend on scheduling)."""
import numpy as np
import pytest
import torch
from PIL import Image

from oracle import rec_oracle as ro
from oracle import det_oracle as do
from surya_amd.config import rec_config, det_config
from surya_amd.settings import settings
from surya_amd.synth import make_rec_weights, make_det_weights, make_line_crops, make_pages

pytestmark = pytest.mark.gpu


def make_rec_predictor(cfg_name="REC-TINY", dtype=torch.float32, max_slots=8, max_tokens=16):
    from surya_amd.recognition.predictor import RecognitionPredictor, RecognitionModelLoader
    cfg = rec_config(cfg_name)
    sd = make_rec_weights(cfg, 0)

    class Loader(RecognitionModelLoader):
        def model(self, device=None, dtype_=None, **caps):
            return super().model("cuda:0", dtype, max_slots=max_slots, max_kv_len=512, max_patches=8192, max_prefill_tokens=2048)

    class Pred(RecognitionPredictor):
        model_loader_cls = Loader
        batch_size = max_slots

    settings.RECOGNITION_MAX_TOKENS = max_tokens
    return cfg, sd, Pred(checkpoint={"config": cfg, "state_dict": sd})


def oracle_tokens(cfg, sd, pred, crops, max_tokens):
    """Oracle on the predictor's own pre-processed prompts (same tiles / ids), one line at a time."""
    from surya_amd.recognition.schema import TaskNames
    flat = {"slices": crops, "input_text": [None] * len(crops), "task_names": [TaskNames.ocr_with_boxes] * len(crops)}
    prep = pred.prepare_lines(flat, math_mode=True)
    out = []
    tiles = prep["tiles"].cpu()
    for i in range(len(crops)):
        a, b = int(prep["tile_offs"][i]), int(prep["tile_offs"][i + 1])
        ids = torch.tensor([prep["prompt_ids"][i]])
        am = torch.ones_like(ids)
        pos = torch.arange(ids.shape[1])[None]
        om = ro.OracleRecModel(cfg, sd, cfg.image_token_id)
        t, _, _, _ = ro.generate(om, ids, tiles[a:b], [(1,) + tuple(prep["grids"][i])], am, pos, max_tokens, cfg.eos_token_id,
                                 cfg.pad_token_id, cfg.nop_token_id)
        out.append(t[0])
    return prep, out


@pytest.mark.parametrize("batch,steps_per_sync", [(8, 1), (3, 4), (5, 16)])
def test_scheduler_tokens_equal_oracle(hip_lib, batch, steps_per_sync):
    cfg, sd, pred = make_rec_predictor(max_slots=8, max_tokens=14)
    crops = [c.astype(np.float32) for c in make_line_crops(11, seed=3)]
    crops.sort(key=lambda c: -c.shape[1])
    prep, ref = oracle_tokens(cfg, sd, pred, crops, 14)
    settings.RECOGNITION_STEPS_PER_SYNC = steps_per_sync
    toks, boxes, scores = pred.generate(prep, batch)
    settings.RECOGNITION_STEPS_PER_SYNC = 4
    assert [list(t) for t in toks] == ref
    assert boxes.shape == (11, 14, 6) and all(len(s) == len(t) for s, t in zip(scores, toks))


def test_recognition_predictor_call_schema(hip_lib):
    from surya_amd.recognition.schema import OCRResult
    cfg, sd, pred = make_rec_predictor(max_slots=4, max_tokens=10)
    page = Image.fromarray(make_pages(1, 256, seed=5)[0])
    bboxes = [[[10, 20, 200, 52], [12, 80, 120, 110], [0, 0, 0, 0]]]           # the last one is degenerate
    out = pred([page], bboxes=bboxes, return_words=True, sort_lines=True)
    assert len(out) == 1 and isinstance(out[0], OCRResult) and out[0].image_bbox == [0, 0, 256, 256]
    assert len(out[0].text_lines) == 3
    for line in out[0].text_lines:
        d = line.model_dump()
        assert {"text", "polygon", "confidence", "chars", "words", "original_text_good", "bbox"} <= set(d)
        assert 0 <= (line.confidence or 0) <= 1
        for ch in line.chars:
            assert len(ch.polygon) == 4
    with pytest.raises(AssertionError):
        pred([page], task_names=["not_a_task"], bboxes=bboxes)
    assert pred([], bboxes=[]) == []


def test_detection_predictor_matches_oracle_boxes(hip_lib):
    """Boxes from our post-processing fed with HIP heat maps == fed with oracle heat maps (SURVEY 8(c) cv2 gap)."""
    from surya_amd.detection.predictor import DetectionPredictor, DetectionModelLoader
    from surya_amd.detection import heatmap as hm
    cfg = det_config("DET-TINY")
    sd = make_det_weights(cfg, 0)

    class Pred(DetectionPredictor):
        model_loader_cls = DetectionModelLoader
    pred = Pred(checkpoint={"config": cfg, "state_dict": sd, "size": 256}, dtype=torch.float32)
    pages = [Image.fromarray(p) for p in make_pages(3, 256, seed=11)]
    res = pred(pages, batch_size=2, include_maps=True)
    assert len(res) == 3
    x = do.normalise_pages([np.asarray(p) for p in pages])
    ref_maps = do.heatmaps(sd, cfg, x).numpy()
    for i, r in enumerate(res):
        assert r.image_bbox == [0, 0, 256, 256] and r.vertical_lines == [] and r.heatmap is not None
        ref = hm.parallel_get_boxes([ref_maps[i, 0], ref_maps[i, 1]], (256, 256))
        assert [b.polygon for b in r.bboxes] == [b.polygon for b in ref.bboxes]
        assert np.allclose([b.confidence for b in r.bboxes], [b.confidence for b in ref.bboxes], atol=1e-3)


def test_detection_tall_page_is_split(hip_lib):
    from surya_amd.detection.predictor import DetectionPredictor
    cfg = det_config("DET-TINY")
    sd = make_det_weights(cfg, 0)
    pred = DetectionPredictor(checkpoint={"config": cfg, "state_dict": sd, "size": 256}, dtype=torch.float32)
    settings.DETECTOR_IMAGE_CHUNK_HEIGHT = 300
    try:
        tall = Image.fromarray(np.vstack(make_pages(3, 256, seed=2))[:600])            # 256 wide, 600 tall -> 3 strips
        gen = list(pred.batch_detection([tall], batch_size=4))
        (preds, sizes), = gen
        assert sizes == [(256, 600)] and preds[0][0].shape == (600, 256) and len(preds[0]) == 2
    finally:
        settings.DETECTOR_IMAGE_CHUNK_HEIGHT = 1400


def test_end_to_end_detect_crop_recognise(hip_lib):
    """BASELINE.json configs[3] in miniature: RecognitionPredictor(images, det_predictor=...) -- detection boxes become
    polygon crops (slice_polys_from_image) that feed the recogniser; every detected line yields one TextLine."""
    from surya_amd.detection.predictor import DetectionPredictor
    cfg_d = det_config("DET-TINY")
    det = DetectionPredictor(checkpoint={"config": cfg_d, "state_dict": make_det_weights(cfg_d, 0), "size": 256}, dtype=torch.float32)
    cfg, sd, rec = make_rec_predictor(max_slots=8, max_tokens=6)
    pages = [Image.fromarray(p) for p in make_pages(2, 256, seed=21)]
    det_res = det(pages)
    out = rec(pages, det_predictor=det)
    assert len(out) == 2
    for r, d in zip(out, det_res):
        assert len(r.text_lines) == len(d.bboxes)
        for line, box in zip(r.text_lines, d.bboxes):
            assert line.polygon == box.polygon
    assert sum(len(r.text_lines) for r in out) > 0


def _det_with_drawn_rows(pages, rows, size, batch):
    """A detector whose text map is replaced by the rows drawn on each page (random weights find one blob per page), as
    bench.py's e2e leg does, so a call yields a realistic number of line boxes per page."""
    from surya_amd.detection.predictor import DetectionPredictor
    cfg_d = det_config("DET-TINY")
    masks = np.zeros((len(pages), size, size), np.uint8)
    for i, rr in enumerate(rows):
        for x0, y0, x1, y1 in rr:
            masks[i, y0 + 5:y1 - 5, x0 + 3:x1 - 3] = 1
    masks_d = torch.from_numpy(masks).to("cuda:0")
    page_of = {id(im): i for i, im in enumerate(pages)}

    class Det(DetectionPredictor):
        batch_size = batch

        def batch_heatmaps(self, images, batch_size=None):
            off = 0
            for heat, split_index, split_heights, sizes in super().batch_heatmaps(images, batch_size):
                n = split_index[-1] + 1
                idx = torch.tensor([page_of[id(im)] for im in images[off:off + n]], device=heat.device)
                heat[:, 0] = masks_d[idx].float() * 0.9 + 0.03
                off += n
                yield heat, split_index, split_heights, sizes

    return Det(checkpoint={"config": cfg_d, "state_dict": make_det_weights(cfg_d, 0), "size": size}, dtype=torch.float32)


@pytest.mark.parametrize("slots,det_batch", [(8, 2), (32, 3)])
def test_streamed_detect_recognise_equals_the_serial_call(hip_lib, slots, det_batch):
    """RecognitionPredictor(images, det_predictor=...) with the detector feeding the scheduler batch by batch
    (_call_streamed: lines admitted while later pages are still being detected) returns the OCRResults of the serial
    call -- detect everything, sort, recognise -- field for field; iter_detect's batches concatenate to __call__'s list."""
    from surya_amd.synth import make_pages_with_lines
    size = 256
    pages_np, rows = make_pages_with_lines(7, size, seed=99)
    pages = [Image.fromarray(p) for p in pages_np]
    det = _det_with_drawn_rows(pages, rows, size, det_batch)
    cfg, sd, rec = make_rec_predictor(max_slots=slots, max_tokens=7)
    whole = det(pages)
    parts = list(det.iter_detect(pages))
    assert [len(p) for p in parts] == [det_batch] * (7 // det_batch) + ([7 % det_batch] if 7 % det_batch else [])
    assert [b.polygon for r in whole for b in r.bboxes] == [b.polygon for p in parts for r in p for b in r.bboxes]
    assert sum(len(r.bboxes) for r in whole) > 3 * slots or slots > 8          # more lines than slots: refills happen mid-stream
    rec.stream_detection = False
    serial = rec(pages, det_predictor=det, return_words=True)
    assert "streamed" not in rec.last_timing
    rec.stream_detection = True
    streamed = rec(pages, det_predictor=det, return_words=True)
    assert rec.last_timing.get("streamed") == 1.0
    assert len(serial) == len(streamed) == 7
    for a, b, d in zip(serial, streamed, whole):
        assert len(a.text_lines) == len(d.bboxes)
        assert a.model_dump() == b.model_dump()
    # pages without any line (blank) in the middle of the stream, and a call that finds nothing at all
    blank = Image.fromarray(np.full((size, size, 3), 255, np.uint8))
    mixed = [pages[0], blank, blank, pages[1], blank]
    det2 = _det_with_drawn_rows(mixed, [rows[0], [], [], rows[1], []], size, 2)
    rec.stream_detection = False
    s2 = rec(mixed, det_predictor=det2)
    rec.stream_detection = True
    t2 = rec(mixed, det_predictor=det2)
    assert [r.model_dump() for r in s2] == [r.model_dump() for r in t2] and len(t2) == 5
    only_blank = [blank, blank, blank]
    det3 = _det_with_drawn_rows(only_blank, [[], [], []], size, 2)
    rec.stream_detection = False
    s3 = rec(only_blank, det_predictor=det3)
    rec.stream_detection = True
    assert rec(only_blank, det_predictor=det3) == s3


@pytest.mark.parametrize("cfg_name,dtype,n_lines,max_slots", [("REC-TINY", torch.bfloat16, 400, 512), ("REC-FULL", torch.bfloat16, 330, 384),
                                                                  ("REC-FULL", torch.bfloat16, 960, 1024)])
def test_ocr_results_identical_across_slot_counts(hip_lib, cfg_name, dtype, n_lines, max_slots):
    """the decode regime above 256 slots. One predictor with max_slots > 256, the same images + boxes through
    RecognitionPredictor.__call__ at recognition_batch_size = 64, 256 and max_slots: the OCRResults must be identical field for field -- a
    line's stream does not depend on how many lines decode beside it (split-K counts are a function of (N, K) only, every tile shape walks K
    in the same order), including the M > 256 tile choices of the gate|up and lm_head GEMMs. 330 / 384 rows take the 128 x 64 split-K tile, 960 / 1024
    rows the 128 x 128 split-K tile and the 8-phase 256 x 256 gate|up tile (csrc/gemm.h, sa::Tuning big_m_split / big_m_gateup)."""
    from surya_amd.recognition.predictor import RecognitionPredictor, RecognitionModelLoader
    cfg = rec_config(cfg_name)
    sd = make_rec_weights(cfg, 0, recipe="conditioned") if cfg_name == "REC-FULL" else make_rec_weights(cfg, 0)

    class Loader(RecognitionModelLoader):
        def model(self, device=None, dtype_=None, **caps):
            return super().model("cuda:0", dtype, max_slots=max_slots, max_kv_len=192, max_patches=max_slots * 260, max_prefill_tokens=max_slots * 72)

    class Pred(RecognitionPredictor):
        model_loader_cls = Loader
        batch_size = max_slots

    settings.RECOGNITION_MAX_TOKENS = 24
    try:
        pred = Pred(checkpoint={"config": cfg, "state_dict": sd})
        crops = make_line_crops(n_lines, seed=11)
        imgs = [Image.fromarray(c) for c in crops]
        boxes = [[[0, 0, im.size[0], im.size[1]]] for im in imgs]
        outs = {}
        for slots in (64, 256, max_slots):
            outs[slots] = [r.model_dump() for r in pred(imgs, bboxes=boxes, recognition_batch_size=slots)]
        assert len(outs[64]) == n_lines and sum(len(r["text_lines"]) for r in outs[64]) == n_lines
        assert outs[256] == outs[64]
        assert outs[max_slots] == outs[64]
        assert len({r["text_lines"][0]["text"] for r in outs[64]}) > n_lines // 4      # the streams are not degenerate
    finally:
        settings.RECOGNITION_MAX_TOKENS = None
