"""GPU: heat map -> boxes on the device (surya_det_boxes, csrc/det_post.h) against the host implementation
(surya_amd/detection/heatmap.py = the restatement of surya/detection/heatmap.py:14-107) on the same maps.

The geometry code is shared with the CPU harness (tests/test_det_post_cpu.py); what only exists on the device is checked here:
radix-select thresholds, union-find labelling with raster-order roots, atomic statistics, ordered compaction, row extremes.
Bar: same number of boxes in the same order, corners within 1e-3 px (float64 calipers on both sides, last-bit float32
differences only), confidences equal to 1e-7; full predictor results (after int() rescale / clean / expand) identical."""
import numpy as np
import pytest
import torch
from PIL import Image

from surya_amd.config import det_config
from surya_amd.detection import heatmap as hm
from surya_amd.settings import settings
from surya_amd.synth import make_det_weights, make_pages
from test_det_post_cpu import synth_map

pytestmark = pytest.mark.gpu


def _compare(post, maps, cap_err=1e-3):
    t = torch.from_numpy(np.stack(maps)).cuda().contiguous()
    got = post(t, 0.6, 0.35)
    total = 0
    for m, (boxes, conf) in zip(maps, got):
        ref_boxes, ref_conf = hm.detect_boxes(m, 0.6, 0.35)
        assert len(boxes) == len(ref_boxes), (len(boxes), len(ref_boxes))
        for b, rb in zip(boxes, ref_boxes):
            assert np.abs(b - np.asarray(rb, np.float32)).max() <= cap_err
        assert np.allclose(conf, np.asarray(ref_conf, np.float32), rtol=0, atol=1e-7)
        total += len(boxes)
    return total


@pytest.mark.parametrize("shape,seeds", [((160, 224), (0, 1, 2)), ((512, 512), (4, 5)), ((1024, 1024), (6, 7, 8, 9)), ((96, 400), (3,))])
def test_device_boxes_equal_host_on_synthetic_maps(hip_lib, shape, seeds):
    from surya_amd.detection.model import HipDetPost
    post = HipDetPost()
    n = _compare(post, [synth_map(*shape, s) for s in seeds])
    assert n > 0


def test_device_boxes_tall_page_and_tall_components(hip_lib):
    """A page taller than any LDS sizing by page height allowed (round 2 refused H >= 2560) with components of all three height
    classes of post_boxes_kernel: text-line shaped ones (16 KB LDS launch), a ~1500-row vertical rule (whole-LDS launch) and a
    ~3000-row one (workspace-scratch launch); then a normal page through the same HipDetPost (the dynamic-LDS attribute must
    follow the larger request)."""
    from surya_amd.detection.model import HipDetPost
    post = HipDetPost()
    h, w = 3200, 512
    m = synth_map(h, w, 41)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    for cx, cy, half in ((400, 1600, 1500.0), (120, 900, 750.0)):
        bar = np.exp(-np.maximum(np.abs(xx - cx) / 9.0, np.abs(yy - cy) / half) ** 4) * 0.9
        m = np.maximum(m, bar.astype(np.float32))
    m = np.ascontiguousarray(m)
    ref_boxes, _ = hm.detect_boxes(m, 0.6, 0.35)
    spans = sorted(np.ptp(np.asarray(b)[:, 1]) for b in ref_boxes)
    assert spans[-1] > 2700 and 1200 < spans[-2] < 2500                       # both tall components are there
    assert _compare(post, [m]) >= 4
    assert _compare(post, [synth_map(1024, 1024, 6)]) > 0


def test_device_boxes_text_like_pages(hip_lib):
    """Pages that look like text (surya_amd.synth.text_like_map: 30 / 100 / 300 line-shaped components per 1024^2 page) -- the
    shapes bench.py times surya_det_boxes on."""
    from surya_amd.detection.model import HipDetPost
    from surya_amd.synth import text_like_map
    post = HipDetPost()
    for n_lines in (30, 100, 300):
        maps = [text_like_map(1024, 1024, n_lines, seed=s) for s in (1, 2)]
        n = _compare(post, maps)
        assert n >= 1.6 * n_lines, (n_lines, n)


def test_device_boxes_more_components_than_max_boxes(hip_lib):
    """A page with more kept components than the fixed-size output holds is run again with a larger one (the reference has no cap)."""
    from surya_amd.detection.model import HipDetPost
    from surya_amd.synth import text_like_map
    post = HipDetPost(max_boxes=64)
    maps = [text_like_map(1024, 1024, 100, seed=3), text_like_map(1024, 1024, 30, seed=4)]
    assert _compare(post, maps) == 130


def test_device_boxes_noise_map_and_empty_page(hip_lib):
    """Hundreds of small components (noise around the threshold), a page with nothing above the threshold, repeatability."""
    from scipy.ndimage import convolve
    from surya_amd.detection.model import HipDetPost
    post = HipDetPost()
    rng = np.random.default_rng(9)
    noisy = convolve((rng.random((256, 256), dtype=np.float32) * 0.5 + 0.3), np.ones((3, 3), np.float32) / 9, mode="nearest")
    flat = np.full((256, 256), 0.05, np.float32)
    maps = [np.ascontiguousarray(noisy.astype(np.float32)), flat]
    _compare(post, maps)
    t = torch.from_numpy(np.stack(maps)).cuda().contiguous()
    a, b = post(t, 0.6, 0.35), post(t, 0.6, 0.35)
    assert len(a[1][0]) == 0
    assert all(np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) for x, y in zip(a, b))       # bit-identical run to run


def test_device_boxes_on_model_heatmaps_bench_pages(hip_lib):
    """DET-DEFAULT bf16 on the 16 pages of bench.py's detection leg: device post-processing of the [16, 2, 1024, 1024] maps
    (page stride = 2 planes) vs the host implementation on the same maps copied back."""
    from oracle.det_oracle import normalise_pages
    from surya_amd.detection.model import HipDetModel, HipDetPost
    cfg = det_config("DET-DEFAULT")
    m = HipDetModel(cfg, make_det_weights(cfg, 0), height=1024, width=1024, dtype=torch.bfloat16, max_batch=16)
    heat = m.forward(normalise_pages(make_pages(16, 1024, seed=1234)).cuda().contiguous())
    post = HipDetPost()
    got = post(heat, settings.DETECTOR_TEXT_THRESHOLD, settings.DETECTOR_BLANK_THRESHOLD)
    maps = heat[:, 0].cpu().numpy()
    n = 0
    for i in range(16):
        ref_boxes, ref_conf = hm.detect_boxes(maps[i], settings.DETECTOR_TEXT_THRESHOLD, settings.DETECTOR_BLANK_THRESHOLD)
        boxes, conf = got[i]
        assert len(boxes) == len(ref_boxes), (i, len(boxes), len(ref_boxes))
        for b, rb in zip(boxes, ref_boxes):
            assert np.abs(b - np.asarray(rb, np.float32)).max() <= 1e-3
        assert np.allclose(conf, np.asarray(ref_conf, np.float32), atol=1e-7)
        n += len(boxes)
    print(f"16 bench pages: {n} boxes, device == host")


def test_predictor_device_path_equals_host_path(hip_lib):
    """DetectionPredictor end to end (split, resize, model, post-processing, rescale, clean, expand): the device
    post-processing (default) and the host post-processing give the same TextDetectionResults, incl. a tall page that is
    split into strips and re-assembled."""
    from surya_amd.detection.predictor import DetectionPredictor
    cfg = det_config("DET-TINY")
    pred = DetectionPredictor(checkpoint={"config": cfg, "state_dict": make_det_weights(cfg, 0), "size": 256}, dtype=torch.float32)
    pages = [Image.fromarray(p) for p in make_pages(5, 256, seed=31)]
    settings.DETECTOR_IMAGE_CHUNK_HEIGHT = 300
    try:
        pages.append(Image.fromarray(np.vstack(make_pages(3, 256, seed=2))[:600]))        # 256 x 600 -> 3 strips
        pred.device_postprocess = True
        dev = pred(pages)
        pred.device_postprocess = False
        host = pred(pages)
    finally:
        settings.DETECTOR_IMAGE_CHUNK_HEIGHT = 1400
    assert len(dev) == len(host) == 6
    for d, h in zip(dev, host):
        assert d.image_bbox == h.image_bbox
        assert [b.polygon for b in d.bboxes] == [b.polygon for b in h.bboxes]
        assert np.allclose([b.confidence for b in d.bboxes], [b.confidence for b in h.bboxes], atol=1e-6)
    assert sum(len(d.bboxes) for d in dev) > 0
