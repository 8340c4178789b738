"""GPU: Pillow's LANCZOS resizes on the device (surya_resample_lanczos_u8, csrc/resample.h) vs Pillow itself, bit for bit, and
DetectionPredictor with the device resize vs the host (Pillow) resize on pages that are not at the processor size."""
import numpy as np
import pytest
import torch
from PIL import Image

from surya_amd.common import pil_resample as pr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("w,h,size", [(1700, 2200, (1024, 1024)), (816, 1056, (1200, 1200)), (640, 480, (512, 512)),
                                      (3000, 500, (1024, 1024)), (333, 777, (512, 512)), (512, 300, (512, 512))])
@pytest.mark.parametrize("spix,dpix", [(3, 4), (4, 3), (4, 4)])
def test_device_chain_equals_pillow(hip_lib, w, h, size, spix, dpix):
    from surya_amd.detection.model import DeviceResampler
    rng = np.random.default_rng(w + h + spix)
    a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    im = Image.fromarray(a)
    im.thumbnail(size, Image.Resampling.LANCZOS)
    ref = np.asarray(im.resize(size, Image.Resampling.LANCZOS))
    src = a if spix == 3 else np.concatenate([a, rng.integers(0, 256, (h, w, 1), dtype=np.uint8)], 2)      # junk in X
    rs = DeviceResampler("cuda:0")
    cur = torch.from_numpy(np.ascontiguousarray(src)).cuda()
    steps = pr.plan(w, h, size)
    for i, tgt in enumerate(steps):
        out = torch.empty((tgt[1], tgt[0], dpix), dtype=torch.uint8, device="cuda") if i == len(steps) - 1 else None
        cur = rs.resize(cur, tgt, out=out)
    got = cur.cpu().numpy()
    assert got.shape == (size[1], size[0], dpix)
    assert np.array_equal(got[..., :3], ref)
    if dpix == 4:
        assert (got[..., 3] == 0).all()


def test_predictor_device_resize_equals_host_resize(hip_lib):
    from surya_amd.config import det_config
    from surya_amd.detection.predictor import DetectionPredictor
    from surya_amd.synth import make_det_weights, make_pages
    cfg = det_config("DET-TINY")
    pred = DetectionPredictor(checkpoint={"config": cfg, "state_dict": make_det_weights(cfg, 0), "size": 256})
    rng = np.random.default_rng(5)
    pages = []
    for (w, h) in [(300, 420), (256, 256), (500, 380), (200, 190), (640, 333), (256, 300)]:
        base = make_pages(1, 256, seed=int(rng.integers(1 << 30)))[0]
        pages.append(Image.fromarray(base).resize((w, h), Image.Resampling.BILINEAR))
    pred.device_resize = True
    dev = pred(pages)
    pred.device_resize = False
    host = pred(pages)
    assert len(dev) == len(host) == len(pages)
    for d, h_ in zip(dev, host):
        assert d.image_bbox == h_.image_bbox and len(d.bboxes) == len(h_.bboxes)
        for b1, b2 in zip(d.bboxes, h_.bboxes):
            assert b1.polygon == b2.polygon and b1.confidence == b2.confidence
