"""CPU: the restatement of Pillow's 8-bit LANCZOS resampler (surya_amd/common/pil_resample.py: sizes of thumbnail + resize,
fixed-point coefficient tables, two integer passes) against Pillow itself -- the library the reference calls
(surya/detection/__init__.py:50-57) -- bit for bit, on noise images (every rounding decision exercised)."""
import numpy as np
import pytest
from PIL import Image

from surya_amd.common import pil_resample as pr

CASES = [(1700, 2200, (1024, 1024)), (816, 1056, (1200, 1200)), (640, 480, (512, 512)), (3000, 500, (1024, 1024)),
         (333, 777, (512, 512)), (1023, 1025, (1024, 1024)), (2048, 1400, (1024, 1024)), (100, 90, (256, 256)), (512, 300, (512, 512))]


def pil_chain(a, size):
    im = Image.fromarray(a)
    im.thumbnail(size, Image.Resampling.LANCZOS)
    mid = im.size
    return np.asarray(im.resize(size, Image.Resampling.LANCZOS)), mid


@pytest.mark.parametrize("w,h,size", CASES)
def test_chain_equals_pillow(w, h, size):
    a = np.random.default_rng(w * 7 + h).integers(0, 256, (h, w, 3), dtype=np.uint8)
    ref, mid = pil_chain(a, size)
    steps = pr.plan(w, h, size)
    assert steps is not None
    if mid != (w, h):
        assert steps[0] == mid                     # thumbnail's aspect-preserving size
    cur = a
    for tw, th in steps:
        cur = pr.resample_reference(cur, tw, th)
    assert np.array_equal(cur, ref)


def test_paths_left_to_pillow():
    assert pr.plan(5000, 6000, (1024, 1024)) is None          # >= 4x shrink: thumbnail inserts a reduce() pre-pass
    assert pr.plan(10, 2000, (1024, 1024)) is None            # very tall image: Pillow resizes the axes in two calls
    assert pr.plan(1024, 1024, (1024, 1024)) == []            # already at the processor size
    assert pr.thumbnail_size(800, 600, (1024, 1024)) is None  # fits: thumbnail is a no-op, resize() then stretches


def test_coefficient_tables_sum_to_one():
    for n_in, n_out in [(2200, 1024), (791, 1024), (500, 171), (171, 1024)]:
        b, kk, ks = pr.lanczos_coeffs(n_in, n_out)
        assert kk.shape == (n_out, ks) and (b[:, 0] >= 0).all() and (b[:, 0] + b[:, 1] <= n_in).all()
        assert np.abs(kk.sum(1) - (1 << pr.PRECISION_BITS)).max() <= ks          # rounding of each tap only
