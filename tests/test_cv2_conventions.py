"""CPU: hand-derived vectors for the OpenCV conventions the host / device restatements rely on (cv2 itself is absent from
this image, so these pin the restatements to the PUBLISHED definitions, not to a cv2 binary -- DESIGN.md section 3 says so):

  cv2.dilate           dst(x, y) = max over kernel of src(x + x' - anchor.x, y + y' - anchor.y), anchor = (k // 2, k // 2):
                       an even kernel grows a blob towards bottom / right (modules/imgproc/src/morph docs)
  cv2.fillPoly         boundary pixels belong to the polygon (edges are drawn, interior by scanline)
  cv2.minAreaRect      minimum-area enclosing rectangle (rotating calipers over the convex hull); boxPoints = its 4 corners
  cv2.resize           half-pixel centres, BORDER_REPLICATE, INTER_CUBIC = Keys a = -0.75, INTER_LANCZOS4 = 8 taps of
                       sinc(x) sinc(x / 4), normalised; OpenCV's own closed form (interpolateLanczos4, cs[] table) is
                       restated below and must agree with imageops to 1e-12
  connectedComponents  connectivity 4, labels in raster order of first pixel
The same vectors run through the C++ core the HIP kernels share (tests/native harness) where it applies."""
import math

import numpy as np
import pytest

from surya_amd.common import imageops
from surya_amd.detection import heatmap as hm


def test_dilate_anchor_convention():
    m = np.zeros((9, 9), np.uint8)
    m[4, 4] = 1
    for k, lo, hi in ((1, 4, 4), (2, 4, 5), (3, 3, 5), (4, 3, 6), (5, 2, 6)):   # a source pixel lights [x - (k-1-k//2), x + k//2]
        d = hm.dilate_rect(m, k)
        ys, xs = np.nonzero(d)
        assert (xs.min(), xs.max(), ys.min(), ys.max()) == (lo, hi, lo, hi), (k, xs.min(), xs.max())
        assert d.sum() == k * k
    # out-of-image neighbours do not contribute and nothing wraps
    m2 = np.zeros((5, 5), np.uint8); m2[0, 0] = 1; m2[4, 4] = 1
    d = hm.dilate_rect(m2, 3)
    assert d[:2, :2].all() and d[3:, 3:].all() and d.sum() == 8


def test_fill_poly_boundary_is_inside():
    rect = imageops.fill_poly_mask(6, 7, [(1, 1), (4, 1), (4, 3), (1, 3)])
    assert rect.sum() == 12 and rect[1:4, 1:5].all()                       # both end rows / columns included
    tri = imageops.fill_poly_mask(6, 6, [(0, 0), (4, 0), (0, 4)])
    want = np.array([[1 if x + y <= 4 else 0 for x in range(6)] for y in range(6)], np.uint8)
    assert np.array_equal(tri, want)                                         # hypotenuse pixels x + y == 4 are drawn
    sliver = imageops.fill_poly_mask(4, 8, [(0, 1), (7, 1), (7, 1), (0, 1)])  # degenerate (zero area): still its boundary line
    assert sliver[1].all() and sliver.sum() == 8


def _rect_points(c0, u, v, n=40, seed=0):
    """integer points inside the rectangle c0 + a u + b v, a, b in [0, 1], corners included"""
    rng = np.random.default_rng(seed)
    pts = [c0, c0 + u, c0 + u + v, c0 + v]
    for _ in range(n):
        a, b = rng.random(2)
        p = np.rint(c0 + a * u + b * v)
        # keep only points that are inside the exact rectangle
        ra, rb = np.dot(p - c0, u) / np.dot(u, u), np.dot(p - c0, v) / np.dot(v, v)
        if 0 <= ra <= 1 and 0 <= rb <= 1:
            pts.append(p)
    return np.array(pts)


@pytest.mark.parametrize("c0,u,v", [((0, 0), (8, 6), (-3, 4)), ((10, 3), (12, 5), (-5, 12)), ((2, 2), (7, 0), (0, 3)),
                                    ((5, 5), (3, 4), (-8, 6))])
def test_min_area_rect_recovers_rotated_rectangle(c0, u, v):
    """Pythagorean edge vectors -> rectangles with integer corners: the minimum-area rectangle of its lattice points is the
    rectangle itself, corner for corner."""
    c0, u, v = np.array(c0, float), np.array(u, float), np.array(v, float)
    assert np.dot(u, v) == 0
    pts = _rect_points(c0, u, v)
    box = hm.min_area_rect_points(pts.astype(np.int64))
    want = np.array([c0, c0 + u, c0 + u + v, c0 + v])
    for w in want:
        assert np.abs(box - w).sum(1).min() < 1e-4, (w, box)
    area = np.linalg.norm(box[0] - box[1]) * np.linalg.norm(box[1] - box[2])
    assert abs(area - np.linalg.norm(u) * np.linalg.norm(v)) < 1e-3


def test_min_area_rect_diamond_and_collinear():
    diamond = np.array([(2, 0), (4, 2), (2, 4), (0, 2), (2, 2)])
    box = hm.min_area_rect_points(diamond)
    for w in diamond[:4]:
        assert np.abs(box - w).sum(1).min() < 1e-5          # area 8, not the upright 16
    line = np.array([(1, 1), (5, 1), (3, 1)])
    b = hm.min_area_rect_points(line)
    assert set(map(tuple, b.tolist())) == {(1.0, 1.0), (5.0, 1.0)}


def test_cubic_and_lanczos_weights_published_values():
    w = imageops._cubic_weights(np.array([0.5]))[0]
    assert np.allclose(w, [-0.09375, 0.59375, 0.59375, -0.09375], atol=1e-15)          # Keys kernel, a = -0.75
    assert np.allclose(imageops._cubic_weights(np.array([0.0]))[0], [0, 1, 0, 0], atol=1e-15)
    assert np.allclose(imageops._lanczos4_weights(np.array([0.0]))[0], [0, 0, 0, 1, 0, 0, 0, 0], atol=1e-15)
    for t in (0.1, 0.25, 0.5, 0.9):
        assert abs(imageops._cubic_weights(np.array([t]))[0].sum() - 1) < 1e-15
        assert abs(imageops._lanczos4_weights(np.array([t]))[0].sum() - 1) < 1e-15
    # OpenCV's interpolateLanczos4 (imgproc/src/imgwarp.cpp) computes the same taps through a rotation table
    s45 = 0.70710678118654752440084436210485
    cs = [(1, 0), (-s45, -s45), (0, 1), (s45, -s45), (-1, 0), (s45, s45), (0, -1), (-s45, s45)]
    for x in (0.13, 0.5, 0.77):
        y0 = -(x + 3) * math.pi * 0.25
        s0, c0 = math.sin(y0), math.cos(y0)
        co = []
        for i in range(8):
            y = -(x + 3 - i) * math.pi * 0.25
            co.append((cs[i][0] * s0 + cs[i][1] * c0) / (y * y))
        co = np.array(co) / sum(co)
        assert np.allclose(co, imageops._lanczos4_weights(np.array([x]))[0], atol=1e-12)


def test_resize_geometry_half_pixel_centres_and_replicate_border():
    row = np.array([[[0.0], [1.0]]], np.float32)                              # 1 x 2 image, one channel
    out = imageops.resize(np.repeat(row, 3, axis=2), 4, 1, "cubic")[0, :, 0]
    # src = (i + 0.5) / 2 - 0.5 = -0.25, 0.25, 0.75, 1.25; taps at floor(src) - 1 .. + 2, indices clamped to [0, 1]
    def cubic_at(s):
        b = math.floor(s); t = s - b
        w = imageops._cubic_weights(np.array([t]))[0]
        return sum(w[k] * float(np.clip(b - 1 + k, 0, 1)) for k in range(4))
    assert np.allclose(out, [cubic_at(-0.25), cubic_at(0.25), cubic_at(0.75), cubic_at(1.25)], atol=1e-7)
    assert out[0] < 0 < out[1] < 0.5 < out[2] < 1 < out[3]                   # cubic overshoot at both ends, symmetric
    assert abs(out[0] + out[3] - 1) < 1e-7 and abs(out[1] + out[2] - 1) < 1e-7


def test_components_are_4_connected_and_raster_ordered():
    heat = np.full((8, 12), 0.05, np.float32)
    heat[1:4, 1:4] = 0.9                 # A (first in raster order)
    heat[4, 4] = 0.9                     # touches A only diagonally: its own component under 4-connectivity (area 1 -> dropped)
    heat[1:5, 7:11] = 0.8                # B
    heat[6:8, 0:6] = 0.85                # C
    boxes, conf = hm.detect_boxes(heat, 0.6, 0.35)
    assert len(boxes) == 2               # A has 9 px (< 10, dropped), the diagonal pixel too; B (16 px) and C (12 px) stay
    assert boxes[0][:, 1].min() < boxes[1][:, 1].min()        # B before C: raster order of the first pixel
    assert abs(conf[0] - 0.8 / 0.85) < 1e-6 and abs(conf[1] - 1.0) < 1e-6


# ---- the one resampling pin this image offers: INTER_CUBIC vs torch's bicubic (same Keys a = -0.75 kernel, half-pixel centres,
# replicate border, no antialiasing). Every real line crop passes through this stage (the x28 round-up, processor/__init__.py:193-212).
CUBIC_GEOMETRIES = [((64, 512), (84, 532)), ((119, 238), (140, 252)), ((64, 333), (84, 336)), ((57, 411), (84, 420)), ((71, 129), (84, 140))]


@pytest.mark.parametrize("src,dst", CUBIC_GEOMETRIES)
def test_cubic_resize_matches_torch_bicubic(src, dst):
    import torch
    import torch.nn.functional as F
    from surya_amd.common import imageops
    rng = np.random.default_rng(src[1])
    img = rng.integers(0, 256, size=src + (3,)).astype(np.float32)
    ours = imageops.resize(img, dst[1], dst[0], "cubic")
    ref = F.interpolate(torch.from_numpy(img).permute(2, 0, 1)[None].double(), size=dst, mode="bicubic", align_corners=False)[0]
    ref = ref.permute(1, 2, 0).numpy()
    err = float(np.abs(ours - ref).max())
    assert err <= 2e-2, err              # on a 0-255 scale
