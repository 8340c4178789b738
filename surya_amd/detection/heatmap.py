"""Heat-map -> text boxes on the host (behaviour of surya/detection/heatmap.py:14-184 and surya/common/util.py:9-36).

CRAFT-style: dynamic thresholds from the top-10 % mean, 4-connected components of `heat > low_text`, per component a
rectangular dilation by sqrt(min(w, h)) + 1 and the minimum-area rectangle of the dilated pixels. OpenCV is absent from
this image, so connected components come from scipy.ndimage.label (raster-order labels like cv2), the dilation follows
cv2.dilate's definition (anchor = kernel centre, out-of-image = ignored) and the minimum-area rectangle is a rotating
calipers pass over the convex hull -- restatements of the published algorithms, NOT pinned against cv2. Box-index
parity is asserted between this post-processing fed with oracle vs HIP heat maps (tests/test_gpu_det.py).
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np
from PIL import Image
from pydantic import BaseModel
from scipy import ndimage

from ..common.geometry import PolygonBox
from ..settings import settings


class ColumnLine(PolygonBox):
    vertical: bool
    horizontal: bool


class TextDetectionResult(BaseModel):       # surya/detection/schema.py:12-17
    bboxes: List[PolygonBox]
    vertical_lines: List[ColumnLine]
    heatmap: Optional[object]
    affinity_map: Optional[object]
    image_bbox: List[float]


def get_dynamic_thresholds(linemap, text_threshold, low_text, typical_top10_avg=0.7):
    flat = linemap.ravel()
    k = int(len(flat) * 0.9)
    avg = np.mean(np.partition(flat, k)[k:])
    scale = np.clip(avg / typical_top10_avg, 0, 1) ** 0.5
    return np.clip(text_threshold * scale, 0.15, 0.8), np.clip(low_text * scale, 0.1, 0.6)


def dilate_rect(mask: np.ndarray, ksize: int) -> np.ndarray:
    """cv2.dilate(mask, getStructuringElement(MORPH_RECT, (k, k))): dst(y, x) = max over the k x k window anchored at its
    centre (k // 2); pixels outside the image do not contribute."""
    if ksize <= 1:
        return mask.copy()
    h, w = mask.shape
    a = ksize // 2
    out = np.zeros_like(mask)
    # window offsets d in [-a, ksize - 1 - a]: dst[y, x] |= src[y + dy, x + dx]
    rows = np.zeros_like(mask)
    for dx in range(-a, ksize - a):
        xs0, xs1 = max(0, dx), min(w, w + dx)
        rows[:, xs0 - dx: xs1 - dx] |= mask[:, xs0:xs1]
    for dy in range(-a, ksize - a):
        ys0, ys1 = max(0, dy), min(h, h + dy)
        out[ys0 - dy: ys1 - dy, :] |= rows[ys0:ys1, :]
    return out


def _convex_hull(pts: np.ndarray) -> np.ndarray:
    """Andrew monotone chain, counter-clockwise in (x, y)."""
    pts = np.unique(pts, axis=0)
    if len(pts) <= 2:
        return pts
    pts = pts[np.lexsort((pts[:, 1], pts[:, 0]))]

    def half(points):
        out = []
        for p in points:
            while len(out) >= 2 and ((out[-1][0] - out[-2][0]) * (p[1] - out[-2][1])
                                     - (out[-1][1] - out[-2][1]) * (p[0] - out[-2][0])) <= 0:
                out.pop()
            out.append(p)
        return out

    lower, upper = half(pts), half(pts[::-1])
    return np.array(lower[:-1] + upper[:-1], dtype=np.float64)


def min_area_rect_points(pts: np.ndarray) -> np.ndarray:
    """4 corners (float32) of the minimum-area rectangle enclosing integer points, like boxPoints(minAreaRect(pts))."""
    hull = _convex_hull(pts.astype(np.float64))
    if len(hull) < 3:
        x0, y0 = pts[:, 0].min(), pts[:, 1].min()
        x1, y1 = pts[:, 0].max(), pts[:, 1].max()
        return np.array([[x0, y0], [x1, y0], [x1, y1], [x0, y1]], np.float32)
    best, best_area = None, np.inf
    for i in range(len(hull)):
        e = hull[(i + 1) % len(hull)] - hull[i]
        n = np.hypot(*e)
        if n == 0:
            continue
        u = e / n
        v = np.array([-u[1], u[0]])
        pu, pv = hull @ u, hull @ v
        area = (pu.max() - pu.min()) * (pv.max() - pv.min())
        if area < best_area - 1e-9:
            best_area = area
            best = (u, v, pu.min(), pu.max(), pv.min(), pv.max())
    u, v, u0, u1, v0, v1 = best
    return np.array([u * u0 + v * v0, u * u1 + v * v0, u * u1 + v * v1, u * u0 + v * v1], np.float32)


def detect_boxes(linemap: np.ndarray, text_threshold: float, low_text: float):
    img_h, img_w = linemap.shape
    text_threshold, low_text = get_dynamic_thresholds(linemap, text_threshold, low_text)
    labels, count = ndimage.label(linemap > low_text, structure=[[0, 1, 0], [1, 1, 1], [0, 1, 0]])   # 4-connectivity
    slices = ndimage.find_objects(labels)
    det, confidences, max_conf = [], [], 0
    for k in range(1, count + 1):
        sl = slices[k - 1]
        y, x = sl[0].start, sl[1].start
        h, w = sl[0].stop - y, sl[1].stop - x
        comp = labels[sl] == k
        if int(comp.sum()) < 10:
            continue
        niter = int(np.sqrt(min(w, h)))
        buffer = 1
        sx, sy = max(0, x - niter - buffer), max(0, y - niter - buffer)
        ex, ey = min(img_w, x + w + niter + buffer), min(img_h, y + h + niter + buffer)
        mask = labels[sy:ey, sx:ex] == k
        line_max = np.max(linemap[sy:ey, sx:ex][mask])
        if line_max < text_threshold:
            continue
        seg = dilate_rect(mask, buffer + niter)
        ys, xs = np.nonzero(seg)
        contour = np.column_stack((xs + sx, ys + sy))
        # the hull only needs each row's extreme pixels (same hull as the full pixel set, far fewer points)
        rows = np.unique(ys)
        first = np.searchsorted(ys, rows, side="left")
        last = np.searchsorted(ys, rows, side="right") - 1
        ext = np.concatenate([contour[first], contour[last]])
        box = min_area_rect_points(ext)
        bw, bh = np.linalg.norm(box[0] - box[1]), np.linalg.norm(box[1] - box[2])
        if abs(1 - max(bw, bh) / (min(bw, bh) + 1e-5)) <= 0.1:                       # near-square: use the upright box
            l, r, t, b = contour[:, 0].min(), contour[:, 0].max(), contour[:, 1].min(), contour[:, 1].max()
            box = np.array([[l, t], [r, t], [r, b], [l, b]], dtype=np.float32)
        # clockwise (screen coordinates) starting at the corner with the smallest x + y
        c = box.mean(0)
        ang = np.arctan2(box[:, 1] - c[1], box[:, 0] - c[0])
        box = box[np.argsort(ang)]                                                  # increasing angle = clockwise with y down
        box = np.roll(box, -int(box.sum(axis=1).argmin()), 0)
        max_conf = max(max_conf, line_max)
        confidences.append(line_max)
        det.append(box)
    if max_conf > 0:
        confidences = [c / max_conf for c in confidences]
    return det, confidences


def get_detected_boxes(textmap, text_threshold=None, low_text=None) -> List[PolygonBox]:
    text_threshold = settings.DETECTOR_TEXT_THRESHOLD if text_threshold is None else text_threshold
    low_text = settings.DETECTOR_BLANK_THRESHOLD if low_text is None else low_text
    if textmap.dtype != np.float32:
        textmap = textmap.astype(np.float32)
    boxes, confs = detect_boxes(textmap, text_threshold, low_text)
    return [PolygonBox(polygon=b, confidence=float(c)) for b, c in zip(boxes, confs)]


def clean_boxes(boxes: List[PolygonBox]) -> List[PolygonBox]:
    """Drop degenerate boxes and boxes fully contained in another one (surya/common/util.py:9-36). Same decisions as the reference's double
    loop; every box's bbox is formed once (the property rebuilds it from the polygon on each access: 26 evaluations per box and page inside
    the detector thread of the streamed call, while the decode loop waits for the interpreter)."""
    bbs = [bo.bbox for bo in boxes]
    kept = []
    for i, bo in enumerate(boxes):
        b = bbs[i]
        if b[2] == b[0] or b[3] == b[1]:
            continue
        inside = False
        for j, other in enumerate(boxes):
            o = bbs[j]
            if b == o:                                       # (covers `other.polygon == bo.polygon`: equal polygons have equal bboxes)
                continue
            if b[0] >= o[0] and b[1] >= o[1] and b[2] <= o[2] and b[3] <= o[3]:
                inside = True
                break
        if not inside:
            kept.append(bo)
    return kept


def get_and_clean_boxes(textmap, processor_size, image_size, text_threshold=None, low_text=None) -> List[PolygonBox]:
    bboxes = get_detected_boxes(textmap, text_threshold, low_text)
    for b in bboxes:
        b.rescale(processor_size, image_size)
        b.fit_to_bounds([0, 0, image_size[0], image_size[1]])
    return clean_boxes(bboxes)


def result_from_device_boxes(boxes: np.ndarray, confs: np.ndarray, processor_size, orig_sizes, heat_img=None,
                             aff_img=None) -> TextDetectionResult:
    """The host remainder of parallel_get_boxes (surya/detection/heatmap.py:160-184) when detect_boxes ran on the device
    (surya_det_boxes): PolygonBox construction, rescale / fit_to_bounds / clean_boxes (get_and_clean_boxes :127-136) and the
    y-expansion, on a few hundred 4-point boxes."""
    bboxes = [PolygonBox(polygon=b, confidence=float(c)) for b, c in zip(boxes, confs)]
    for b in bboxes:
        b.rescale(processor_size, orig_sizes)
        b.fit_to_bounds([0, 0, orig_sizes[0], orig_sizes[1]])
    bboxes = clean_boxes(bboxes)
    for box in bboxes:
        if box.height < 3 * box.width:
            box.expand(x_margin=0, y_margin=settings.DETECTOR_BOX_Y_EXPAND_MARGIN)
            box.fit_to_bounds([0, 0, orig_sizes[0], orig_sizes[1]])
    return TextDetectionResult(bboxes=bboxes, vertical_lines=[], heatmap=heat_img, affinity_map=aff_img,
                               image_bbox=[0, 0, orig_sizes[0], orig_sizes[1]])


def parallel_get_boxes(preds, orig_sizes, include_maps=False) -> TextDetectionResult:
    heatmap, affinity_map = preds
    heat_img = aff_img = None
    if include_maps:
        heat_img = Image.fromarray((heatmap * 255).astype(np.uint8))
        aff_img = Image.fromarray((affinity_map * 255).astype(np.uint8))
    bboxes = get_and_clean_boxes(heatmap, list(reversed(heatmap.shape)), orig_sizes)
    for box in bboxes:
        if box.height < 3 * box.width:                                              # not for vertical boxes
            box.expand(x_margin=0, y_margin=settings.DETECTOR_BOX_Y_EXPAND_MARGIN)
            box.fit_to_bounds([0, 0, orig_sizes[0], orig_sizes[1]])
    return TextDetectionResult(bboxes=bboxes, vertical_lines=[], heatmap=heat_img, affinity_map=aff_img,
                               image_bbox=[0, 0, orig_sizes[0], orig_sizes[1]])
