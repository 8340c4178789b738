"""cv2-free image primitives used on the host side of the hot path.

OpenCV is not in this image, so these are numpy restatements of the published algorithms:
  * resize_cubic / resize_lanczos4: separable resampling with OpenCV's geometry (half-pixel centres
    `src = (dst + 0.5) * scale - 0.5`, replicate border, no antialiasing -- what cv2.resize does for
    INTER_CUBIC (Keys a = -0.75) and INTER_LANCZOS4 (8 taps)).
  * fill_convex_poly_mask: even-odd scanline fill of an integer polygon, boundary pixels included
    (cv2.fillPoly convention).
They are NOT pinned against cv2 (absent); the parity boundary of the model path is downstream of them
(image_tiles / pixel_values), and both oracle and HIP path consume the same arrays.
"""
from __future__ import annotations

import numpy as np


def _cubic_weights(t: np.ndarray, a: float = -0.75) -> np.ndarray:
    """Keys cubic kernel weights for taps at offsets -1, 0, 1, 2 given fractional position t in [0, 1)."""
    w = np.empty(t.shape + (4,), np.float64)
    w[..., 0] = ((a * (t + 1) - 5 * a) * (t + 1) + 8 * a) * (t + 1) - 4 * a
    w[..., 1] = ((a + 2) * t - (a + 3)) * t * t + 1
    w[..., 2] = ((a + 2) * (1 - t) - (a + 3)) * (1 - t) * (1 - t) + 1
    w[..., 3] = 1.0 - w[..., 0] - w[..., 1] - w[..., 2]
    return w


def _lanczos4_weights(t: np.ndarray) -> np.ndarray:
    """8-tap Lanczos (a = 4) weights for taps at offsets -3..4, normalised to sum 1 (OpenCV interpolateLanczos4)."""
    offs = np.arange(-3, 5, dtype=np.float64)
    x = t[..., None] - offs                       # distance from each tap to the sample point
    with np.errstate(divide="ignore", invalid="ignore"):
        w = np.where(np.abs(x) < 1e-12, 1.0, np.sinc(x) * np.sinc(x / 4.0))
    w[np.abs(x) >= 4.0] = 0.0
    return w / w.sum(-1, keepdims=True)


def _resample_axis(img: np.ndarray, out_len: int, axis: int, kind: str) -> np.ndarray:
    in_len = img.shape[axis]
    if in_len == out_len:
        return img
    scale = in_len / out_len
    src = (np.arange(out_len, dtype=np.float64) + 0.5) * scale - 0.5
    base = np.floor(src)
    t = src - base
    if kind == "cubic":
        w, first = _cubic_weights(t), -1
    else:
        w, first = _lanczos4_weights(t), -3
    taps = w.shape[-1]
    idx = np.clip(base[:, None].astype(np.int64) + first + np.arange(taps)[None, :], 0, in_len - 1)   # replicate
    moved = np.moveaxis(img, axis, 0).astype(np.float64)
    gathered = moved[idx]                          # [out_len, taps, ...]
    wshape = (out_len, taps) + (1,) * (gathered.ndim - 2)
    out = (gathered * w.reshape(wshape)).sum(1)
    return np.moveaxis(out, 0, axis)


def resize(img: np.ndarray, new_w: int, new_h: int, kind: str) -> np.ndarray:
    """float image [H, W, C] -> [new_h, new_w, C]; kind in {'cubic', 'lanczos4'}; result float32."""
    out = _resample_axis(img, new_w, 1, kind)
    out = _resample_axis(out, new_h, 0, kind)
    return out.astype(np.float32)


def fill_poly_mask(h: int, w: int, pts) -> np.ndarray:
    """uint8 mask [h, w] = 1 inside / on the boundary of the polygon with integer vertices `pts` (x, y)."""
    pts = np.asarray(pts, np.float64)
    n = len(pts)
    mask = np.zeros((h, w), np.uint8)
    if n < 3 or h == 0 or w == 0:
        return mask
    ys = np.arange(h, dtype=np.float64)
    # boundary: rasterise each edge densely
    for i in range(n):
        x0, y0 = pts[i]
        x1, y1 = pts[(i + 1) % n]
        steps = int(max(abs(x1 - x0), abs(y1 - y0))) + 1
        xs = np.rint(np.linspace(x0, x1, steps + 1)).astype(np.int64)
        yy = np.rint(np.linspace(y0, y1, steps + 1)).astype(np.int64)
        ok = (xs >= 0) & (xs < w) & (yy >= 0) & (yy < h)
        mask[yy[ok], xs[ok]] = 1
    # interior: even-odd rule per scanline at pixel centres
    for y in range(h):
        xints = []
        for i in range(n):
            x0, y0 = pts[i]
            x1, y1 = pts[(i + 1) % n]
            if y0 == y1:
                continue
            lo, hi = (y0, y1) if y0 < y1 else (y1, y0)
            if lo <= ys[y] < hi:
                xints.append(x0 + (ys[y] - y0) * (x1 - x0) / (y1 - y0))
        xints.sort()
        for a, b in zip(xints[0::2], xints[1::2]):
            xa, xb = int(np.ceil(a)), int(np.floor(b))
            if xb >= xa:
                mask[y, max(xa, 0): min(xb, w - 1) + 1] = 1
    return mask


def page_pixels(image) -> np.ndarray:
    """uint8 [H, W, 3 or 4] pixels of a PIL RGB page. Pillow stores RGB as 4 bytes per pixel; np.asarray(image) repacks that to
    3 (~2-3 ms per 1024^2 page under the GIL: 128 pages cost more host time than their detection forward pass). When Pillow can
    export its memory through the Arrow C interface (>= 11.2, single-block images) and pyarrow is present, the page's OWN memory
    is returned as an RGBX view instead -- no copy; the device kernels read either stride."""
    try:
        import pyarrow as pa
        if image.mode == "RGB" and hasattr(image, "__arrow_c_array__"):
            a = pa.array(image)
            v = a.values.to_numpy(zero_copy_only=True)
            if v.dtype == np.uint8 and v.size == image.size[0] * image.size[1] * 4:
                return v.reshape(image.size[1], image.size[0], 4)
    except Exception:                                   # multi-block images, exotic builds: fall through to the copying path
        pass
    return np.ascontiguousarray(np.asarray(image, dtype=np.uint8))


_COPY_POOL = None


def copy_pool():
    """One small thread pool per process for staging copies and page views (thread start-up costs ~0.7 ms each: never per call)."""
    global _COPY_POOL
    if _COPY_POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        import os
        _COPY_POOL = ThreadPoolExecutor(max(1, min(8, os.cpu_count() or 1)), thread_name_prefix="surya-amd-copy")
    return _COPY_POOL


def parallel_copy(dsts, srcs, min_bytes: int = 4 << 20) -> None:
    """dsts[i][...] = srcs[i] for numpy arrays of equal shapes. Pages are 3-8 MB each and a call stages 16-128 of them into pinned
    memory: one memcpy stream moves ~10 GB/s, i.e. 6 ms per 16 RGBX pages at 1024^2 -- as long as the detector's forward pass takes for
    five of them. numpy releases the GIL inside a contiguous copy, so a few threads run them side by side."""
    total = sum(int(s_.nbytes) for s_ in srcs)
    if len(srcs) < 2 or total < min_bytes:
        for d, s_ in zip(dsts, srcs):
            d[...] = s_
        return

    def one(i):
        dsts[i][...] = srcs[i]

    list(copy_pool().map(one, range(len(srcs))))

