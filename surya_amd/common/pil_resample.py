"""Host side of the device LANCZOS resize of detection pages (csrc/resample.h, SURVEY 8(f) rank 2, detection side).

The reference resizes every page twice with Pillow -- `img.thumbnail(size, LANCZOS)` then `img.resize(size, LANCZOS)`
(surya/detection/__init__.py:50-57) -- on the host, ~10-20 ms per page. Pillow's 8-bit resampler (src/libImaging/Resample.c,
Pillow is an installed dependency, not part of the reference tree) is integer arithmetic once its coefficient tables exist:
    support = 3 * max(scale, 1); per output position the taps lanczos((x - center + 0.5) / max(scale, 1)) over
    [int(center - support + 0.5), int(center + support + 0.5)) clipped to the image, normalised to sum 1 in float64, then
    rounded to 22-bit fixed point; each pass accumulates int32 from 2^21 and stores clip8(acc >> 22); horizontal pass first,
    its uint8 result feeds the vertical pass.
This module computes the sizes and the fixed-point tables exactly as Pillow does (math.sin = the same libm); the kernels apply
them. Only what the GPU path covers is restated: RGB pages, no `reduce()` pre-pass (thumbnail's reducing_gap = 2.0 triggers
one when a side shrinks by 4x or more) -- `plan()` returns None for anything else and the caller keeps Pillow.
"""
from __future__ import annotations

import math
from functools import lru_cache
from typing import List, Optional, Tuple

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _lanczos(x: float) -> float:
    if -3.0 <= x < 3.0:
        if x == 0.0:
            return 1.0
        a, b = x * math.pi, x / 3.0 * math.pi
        return (math.sin(a) / a) * (math.sin(b) / b) if b != 0.0 else math.sin(a) / a
    return 0.0


@lru_cache(maxsize=256)
def lanczos_coeffs(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray, int]:
    """precompute_coeffs + normalize_coeffs_8bpc for the whole axis (box = (0, in_size)):
    bounds int32 [out, 2] = (first source index, tap count), taps int32 [out, ksize] in 22-bit fixed point."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 3.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    one = float(1 << PRECISION_BITS)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [_lanczos((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            kk[xx, x] = int(-0.5 + v * one) if v < 0 else int(0.5 + v * one)
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def thumbnail_size(w: int, h: int, size: Tuple[int, int]) -> Optional[Tuple[int, int]]:
    """Image.thumbnail's preserve_aspect_ratio: the size the first resize goes to, or None when the image already fits."""
    x, y = int(math.floor(size[0])), int(math.floor(size[1]))
    if x >= w and y >= h:
        return None
    aspect = w / h

    def round_aspect(number, key):
        return max(min(math.floor(number), math.ceil(number), key=key), 1)

    if x / y >= aspect:
        x = round_aspect(y * aspect, key=lambda n: abs(aspect - n / y))
    else:
        y = round_aspect(x / aspect, key=lambda n: 0 if n == 0 else abs(aspect - x / n))
    return x, y


def plan(w: int, h: int, size: Tuple[int, int]) -> Optional[List[Tuple[int, int]]]:
    """The chain of sizes `thumbnail(size, LANCZOS)` + `resize(size, LANCZOS)` passes through, as a list of (w, h) targets
    (empty = already at `size`), or None when Pillow would take a path the kernels do not restate."""
    steps: List[Tuple[int, int]] = []
    cw, ch = w, h
    t = thumbnail_size(w, h, size)
    if t is not None and t != (w, h):
        # resize(t, reducing_gap=2.0): a reduce() pre-pass when a side shrinks >= 4x; the tall-image special case
        if int(w / t[0] / 2.0) > 1 or int(h / t[1] / 2.0) > 1:
            return None
        if h > w * 100 and t[1] < h:
            return None
        steps.append(t)
        cw, ch = t
    if (cw, ch) != tuple(size):
        if ch > cw * 100 and size[1] < ch:
            return None
        steps.append((int(size[0]), int(size[1])))
    return steps


def resample_reference(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """numpy statement of ImagingResample for uint8 [H, W, C] (the checker of the kernels in tests; Pillow itself is the
    checker of this function)."""
    h, w, c = img.shape
    cur = img
    if out_w != w:
        b, kk, _ = lanczos_coeffs(w, out_w)
        out = np.empty((h, out_w, c), np.uint8)
        for xx in range(out_w):
            x0, n = int(b[xx, 0]), int(b[xx, 1])
            acc = (cur[:, x0:x0 + n].astype(np.int64) * kk[xx, :n].astype(np.int64)[None, :, None]).sum(1) + (1 << (PRECISION_BITS - 1))
            out[:, xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
        cur = out
    if out_h != h:
        b, kk, _ = lanczos_coeffs(h, out_h)
        out = np.empty((out_h, cur.shape[1], c), np.uint8)
        for yy in range(out_h):
            y0, n = int(b[yy, 0]), int(b[yy, 1])
            acc = (cur[y0:y0 + n].astype(np.int64) * kk[yy, :n].astype(np.int64)[:, None, None]).sum(0) + (1 << (PRECISION_BITS - 1))
            out[yy] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
        cur = out
    return cur
