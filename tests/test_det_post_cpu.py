"""CPU: the heat-map -> boxes pipeline the HIP kernels implement (surya_amd/csrc/det_post_core.h, compiled here with g++ into a
host harness, tests/native/det_post_host.cpp) against surya_amd/detection/heatmap.py (the restatement of
surya/detection/heatmap.py:14-107) on synthetic heat maps: blobs, rotated bars, components touching the borders, noise.
The same header is included by the device code, so this pins the per-component geometry (dilated row extremes, hull order,
calipers tie rule, corner order) without a GPU; the device-side labelling / compaction is checked in tests/test_gpu_det.py."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from surya_amd.detection import heatmap as hm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def native(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("native") / "libdet_post_host.so")
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", out,
                    os.path.join(ROOT, "tests", "native", "det_post_host.cpp")], check=True)
    lib = C.CDLL(out)
    lib.det_post_host.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p]
    return lib


def synth_map(h, w, seed):
    """Heat map in (0, 1): low noise floor + soft blobs: axis-aligned bars, rotated bars, dots, border-touching pieces."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    m = rng.random((h, w), dtype=np.float32) * 0.12
    for _ in range(int(rng.integers(6, 30))):
        cx, cy = rng.uniform(-10, w + 10), rng.uniform(-10, h + 10)
        L, T = rng.uniform(4, w / 3), rng.uniform(2, 14)
        th = rng.choice([0.0, 0.0, rng.uniform(-0.6, 0.6), np.pi / 2])
        u = (xx - cx) * np.cos(th) + (yy - cy) * np.sin(th)
        v = -(xx - cx) * np.sin(th) + (yy - cy) * np.cos(th)
        blob = np.exp(-np.maximum(np.abs(u) / L, np.abs(v) / T) ** 4) * rng.uniform(0.45, 0.98)
        m = np.maximum(m, blob.astype(np.float32))
    return np.ascontiguousarray(np.clip(m, 0.001, 0.999).astype(np.float32))


def run_native(lib, heat, cap=4096):
    h, w = heat.shape
    boxes = np.zeros((cap, 4, 2), np.float32); conf = np.zeros(cap, np.float32); n = C.c_int(0); thr = np.zeros(3, np.float32)
    rc = lib.det_post_host(heat.ctypes.data, h, w, 0.6, 0.35, cap, boxes.ctypes.data, conf.ctypes.data, C.byref(n), thr.ctypes.data)
    assert rc == 0
    return boxes[: n.value], conf[: n.value], thr


@pytest.mark.parametrize("seed,shape", [(0, (160, 224)), (1, (256, 256)), (2, (96, 400)), (3, (333, 257)), (4, (512, 512)), (5, (64, 64))])
def test_native_pipeline_matches_python(native, seed, shape):
    heat = synth_map(*shape, seed)
    ref_boxes, ref_conf = hm.detect_boxes(heat, 0.6, 0.35)
    tt, lt = hm.get_dynamic_thresholds(heat, 0.6, 0.35)
    boxes, conf, thr = run_native(native, heat)
    assert abs(thr[0] - tt) <= 1e-6 and abs(thr[1] - lt) <= 1e-6
    assert len(boxes) == len(ref_boxes) and len(boxes) > 0
    worst = 0.0
    for b, rb in zip(boxes, ref_boxes):
        worst = max(worst, float(np.abs(b - np.asarray(rb, np.float32)).max()))
    # same rectangle, same corner order; float64 calipers on both sides differ at most in the last float32 bit
    assert worst <= 1e-3, worst
    assert np.allclose(conf, np.asarray(ref_conf, np.float32), rtol=0, atol=1e-7)


def test_noise_map_many_small_components(native):
    """A map that is mostly noise around the threshold (what random weights produce): hundreds of small components, the area and
    text-threshold filters, label order."""
    rng = np.random.default_rng(9)
    heat = np.ascontiguousarray((rng.random((192, 192), dtype=np.float32) * 0.5 + 0.3).astype(np.float32))
    k = np.ones((3, 3), np.float32) / 9
    from scipy.ndimage import convolve
    heat = np.ascontiguousarray(convolve(heat, k, mode="nearest").astype(np.float32))
    ref_boxes, ref_conf = hm.detect_boxes(heat, 0.6, 0.35)
    boxes, conf, _ = run_native(native, heat)
    assert len(boxes) == len(ref_boxes)
    for b, rb in zip(boxes, ref_boxes):
        assert np.abs(b - np.asarray(rb, np.float32)).max() <= 1e-3
    assert np.allclose(conf, np.asarray(ref_conf, np.float32), atol=1e-7)
