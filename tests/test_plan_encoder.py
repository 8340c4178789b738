"""CPU: the library's host-side encoder planning (window order, rotary position ids, window boundaries) against
the oracle's restatement of get_window_index / rot_pos_emb (encoder/__init__.py:523-597)."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import rec_oracle as ro
from surya_amd import _lib as L

GRID_SETS = [
    [(6, 38)], [(10, 18), (8, 24)], [(28, 28)], [(8, 8)], [(2, 2)], [(2, 116), (16, 16), (6, 10)],
    [(38, 74)], [(4, 4), (4, 6), (12, 12)],
]


@pytest.mark.parametrize("grids", GRID_SETS)
def test_plan_matches_oracle(hip_lib, grids):
    cfg = L.RecConfigC(merge=2, window_tokens=4, embed_multiplier=256)
    g = np.asarray(grids, np.int32).reshape(-1)
    P = int(sum(h * w for h, w in grids))
    src = np.zeros(P, np.int32); pos = np.zeros(2 * P, np.int32)
    cu = np.zeros(P // 4 + len(grids) + 2, np.int32); nwin = C.c_int(0); msrc = np.zeros(P // 4, np.int32)
    rc = hip_lib.surya_rec_plan_encoder(C.byref(cfg), L.np_ptr(g), len(grids), L.np_ptr(src), L.np_ptr(pos), L.np_ptr(cu),
                                        C.byref(nwin), L.np_ptr(msrc))
    assert rc == 0
    thw = [(1, h, w) for h, w in grids]
    widx, cu_ref = ro.window_index(thw, 112, 2, 14)
    assert np.array_equal(msrc, widx.numpy())
    assert np.array_equal(cu[: nwin.value + 1], cu_ref.numpy())
    # row permutation: window-ordered row r <- original row
    ref_rows = (widx[:, None] * 4 + torch.arange(4)[None, :]).reshape(-1).numpy()
    assert np.array_equal(src, ref_rows)
    ref_pos = ro.vision_pos_ids(thw, 2).numpy()[ref_rows]
    assert np.array_equal(pos.reshape(-1, 2), ref_pos)


def test_plan_rejects_odd_grid(hip_lib):
    cfg = L.RecConfigC(merge=2, window_tokens=4, embed_multiplier=256)
    g = np.asarray([3, 4], np.int32)
    assert hip_lib.surya_rec_plan_encoder(C.byref(cfg), L.np_ptr(g), 1, None, None, None, None, None) == -2
