"""GPU: device-fed decode runs of the layout / table decoder (surya_layout_set_feedback / _decode_steps / _wait_steps, round 4).

The heads kernel forms the fed-back token itself, embeds it and applies the next step's first norm; the host reads the records of a
whole run. Contract: the records are BIT-identical to the host-fed loop (surya_layout_decode_step with the same tokens: the fused tail
repeats the stand-alone embedding / norm kernels' arithmetic), the fed tokens equal the oracle's restatement of the reference's host
rule (oracle/layout_oracle.fed_token_*: surya/layout/__init__.py:117-169, surya/table_rec/__init__.py:80-118 + shaper.py:12-51) on
the device's own records -- including the PageHeader / PageFooter rule, made to fire by naming the most frequent synthetic class --,
and a hipGraph replay of a run equals its eager launch."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import layout_oracle as lo
from surya_amd import _lib as L
from surya_amd.layout.config import layout_config
from surya_amd.layout.model import FedRuns, HipLayoutModel, LayoutFeedbackC
from surya_amd.synth import make_layout_weights, make_table_weights
from surya_amd.table_rec.config import table_config

pytestmark = pytest.mark.gpu


def _pixels(cfg, batch, seed):
    return torch.randn(batch, 3, *cfg.encoder.image_size, generator=torch.Generator().manual_seed(seed))


def _host_fed(m, first, pos0, steps, rule):
    """The reference-shaped loop: one decode_step per box, the next token formed on the host by `rule(cls, box) -> int32 [B, w]`."""
    rec, tok = [], np.asarray(first, np.int32)
    for k in range(steps):
        cls, box = m.decode_step(tok, pos0 + k)
        tok = rule(cls, box)
        rec.append((cls, box, tok))
    return rec


def _layout_rule(d, dtype, sizes, relabel_ids):
    def rule(cls, box):
        rows = [lo.fed_token_layout(torch.from_numpy(cls[j]).to(dtype), torch.from_numpy(box[j]).to(dtype), d, None if sizes is None else tuple(sizes[j]),
                                    relabel_ids) for j in range(cls.shape[0])]
        return torch.stack(rows).numpy().astype(np.int32)
    return rule


def _set_feedback_with_ids(m, sizes, relabel_ids, skew_scaler):
    fb = LayoutFeedbackC()
    fb.skew_scaler = skew_scaler
    fb.relabel_ids[:] = list(relabel_ids)
    keep = np.ascontiguousarray(sizes, np.int32)
    fb.page_sizes = L.np_ptr(keep)
    L.check(m.lib.surya_layout_set_feedback(m.handle, C.byref(fb), C.c_int(m.batch), m._stream), "surya_layout_set_feedback")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("name", ["LAYOUT-TINY", "LAYOUT-SMALL"])
def test_layout_device_fed_runs_equal_the_host_fed_loop(hip_lib, name, dtype):
    cfg = layout_config(name)
    d = cfg.decoder
    B, steps = 5, 22
    m = HipLayoutModel(cfg, make_layout_weights(cfg, 0), dtype=dtype, max_batch=8, max_boxes=32)
    px = _pixels(cfg, B, 11).cuda().contiguous()
    first = np.full((B, 7), d.bos_token_id, np.int32)
    sizes = np.array([[612, 792], [1200, 300], [90, 2000], [1024, 1024], [777, 333]], np.int32)
    # pass 1, no rule: find the class the synthetic weights like best, so that the header / footer rule can be made to fire
    m.encode(px)
    plain = _host_fed(m, first, 0, steps, _layout_rule(d, dtype, None, None))
    classes = np.concatenate([r[0].argmax(-1) for r in plain])
    ids = [int(np.bincount(classes).argmax()), int(classes[-1])]
    # pass 2: host-fed with the rule on; pass 3: device-fed, runs of 7 + 7 + 8 steps
    m.encode(px)
    want = _host_fed(m, first, 0, steps, _layout_rule(d, dtype, sizes, ids))
    fired = sum(int((r[2][:, 6] != r[0].argmax(-1)).sum()) for r in want)
    assert fired > 0, "the header / footer rule never fired: the test would not cover it"
    m.encode(px)
    _set_feedback_with_ids(m, sizes, ids, d.skew_scaler)
    got, pos = [], 0
    for run, n in enumerate((7, 7, 8)):
        m.decode_steps(first if run == 0 else None, pos, n, run & 1)
        cls, box, tok = m.wait_steps(n, run & 1)
        got += [(cls[k], box[k], tok[k]) for k in range(n)]
        pos += n
    for k, (a, b) in enumerate(zip(want, got)):
        assert np.array_equal(a[2], b[2]), (k, a[2], b[2])
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), k
    # a run that does not continue where the last one stopped is refused
    assert m.lib.surya_layout_decode_steps(m.handle, None, C.c_int(B), C.c_int(3), C.c_int(2), C.c_int(0), m._stream) == L.SA_ERR_STATE


def test_layout_graph_replay_equals_eager_and_fedruns_serves_single_steps(hip_lib):
    """The same (rows, steps) shape three times: eager, capture + replay, replay -- identical records; then FedRuns hands the same
    records out one step at a time and accepts the host's tokens."""
    cfg = layout_config("LAYOUT-SMALL")
    d = cfg.decoder
    B = 4
    m = HipLayoutModel(cfg, make_layout_weights(cfg, 0), dtype=torch.bfloat16, max_batch=4, max_boxes=40)
    L.check(m.lib.surya_set_tuning(b"graph", C.c_int(1)), "surya_set_tuning(graph)")          # replay is opt-in (measured slower than plain launches)
    px = _pixels(cfg, B, 3).cuda().contiguous()
    first = np.full((B, 7), d.bos_token_id, np.int32)
    sizes = np.array([[612, 792]] * B, np.int32)
    runs = []
    for rep in range(3):
        m.encode(px)
        m.set_feedback(sizes)
        rec = []
        for run in range(3):
            m.decode_steps(first if run == 0 else None, run * 8, 8, run & 1)
            rec.append(m.wait_steps(8, run & 1))
        runs.append(rec)
    for rep in (1, 2):
        for x, y in zip(runs[0], runs[rep]):
            assert all(np.array_equal(a, b) for a, b in zip(x, y)), rep
    m.encode(px)
    m.set_feedback(sizes)
    fr = FedRuns(m, 0, 24, 8)
    rule = _layout_rule(d, torch.bfloat16, sizes, None)
    tok = first
    flat = [(c[k], b[k], t[k]) for c, b, t in runs[0] for k in range(8)]
    for k in range(24):
        cls, box = fr.step(tok)
        assert np.array_equal(cls, flat[k][0]) and np.array_equal(box, flat[k][1])
        tok = rule(cls, box)
    with pytest.raises(L.SuryaAmdError):
        fr.step(tok)                                     # max_steps exhausted
    # a host token that is not the device's: refused loudly
    m.encode(px)
    m.set_feedback(sizes)
    fr = FedRuns(m, 0, 8, 4)
    cls, box = fr.step(first)
    wrong = rule(cls, box)
    wrong[1, 6] += 1
    with pytest.raises(L.SuryaAmdError, match="device-fed token differs"):
        fr.step(wrong)
    L.check(m.lib.surya_set_tuning(b"graph", C.c_int(0)), "surya_set_tuning(graph)")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_table_device_fed_runs_equal_the_host_fed_loop(hip_lib, dtype):
    cfg = table_config("TABLE-SMALL")
    d = cfg.decoder
    B, T, steps = 6, 3, 20
    m = HipLayoutModel(cfg, make_table_weights(cfg, 0), dtype=dtype, max_batch=8, max_boxes=64)
    px = _pixels(cfg, 3, 5).cuda().contiguous()
    rng = np.random.default_rng(2)
    prompt = np.concatenate([rng.integers(0, 1025, (B, T, 6)), rng.integers(5, 10, (B, T, 1)), rng.integers(5, 9, (B, T, 1)),
                             rng.integers(1, 4, (B, T, 1)), rng.integers(5, 7, (B, T, 1))], -1).astype(np.int32)

    def rule(cls, box):
        return torch.stack([lo.fed_token_table(torch.from_numpy(cls[j]).to(dtype).float(), torch.from_numpy(box[j]).to(dtype).float(), d)
                            for j in range(cls.shape[0])]).numpy().astype(np.int32)

    src = [0, 1, 2, 2, 0, 1]
    m.encode(px)
    m.select(src)
    cls, box = m.prefill(prompt)
    first = rule(cls, box)
    want = _host_fed(m, first, T, steps, rule)
    m.encode(px)
    m.select(src)
    m.prefill(prompt)
    m.set_feedback()
    got, pos = [], T
    for run, n in enumerate((16, 4)):
        m.decode_steps(first if run == 0 else None, pos, n, run & 1)
        c, b, t = m.wait_steps(n, run & 1)
        got += [(c[k], b[k], t[k]) for k in range(n)]
        pos += n
    for k, (a, b) in enumerate(zip(want, got)):
        assert np.array_equal(a[2], b[2]), (k, a[2], b[2])
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), k
