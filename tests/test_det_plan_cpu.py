"""CPU: the detector's op list (detection/plan.py) interpreted in plain PyTorch (tests/det_plan_interp.py) against the oracle, which
keeps the reference's op order (oracle/det_oracle.py == EfficientViTForSemanticSegmentation, pinned by tests/golden). Covers what the
plan lowers at load -- BatchNorm folding with both epsilons, NHWC weight layouts and K padding, LiteMLA's split over the qkv and the
aggregated tensor, and the decode head's algebraic fold -- without a GPU; the GPU tests then only have the kernels left to prove."""
import pytest
import torch

from oracle import det_oracle as do
from surya_amd.config import det_config
from surya_amd.detection import plan as P
from surya_amd.synth import make_det_weights, make_pages

from det_plan_interp import run_plan


@pytest.mark.parametrize("name,size,n", [("DET-TINY", 128, 2), ("DET-TINY", 160, 1), ("DET-DEFAULT", 64, 1)])
@pytest.mark.parametrize("folded", [True, False])
def test_plan_equals_oracle(name, size, n, folded):
    cfg = det_config(name)
    sd = make_det_weights(cfg, 0)
    x = do.normalise_pages(make_pages(n, size, seed=11))
    pl = P.build_det_plan(cfg, sd, size, size, folded_head=folded)
    types = [o["type"] for o in pl.ops]
    assert (P.OP_UPSUM_CLASSIFY in types) == folded and (P.OP_UPCAT in types) == (not folded)
    with torch.inference_mode():
        planes, heat = run_plan(pl, x)
    ref_low, ref_up = do.forward(sd, cfg, x), do.heatmaps(sd, cfg, x)
    assert (planes - ref_low).abs().max().item() <= 2e-5, (planes - ref_low).abs().max().item()
    assert (heat - ref_up).abs().max().item() <= 2e-5
    assert ref_low.std().item() > 0.02                                   # a non-trivial map


def test_folded_head_algebra_and_accounting():
    """The folded head's merged weights reproduce the reference order to fp32 re-association on the SAME stage features, its
    addends are declared coarse-to-fine exactly once, and the FLOP accounting keeps the reference's figure as the algorithmic one."""
    cfg = det_config("DET-TINY")
    sd = make_det_weights(cfg, 1)
    size = 192
    a = P.build_det_plan(cfg, sd, size, size, folded_head=True)
    b = P.build_det_plan(cfg, sd, size, size, folded_head=False)
    x = do.normalise_pages(make_pages(2, size, seed=5))
    with torch.inference_mode():
        pa, ha = run_plan(a, x)
        pb, hb = run_plan(b, x)
    assert (pa - pb).abs().max().item() <= 5e-6 and (ha - hb).abs().max().item() <= 5e-6
    srcs = [o for o in a.ops if o["type"] == P.OP_UPSUM_SRC]
    head = [o for o in a.ops if o["type"] == P.OP_UPSUM_CLASSIFY][0]
    assert [(o["hin"], o["win"]) for o in srcs] == [(head["hin"] >> s, head["win"] >> s) for s in (1, 2, 3)]
    assert all(o["cin"] == head["cin"] == cfg.decoder_hidden_size for o in srcs)
    assert abs(a.reference_flops_per_image - b.flops_per_image) <= 1e-9 * b.flops_per_image
    assert b.reference_flops_per_image == b.flops_per_image
    assert a.flops_per_image < b.flops_per_image                           # the fold executes less
    assert len(a.buf_elems) < len(b.buf_elems) + 4 and max(a.buf_elems) <= max(b.buf_elems)   # no 4 x 128-channel concat buffer


def test_bucket_rows_credit_a_whole_mbconv_launch_with_its_three_ops():
    """detection/buckets.py::launch_rows (bench.py's detection.roofline keys, tools/det_op_times.py): an MBConv block run as ONE launch
    (csrc/det_mbconv.h) shows as an expand op with a time and a depthwise + projection with 0 -- it must land in the `mbconv` bucket with
    the FLOPs of all three ops and the block's boundary tensors only; with the pair form (depthwise + projection in one launch) the
    expand stays a conv1x1 row and the depthwise row keeps its own figures."""
    from surya_amd.detection.buckets import BUCKETS, launch_rows, op_bytes, op_flops
    cfg = det_config("DET-DEFAULT")
    pl = P.build_det_plan(cfg, make_det_weights(cfg, 0), 256, 256)
    ops = pl.ops
    idx = [i for i, o in enumerate(ops) if o["tag"] == "mb_expand" and ops[i + 1]["stride"] == 2]
    assert len(idx) == 2 and "mbconv" in BUCKETS
    times = [1.0] * len(ops)
    for i in idx:                                            # whole-block form on the two stride-2 transitions
        times[i + 1] = times[i + 2] = 0.0
    rows = {r[0]: r for r in launch_rows(ops, times)}
    for i in idx:
        _, b, t, fl, by = rows[i]
        assert b == "mbconv" and t == 1.0
        assert fl == op_flops(ops[i]) + op_flops(ops[i + 1]) + op_flops(ops[i + 2])
        pj = ops[i + 2]
        assert by == (ops[i]["hin"] * ops[i]["win"] * ops[i]["cin"] + pj["hout"] * pj["wout"] * pj["cout"]) * 2.0
        assert by < op_bytes(ops[i]) and i + 1 not in rows and i + 2 not in rows
    # pair form: depthwise + projection folded, expand on its own
    j = [i for i, o in enumerate(ops) if o["tag"] == "mb_expand" and ops[i + 1]["stride"] == 1][0]
    times = [1.0] * len(ops)
    times[j + 2] = 0.0
    rows = {r[0]: r for r in launch_rows(ops, times)}
    assert rows[j][1] == "conv1x1" and rows[j + 1][1] == "depthwise" and j + 2 not in rows
    assert sum(1 for r in rows.values() if r[1] == "mbconv") == 0
