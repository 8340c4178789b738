"""Where the detection forward goes: buckets over the op list of detection/plan.py, with each op's own FLOPs and tensor bytes.

Shared by bench.py (`detection.roofline` bucket keys) and tools/det_op_times.py. An op's bytes are the tensors it
reads and writes ONCE each in the storage type (bf16 = 2 bytes: the detector's product path) -- the op list's own traffic; a fused
form that runs two ops in one launch is timed under its first op and moves fewer bytes than the sum listed here.
"""
from __future__ import annotations

from .plan import (OP_CLASSIFY, OP_CONV, OP_DWCONV, OP_GROUPED1X1, OP_INPUT, OP_LITEMLA, OP_UPCAT, OP_UPSAMPLE_OUT, OP_UPSUM_CLASSIFY,
                   OP_UPSUM_SRC)

BUCKETS = ("conv3x3", "conv1x1", "depthwise", "mbconv", "litemla", "head", "other")

_TAG_BUCKET = {"stem": "conv3x3", "fmb_3x3": "conv3x3", "fmb_proj": "conv1x1", "mb_expand": "conv1x1", "mb_proj": "conv1x1",
               "mla_qkv": "conv1x1", "mla_proj": "conv1x1", "mb_dw": "depthwise", "mla_dw5": "litemla", "mla_g1x1": "litemla",
               "mla_attn": "litemla", "head_z": "head", "head": "head", "input": "other", "conv": "conv1x1"}


def bucket_of(op: dict) -> str:
    return _TAG_BUCKET.get(op.get("tag", ""), "other")


def op_flops(op: dict) -> float:
    """FLOPs per image of one op (multiply-add = 2)."""
    t = op["type"]
    px = op["hout"] * op["wout"]
    if t == OP_CONV:
        return 2.0 * px * op["cout"] * op["k"] * op["k"] * op["cin"]
    if t == OP_DWCONV:
        return 2.0 * px * op["cin"] * op["k"] * op["k"]
    if t == OP_GROUPED1X1:
        return 2.0 * px * op["cin"] * op["p0"]
    if t == OP_LITEMLA:
        return 2.0 * 2 * px * (op["cout"] // op["p0"]) * op["p0"] * (op["p0"] + 1)
    if t in (OP_CLASSIFY, OP_UPSUM_CLASSIFY):
        return 2.0 * px * op["cin"] * op["cout"]
    return 0.0


def op_bytes(op: dict, elem: int = 2) -> float:
    """Bytes per image the op reads + writes, each tensor once."""
    t = op["type"]
    pin, pout = op["hin"] * op["win"], op["hout"] * op["wout"]
    if t == OP_INPUT:
        return pin * op["cin"] * 4.0 + pout * op["cout"] * elem
    if t == OP_CONV:
        return (pin * op["cin"] + pout * op["cout"] * (2 if op["res"] >= 0 else 1)) * float(elem)
    if t in (OP_DWCONV, OP_GROUPED1X1):
        return (pin * op["cin"] + pout * op["cout"]) * float(elem)
    if t == OP_LITEMLA:
        return (pin * op["cin"] * 2.0 + pout * op["cout"]) * elem      # q | k | v of the conv and of the aggregation; the heads' outputs
    if t == OP_UPCAT:
        return (pin * op["cin"] + pout * op["cin"]) * float(elem)
    if t == OP_CLASSIFY:
        return pin * op["cin"] * float(elem) + pout * op["cout"] * 4.0
    if t == OP_UPSUM_CLASSIFY:
        return pin * op["cin"] * float(elem) * (1 + 1 / 4 + 1 / 16 + 1 / 64) + pout * op["cout"] * 4.0
    if t == OP_UPSAMPLE_OUT:
        return (pin + pout) * op["cout"] * 4.0
    if t == OP_UPSUM_SRC:
        return 0.0
    return 0.0


def launch_rows(ops, times_ms):
    """One row per op that RAN: (index, bucket, ms, flops per image, bytes per image).

    A whole-MBConv launch (csrc/det_mbconv.h, det_fuse bit 6) shows as an mb_expand op with a time whose depthwise and projection both report
    0: it goes to the `mbconv` bucket with the FLOPs of all three ops and the block's boundary tensors only (the expanded tensor is neither
    written nor read). Every other fused pair stays under its first op's bucket, as before, with that op's own figures."""
    out = []
    n = len(ops)
    for i, (o, t) in enumerate(zip(ops, times_ms)):
        if t <= 0:
            continue
        if (o.get("tag") == "mb_expand" and i + 2 < n and ops[i + 1].get("tag") == "mb_dw" and ops[i + 2].get("tag") == "mb_proj"
                and times_ms[i + 1] <= 0 and times_ms[i + 2] <= 0):
            pj = ops[i + 2]
            fl = op_flops(o) + op_flops(ops[i + 1]) + op_flops(pj)
            by = (o["hin"] * o["win"] * o["cin"] + pj["hout"] * pj["wout"] * pj["cout"] * (2 if pj["res"] >= 0 else 1)) * 2.0
            out.append((i, "mbconv", t, fl, by))
        else:
            out.append((i, bucket_of(o), t, op_flops(o), op_bytes(o)))
    return out
