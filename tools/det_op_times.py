"""Per-op event times of the detection forward (surya_det_forward_timed) on the GPU box, op list vs fused forms.

  python tools/det_op_times.py [--pages 16] [--size 1024] [--fuse 0,15] [--reps 5]

For every value of sa::Tuning det_fuse given: warm up, take the per-op MINIMUM over `reps` timed forwards, print one row per op
(type, shape, ms, TFLOP/s of the op's own FLOPs, GB/s of its own tensors) and bucket sums (conv3x3 / conv1x1 / depthwise / litemla /
head / other); then the wall clock of `steps` plain forwards per arm, and the heat maps of every arm against arm 0 (max abs diff).
Everything goes to stdout as text; --json adds one JSON line with the bucket sums (bench.py's detection roofline keys are built
the same way, surya_amd/detection/buckets.py)."""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pages", type=int, default=16)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--config", default="DET-DEFAULT")
    ap.add_argument("--fuse", default="0,1023")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--json", action="store_true")
    ap.add_argument("--tuning", action="append", default=[], help="sa::Tuning key=value applied before the runs (repeatable), e.g. bigtile_min_k=192")
    ap.add_argument("--rows", action="store_true", help="print the per-op rows (default: buckets only for arms after the first)")
    args = ap.parse_args()
    from surya_amd import _lib as L
    from surya_amd.config import det_config
    from surya_amd.detection.buckets import bucket_of, launch_rows, op_bytes, op_flops
    from surya_amd.detection.model import HipDetModel
    from surya_amd.synth import make_det_weights, make_pages
    from oracle import det_oracle as do

    cfg = det_config(args.config)
    sd = make_det_weights(cfg, 0)
    m = HipDetModel(cfg, sd, height=args.size, width=args.size, dtype=torch.bfloat16, max_batch=args.pages)
    x = do.normalise_pages(make_pages(args.pages, args.size, seed=1234)).cuda().contiguous()
    lib = L.lib()
    for kv in args.tuning:
        k, v = kv.split("=")
        L.check(lib.surya_set_tuning(k.encode(), C.c_int(int(v))), f"surya_set_tuning({kv})")
    arms = [int(v) for v in args.fuse.split(",")]
    heats, summary = {}, {}
    for arm in arms:
        L.check(lib.surya_set_tuning(b"det_fuse", C.c_int(arm)), "surya_set_tuning")
        for _ in range(2):
            m.forward(x)
        torch.cuda.synchronize()
        best = None
        for _ in range(args.reps):
            heat, rows = m.forward_timed(x)
            ms = [r[1] for r in rows]
            best = ms if best is None else [min(a, b) for a, b in zip(best, ms)]
        heats[arm] = heat.clone()
        ops = [r[0] for r in rows]
        buckets = {}
        print(f"\n=== det_fuse = {arm}: {args.pages} pages {args.size}^2, {args.config}, bf16; per-op min of {args.reps} event-timed forwards ===")
        ran = {r[0]: r for r in launch_rows(ops, best)}
        for i, (o, t) in enumerate(zip(ops, best)):
            _, bk, _, fl, by = ran.get(i, (i, bucket_of(o), 0.0, op_flops(o), op_bytes(o)))
            fl, by = fl * args.pages, by * args.pages
            if t > 0:
                b = buckets.setdefault(bk, [0.0, 0.0, 0.0, 0])
                b[0] += t; b[1] += fl; b[2] += by; b[3] += 1
            shape = f"{o['cin']:>5}->{o['cout']:<5} k{o['k']} s{o['stride']} {o['hin']}x{o['win']}->{o['hout']}x{o['wout']}"
            tf = fl / t / 1e9 if t > 0 else 0.0
            gb = by / t / 1e6 if t > 0 else 0.0
            print(f"{i:3d} {bk:10s} type {o['type']:2d} {shape:42s} {t*1e3:9.1f} us  {tf:8.1f} TF/s  {gb:8.0f} GB/s")
        tot = sum(best)
        print(f"--- buckets (sum of op times {tot:.3f} ms)")
        for bk, (t, fl, by, n) in sorted(buckets.items(), key=lambda kv: -kv[1][0]):
            print(f"    {bk:10s} {n:3d} launches-ish {t:8.3f} ms  {fl / t / 1e9 if t else 0:8.1f} TF/s  {by / t / 1e6 if t else 0:8.0f} GB/s")
        # wall clock
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            m.forward(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        print(f"--- wall: {dt*1e3:.3f} ms per forward = {args.pages / dt:.1f} pages/s")
        summary[arm] = {"ms_per_forward": round(dt * 1e3, 3), "pages_per_s": round(args.pages / dt, 1),
                        "buckets_ms": {k: round(v[0], 3) for k, v in buckets.items()}}
    base = heats[arms[0]]
    for arm in arms[1:]:
        d = (heats[arm] - base).abs()
        print(f"heat maps det_fuse={arm} vs {arms[0]}: max abs diff {d.max().item():.3e}, mean {d.mean().item():.3e}, "
              f"bit-identical: {bool(torch.equal(heats[arm].view(torch.int32), base.view(torch.int32)))}")
    L.check(lib.surya_set_tuning(b"det_fuse", C.c_int(1023)), "surya_set_tuning")
    if args.json:
        print(json.dumps(summary))


if __name__ == "__main__":
    main()
