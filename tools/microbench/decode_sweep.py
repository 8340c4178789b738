"""Decode-step A/B sweep inside ONE process (one gpurun call): REC-FULL bf16, 256 active slots, bench-shaped prompts.

For every tuning configuration (surya_set_tuning, csrc/common.h sa::Tuning) this re-prefills the same 256 lines, runs
`--steps` decode steps in calls of 4 (the predictor's steps_per_sync) with the next call enqueued before the previous one is
read, and reports wall us/step (HIP events around the whole run) plus whether the greedy tokens equal the first
configuration's (tile / split-K changes re-order fp32 sums, so bf16 argmax near-ties may flip; reported, not asserted).

    python tools/microbench/decode_sweep.py [--steps 32] [--configs all|base|fp8]

`fp8`: bf16 decode vs the MXFP8 decode path (HipRecModel.set_decode_fp8, csrc/gemm_mx.h) on the same lines.

The tile-shape / dual-stream / lm_head-ring / skinny-GEMM variants swept in round 2 lost and were removed from the library; their
results are in profiles/r02_decode_sweeps.md.
"""
from __future__ import annotations

import argparse
import ctypes as C
import itertools
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--slots", type=int, default=256)
    ap.add_argument("--configs", default="all")
    args = ap.parse_args()
    from surya_amd import _lib as L
    from surya_amd.config import rec_config
    from surya_amd.recognition.model import HipRecModel
    from surya_amd.synth import make_rec_weights
    from util import make_prompts, crop_grid

    lib = L.lib()
    cfg = rec_config("REC-FULL")
    sd = make_rec_weights(cfg, 0)
    n = args.slots
    m = HipRecModel(cfg, sd, image_token_id=cfg.image_token_id, pad_token_id=cfg.pad_token_id, eos_token_id=cfg.eos_token_id,
                    dtype=torch.bfloat16, max_slots=n, max_kv_len=64 + args.steps + 40, max_patches=65536, max_prefill_tokens=n * 72)
    rng = np.random.default_rng(1234)
    grids = [crop_grid(64, int(w)) for w in sorted(rng.integers(128, 513, size=n), reverse=True)]
    tiles, seqs = make_prompts(cfg, grids, seed=3)
    tiles = tiles.cuda().contiguous()
    slots = list(range(n))

    def setk(**kw):
        for k, v in kw.items():
            L.check(lib.surya_set_tuning(k.encode(), C.c_int(v)), f"surya_set_tuning({k})")

    base = dict(graph=0, split_target=256, split_min_kt=4, split_max=8, dattn=4, rnorm=2, ghead=2, fuse_embed=1, lmhead=1, kvprefetch=0, gateup_ring=2, big_m_split=-1, big_m_gateup=-1)
    if args.configs == "base":
        variants = [dict()]
    elif args.configs == "fewer":        # fewer, longer split-K slices for the bf16 path
        variants = [dict(), dict(split_min_kt=6), dict(split_min_kt=8), dict(split_min_kt=10), dict(split_max=2), dict(split_target=192),
                    dict(split_target=128), dict(split_max=1)]
    elif args.configs == "r4":           # round 4: every new decode-step kernel against the round-3 kernel it replaces, one at a time
        variants = [dict(), dict(dattn=3), dict(rnorm=1), dict(ghead=1, fuse_embed=0), dict(fuse_embed=0), dict(lmhead=0), dict(lmhead=2),
                    dict(dattn=3, rnorm=1, ghead=1, fuse_embed=0, lmhead=0), dict(graph=1), dict()]
    elif args.configs == "r5":           # round 5: LDS stages of the decode gate|up loop (2 = unrolled pair, 3 / 4 = ring with counted vmcnt), interleaved
        variants = [dict(), dict(gateup_ring=3), dict(gateup_ring=4), dict(), dict(gateup_ring=3), dict(gateup_ring=4)]
    elif args.configs == "bigm":         # round 6: tiles of the decode projections above 256 rows (run with --slots 512 / 1024), interleaved
        variants = [dict(big_m_split=0, big_m_gateup=0), dict(), dict(big_m_split=2, big_m_gateup=0), dict(big_m_split=3, big_m_gateup=0),
                    dict(big_m_split=0, big_m_gateup=2), dict(big_m_split=2, big_m_gateup=2), dict(big_m_split=2, big_m_gateup=1),
                    dict(big_m_split=0, big_m_gateup=0), dict()]
    elif args.configs == "pf":           # K/V prefetch workgroups in the reduce kernels, on / off, interleaved
        variants = [dict(kvprefetch=1), dict(), dict(kvprefetch=1), dict(), dict(lmhead=2, kvprefetch=1), dict(lmhead=2)]
    elif args.configs == "fp8only":
        variants = [dict(fp8=1)]
    elif args.configs == "fp8":
        variants = [dict(), dict(fp8=1), dict(fp8=1, split_min_kt=2), dict(fp8=1, split_min_kt=3), dict(fp8=1, split_target=320)]
    else:
        variants = [dict(), dict(graph=1), dict(split_target=512), dict(split_target=384), dict(split_target=192), dict(split_min_kt=2),
                    dict(split_max=4)]

    def run(n_steps):
        m.prefill(tiles, grids, seqs, slots)
        m.read_outputs(1)
        m.set_active(slots)
        toks = []
        calls = [(min(4, n_steps - i), (i // 4) & 1) for i in range(0, n_steps, 4)]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        m.decode_async(*calls[0])
        for i, call in enumerate(calls):
            if i + 1 < len(calls):
                m.decode_async(*calls[i + 1])
            t, _, _ = m.wait_outputs(*call)
            toks.append(t[: call[0], :n].copy())
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n_steps, (time.perf_counter() - t0) * 1e6 / n_steps, np.concatenate(toks)

    ref = None
    print(f"# REC-FULL bf16, {n} active slots, {args.steps} decode steps per run, us/step (event) | us/step (host wall) | tokens == config 0")
    for v in variants:
        v = dict(v)
        m.set_decode_fp8(bool(v.pop("fp8", 0)))
        setk(**{**base, **v})
        v = {**v, "fp8": int(m.decode_fp8)}
        run(8)                                   # warm-up (attribute set, graph capture on 2nd sight)
        run(8)
        best = min((run(args.steps) for _ in range(3)), key=lambda r: r[0])
        if ref is None:
            ref = best[2]
        same = float((best[2] == ref).all(axis=0).mean())
        print(f"{str(v):90s} {best[0]:8.1f} {best[1]:8.1f}   lines identical {same:.3f}", flush=True)
    setk(**base)
    m.set_decode_fp8(False)


if __name__ == "__main__":
    main()
