"""Generate the BASELINE-configuration fixtures from the REAL reference modules (build container only; takes minutes).

    python oracle/make_golden_full.py [rec8] [rec256] [small256] [det1024] [rec8c] [rec256c]

Like oracle/make_golden.py this imports VikParuchuri/surya @ v0.14.6 from /root/reference through oracle/ref_shim and
loads the seeded synthetic weights into the reference's own nn.Modules. What it records is the bench's own workload:

  rec_full_bench8.pt    REC-FULL, 8 of bench.py's 256 line crops (spread over the widths), 48 greedy tokens each:
                        tokens, bbox ints + un-truncated values, scores, the reference's top-32 logits / logsumexp per
                        step, and the reference's OWN bf16-vs-fp32 logit deviation per step (teacher forced; the
                        rounding model the bf16 HIP path is held to)
  rec_full_bench256.pt  REC-FULL, all 256 bench crops in one left-padded batch: prefill + 3 decode steps (M = 256 tiles,
                        split-K with 4 M-tiles, the 128x128 fused-argmax lm_head) -- tokens, bbox ints, top-8 logits
  rec_full_cond8.pt     the same 8 crops x 48 tokens on the CONDITIONED weight set (rec8c): a model whose bf16 run is a small
                        perturbation of its fp32 run, + the reference's own free-running bf16 stream
  rec_full_cond256.pt   all 256 crops, conditioned weights, 48 tokens each (the headline configuration over its full extent), with the
                        reference's bf16 deviation per step and its own free-running bf16 stream (rec256c)
  rec_small_256.pt      REC-SMALL, 256 ragged synthetic prompts, 12 steps (same tile paths, deeper)
  det_default_1024.pt   DET-DEFAULT, one synthetic 1024^2 page: the module's [1, 2, 256, 256] output, the x4 upsample
                        subsampled, and the reference's own bf16-vs-fp32 deviation

The fixtures travel to the GPU box, where /root/reference does not exist; tests/test_gpu_baseline_parity.py compares
the HIP path with them, tests/test_oracle_golden.py pins oracle/*.py against them on the CPU.
"""
from __future__ import annotations

import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch
import torch.nn.functional as F

from oracle import ref_shim

ref_shim.install()

from oracle.make_golden import build_reference_rec            # noqa: E402
from surya_amd.config import rec_config, det_config            # noqa: E402
from surya_amd.synth import make_rec_weights, make_det_weights, make_pages   # noqa: E402
from util import bench_line_inputs, make_prompts, left_pad_batch, crop_grid   # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
BENCH8 = [0, 36, 73, 109, 146, 182, 219, 255]     # positions in the widest-first order of bench.py's 256 crops


def run_reference(ref, cfg, tiles, grids, seqs, steps, forced=None):
    """Greedy (or teacher-forced) loop on the reference's SuryaModel: prefill + steps-1 decode calls
    (recognition/__init__.py:326-352, 398-409 call shapes). Returns per-step logits [steps, B, V] and bbox logits."""
    from transformers import DynamicCache
    ids, am, pos = left_pad_batch(cfg, seqs)
    grid = torch.tensor([(1, h, w) for h, w in grids])
    dt = next(ref.parameters()).dtype
    logits, bboxes, tokens = [], [], []
    with torch.inference_mode():
        cache = DynamicCache()
        out = ref(input_ids=ids, image_tiles=tiles.to(dt), grid_thw=grid, attention_mask=am, position_ids=pos,
                  past_key_values=cache, use_cache=True, logits_to_keep=1, encoder_chunk_size=4096)
        for step in range(steps):
            lm, bb = out.lm_logits[:, -1].float(), out.bbox_logits[:, -1].float()
            logits.append(lm.clone()); bboxes.append(bb.clone())
            nxt = lm.argmax(-1, keepdim=True) if forced is None else forced[step].reshape(-1, 1)
            tokens.append(nxt[:, 0].clone())
            if step + 1 == steps:
                break
            am = F.pad(am, (0, 1), value=1)
            pos = pos[:, -1:] + 1
            out = ref(input_ids=nxt, attention_mask=am, position_ids=pos, use_cache=True, past_key_values=cache, logits_to_keep=1)
    return torch.stack(logits), torch.stack(bboxes), torch.stack(tokens)


def pack(cfg, lg, bb, tk, topk):
    t = torch.topk(lg, topk, dim=-1)
    return {"tokens": tk, "bbox_raw": bb * cfg.bbox_size, "bbox_ints": (bb * cfg.bbox_size).to(torch.long),
            "scores": torch.softmax(lg, -1).amax(-1), "logits_top": {"values": t.values.clone(), "indices": t.indices.clone()},
            "logits_lse": torch.logsumexp(lg, -1), "logits_absmax": lg.abs().amax(-1)}


def rec8():
    cfg = rec_config("REC-FULL")
    sd = make_rec_weights(cfg, 0)
    tiles, grids, seqs = bench_line_inputs(cfg, 256, seed=1234, pick=BENCH8)
    ref = build_reference_rec(cfg, sd, "eager")
    t0 = time.time()
    lg, bb, tk = run_reference(ref, cfg, tiles, grids, seqs, 48)
    print(f"rec8 fp32 reference: {time.time() - t0:.1f}s", flush=True)
    g = {"config": "REC-FULL", "attn": "eager", "lines": 256, "line_seed": 1234, "pick": BENCH8, "grids": grids,
         "tiles_sum": float(tiles.double().sum()), **pack(cfg, lg, bb, tk, 32)}
    t0 = time.time()
    lgb, _, _ = run_reference(ref.bfloat16(), cfg, tiles, grids, seqs, 48, forced=tk)
    print(f"rec8 bf16 reference: {time.time() - t0:.1f}s", flush=True)
    g["bf16_dev"] = (lgb - lg).abs().amax(-1)                 # [steps, B]: the reference's own rounding deviation
    g["bf16_dev_top"] = (torch.gather(lgb, -1, g["logits_top"]["indices"]) - g["logits_top"]["values"]).abs().amax(-1)
    torch.save(g, os.path.join(GOLD, "rec_full_bench8.pt"))


def rec8c():
    """The same 8 bench crops on the CONDITIONED weight set (surya_amd.synth.make_rec_weights_conditioned): the reference's own
    bf16 run stays within ~1-2 % of max|logit| of its fp32 run there, so the bf16 HIP path can be held to a tolerance that a wrong
    kernel fails, and to an argmax check that covers most positions. Also records the reference's OWN free-running bf16 greedy
    stream: how far bf16 token agreement can be expected to go at all."""
    cfg = rec_config("REC-FULL")
    sd = make_rec_weights(cfg, 0, recipe="conditioned")
    tiles, grids, seqs = bench_line_inputs(cfg, 256, seed=1234, pick=BENCH8)
    ref = build_reference_rec(cfg, sd, "eager")
    t0 = time.time()
    lg, bb, tk = run_reference(ref, cfg, tiles, grids, seqs, 48)
    print(f"rec8c fp32 reference: {time.time() - t0:.1f}s", flush=True)
    g = {"config": "REC-FULL", "recipe": "conditioned", "attn": "eager", "lines": 256, "line_seed": 1234, "pick": BENCH8, "grids": grids,
         "tiles_sum": float(tiles.double().sum()), **pack(cfg, lg, bb, tk, 32)}
    t0 = time.time()
    refb = ref.bfloat16()
    lgb, _, _ = run_reference(refb, cfg, tiles, grids, seqs, 48, forced=tk)
    g["bf16_dev"] = (lgb - lg).abs().amax(-1)
    g["bf16_dev_top"] = (torch.gather(lgb, -1, g["logits_top"]["indices"]) - g["logits_top"]["values"]).abs().amax(-1)
    _, _, tkb = run_reference(refb, cfg, tiles, grids, seqs, 48)
    g["bf16_free_tokens"] = tkb                                  # [steps, B]
    print(f"rec8c bf16 reference (teacher forced + free running): {time.time() - t0:.1f}s", flush=True)
    same = (tkb == tk).all(0)
    print(f"rec8c: reference bf16 free-running == fp32 on {int(same.sum())}/{len(same)} lines; bf16 dev / max = "
          f"{float((g['bf16_dev'].amax(-1) / g['logits_absmax'].amax(-1)).max()):.4f}", flush=True)
    torch.save(g, os.path.join(GOLD, "rec_full_cond8.pt"))


class RefStepper:
    """run_reference, one step at a time (the 256-line x 48-step fixture cannot hold [48, 256, V] logits: 4 GB per run): `.lm` / `.bb`
    are the current step's fp32 views of the reference's outputs, `.advance(tokens)` feeds the next ids."""

    def __init__(self, ref, cfg, tiles, grids, seqs):
        from transformers import DynamicCache
        self.ref = ref
        ids, self.am, self.pos = left_pad_batch(cfg, seqs)
        grid = torch.tensor([(1, h, w) for h, w in grids])
        dt = next(ref.parameters()).dtype
        self.cache = DynamicCache()
        with torch.inference_mode():
            out = ref(input_ids=ids, image_tiles=tiles.to(dt), grid_thw=grid, attention_mask=self.am, position_ids=self.pos,
                      past_key_values=self.cache, use_cache=True, logits_to_keep=1, encoder_chunk_size=4096)
        self._take(out)

    def _take(self, out):
        self.lm, self.bb = out.lm_logits[:, -1].float(), out.bbox_logits[:, -1].float()

    def advance(self, nxt):
        self.am = F.pad(self.am, (0, 1), value=1)
        self.pos = self.pos[:, -1:] + 1
        with torch.inference_mode():
            self._take(self.ref(input_ids=nxt.reshape(-1, 1), attention_mask=self.am, position_ids=self.pos, use_cache=True,
                                past_key_values=self.cache, logits_to_keep=1))


def rec256c(steps=48):
    """All 256 bench crops, conditioned weights, prefill + 47 decode steps: the headline configuration over its FULL extent (round 4
    recorded 4 steps; VERDICT r04 "missing" #3). Three reference runs, reduced step by step: fp32 free-running (the tokens, bbox,
    scores, top-8 logits), the reference's bf16 modules teacher-forced with those tokens in lock-step (its own rounding deviation per
    step and line), and its bf16 modules free-running (how far bf16 token agreement can be expected to go at all)."""
    import copy
    cfg = rec_config("REC-FULL")
    sd = make_rec_weights(cfg, 0, recipe="conditioned")
    tiles, grids, seqs = bench_line_inputs(cfg, 256, seed=1234)
    ref = build_reference_rec(cfg, sd, "sdpa")
    refb = copy.deepcopy(ref).bfloat16()
    t0 = time.time()
    a, b = RefStepper(ref, cfg, tiles, grids, seqs), RefStepper(refb, cfg, tiles, grids, seqs)
    print(f"rec256c prefill fp32 + bf16: {time.time() - t0:.1f}s", flush=True)
    rows = {k: [] for k in ("tokens", "bbox_raw", "bbox_ints", "scores", "tv", "ti", "logits_lse", "logits_absmax", "bf16_dev")}
    for step in range(steps):
        lg = a.lm
        tk = lg.argmax(-1)
        t = torch.topk(lg, 8, dim=-1)
        rows["tokens"].append(tk.clone()); rows["bbox_raw"].append(a.bb * cfg.bbox_size); rows["bbox_ints"].append((a.bb * cfg.bbox_size).to(torch.long))
        rows["scores"].append(torch.softmax(lg, -1).amax(-1)); rows["tv"].append(t.values.clone()); rows["ti"].append(t.indices.clone())
        rows["logits_lse"].append(torch.logsumexp(lg, -1)); rows["logits_absmax"].append(lg.abs().amax(-1))
        rows["bf16_dev"].append((b.lm - lg).abs().amax(-1))
        if step + 1 < steps:
            a.advance(tk); b.advance(tk)
        if step % 8 == 7:
            print(f"rec256c step {step + 1}/{steps}: {time.time() - t0:.1f}s", flush=True)
    g = {"config": "REC-FULL", "recipe": "conditioned", "attn": "sdpa", "lines": 256, "line_seed": 1234, "grids": grids,
         "tiles_sum": float(tiles.double().sum())}
    for k in ("tokens", "bbox_raw", "bbox_ints", "scores", "logits_lse", "logits_absmax", "bf16_dev"):
        g[k] = torch.stack(rows[k])
    g["logits_top"] = {"values": torch.stack(rows["tv"]), "indices": torch.stack(rows["ti"])}
    del a, b
    t0 = time.time()
    c = RefStepper(refb, cfg, tiles, grids, seqs)
    free = []
    for step in range(steps):
        tk = c.lm.argmax(-1)
        free.append(tk.clone())
        if step + 1 < steps:
            c.advance(tk)
    g["bf16_free_tokens"] = torch.stack(free)                    # [steps, 256]: the reference's OWN bf16 greedy stream
    same = (g["bf16_free_tokens"] == g["tokens"]).all(0)
    print(f"rec256c bf16 free-running: {time.time() - t0:.1f}s; reference bf16 == fp32 on {int(same.sum())}/256 lines over {steps} steps; "
          f"bf16 dev / max = {float((g['bf16_dev'].amax(-1) / g['logits_absmax'].amax(-1)).max()):.4f}", flush=True)
    torch.save(g, os.path.join(GOLD, "rec_full_cond256.pt"))


def rec256():
    cfg = rec_config("REC-FULL")
    sd = make_rec_weights(cfg, 0)
    tiles, grids, seqs = bench_line_inputs(cfg, 256, seed=1234)
    ref = build_reference_rec(cfg, sd, "sdpa")
    t0 = time.time()
    lg, bb, tk = run_reference(ref, cfg, tiles, grids, seqs, 4)
    print(f"rec256 fp32 reference: {time.time() - t0:.1f}s", flush=True)
    g = {"config": "REC-FULL", "attn": "sdpa", "lines": 256, "line_seed": 1234, "grids": grids,
         "tiles_sum": float(tiles.double().sum()), **pack(cfg, lg, bb, tk, 8)}
    torch.save(g, os.path.join(GOLD, "rec_full_bench256.pt"))


def small256_grids():
    rng = np.random.default_rng(77)
    return [crop_grid(64, int(w)) for w in sorted(rng.integers(128, 513, size=256), reverse=True)]


def small256():
    cfg = rec_config("REC-SMALL")
    sd = make_rec_weights(cfg, 0)
    grids = small256_grids()
    tiles, seqs = make_prompts(cfg, grids, seed=11)
    ref = build_reference_rec(cfg, sd, "sdpa")
    t0 = time.time()
    lg, bb, tk = run_reference(ref, cfg, tiles, grids, seqs, 12)
    print(f"small256 fp32 reference: {time.time() - t0:.1f}s", flush=True)
    g = {"config": "REC-SMALL", "attn": "sdpa", "grids": grids, "seed": 11, **pack(cfg, lg, bb, tk, 8)}
    lgb, _, _ = run_reference(ref.bfloat16(), cfg, tiles, grids, seqs, 12, forced=tk)
    g["bf16_dev"] = (lgb - lg).abs().amax(-1)
    torch.save(g, os.path.join(GOLD, "rec_small_256.pt"))


def det1024():
    from surya.detection.model.config import EfficientViTConfig
    from surya.detection.model.encoderdecoder import EfficientViTForSemanticSegmentation
    from oracle.det_oracle import normalise_pages
    c = det_config("DET-DEFAULT")
    rc = EfficientViTConfig(widths=c.widths, depths=c.depths, head_dim=c.head_dim,
                            decoder_layer_hidden_size=c.decoder_layer_hidden_size, decoder_hidden_size=c.decoder_hidden_size,
                            num_labels=c.num_labels)
    m = EfficientViTForSemanticSegmentation(rc).eval()
    m.load_state_dict(make_det_weights(c, 0), strict=True)
    pages = make_pages(16, 1024, seed=1234)[:1]                          # page 0 of bench.py's detection leg
    x = normalise_pages(pages)
    from surya.detection.processor import SegformerImageProcessor            # the reference's own rescale + normalise
    rp = SegformerImageProcessor(size={"height": 1024, "width": 1024})
    assert np.array_equal(rp(pages[0])["pixel_values"][0], x[0].numpy()), "normalise_pages != reference processor"
    with torch.inference_mode():
        t0 = time.time()
        out = m(pixel_values=x).logits
        print(f"det1024 fp32 reference: {time.time() - t0:.1f}s", flush=True)
        up = F.interpolate(out, size=(1024, 1024), mode="bilinear", align_corners=False)      # detection/__init__.py:121-129
        t0 = time.time()
        outb = m.bfloat16()(pixel_values=x.bfloat16()).logits.float()
        print(f"det1024 bf16 reference: {time.time() - t0:.1f}s", flush=True)
    g = {"config": "DET-DEFAULT", "size": 1024, "pages": 16, "page_seed": 1234, "page": 0, "logits": out.clone(),
         "upsampled_sample": up[:, :, ::4, ::4].clone(), "bf16_dev": float((outb - out).abs().max()),
         "bf16_dev_mean": float((outb - out).abs().mean())}
    torch.save(g, os.path.join(GOLD, "det_default_1024.pt"))


def main():
    os.makedirs(GOLD, exist_ok=True)
    which = sys.argv[1:] or ["rec8", "rec256", "small256", "det1024"]
    for w in which:
        {"rec8": rec8, "rec256": rec256, "small256": small256, "det1024": det1024, "rec8c": rec8c, "rec256c": rec256c}[w]()
    for f in sorted(os.listdir(GOLD)):
        print(f, os.path.getsize(os.path.join(GOLD, f)))


if __name__ == "__main__":
    main()
