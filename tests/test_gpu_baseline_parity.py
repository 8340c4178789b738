"""GPU: the BASELINE.json configurations themselves against the REAL reference (fixtures of oracle/make_golden_full.py).

Until round 2 the two benchmark configurations were only compared with themselves (scheduling / batch invariance). Here
the HIP path runs bench.py's own workload and is held to what the reference's SuryaModel /
EfficientViTForSemanticSegmentation computed for it on the CPU (fp32), through the C ABI:

  REC-FULL, 8 of the bench's 256 crops, 48 tokens   fp32 mode: token ids bit-exact, bbox ints bit-exact (except at
                                                    truncation boundaries), scores, top-32 logits + logsumexp <= 2e-4 x max;
                                                    bf16: teacher-forced logits within 2 x the reference's OWN bf16
                                                    deviation + 1e-2 x max, argmax equal where the margin exceeds 4 x that
  REC-FULL, all 256 bench crops, prefill + 3 steps   the M = 256 launches the bench times: 64x64 split-K with 4 M-tiles per
                                                    XCD group, 128x128 fused-argmax lm_head, 256-row flash decode attention
  REC-SMALL, 256 ragged prompts, 12 steps            same launch shapes, deeper, with its own bf16 deviation
  DET-DEFAULT, one 1024^2 bench page                 fp32 <= 1e-4 on the [0, 1] maps; bf16 <= max(3e-2, 2 x reference bf16 dev)

Tolerances follow SURVEY 8(d). The fixtures were produced by the reference modules themselves, so no oracle runs here.
"""
import os

import numpy as np
import pytest
import torch

from surya_amd.config import rec_config, det_config
from surya_amd.synth import make_rec_weights, make_det_weights, make_pages
from util import bench_line_inputs, make_prompts

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _model(cfg_name, dtype, slots, max_kv=160):
    from surya_amd.recognition.model import HipRecModel
    cfg = rec_config(cfg_name)
    sd = make_rec_weights(cfg, 0)
    return cfg, HipRecModel(cfg, sd, image_token_id=cfg.image_token_id, pad_token_id=cfg.pad_token_id,
                            eos_token_id=cfg.eos_token_id, dtype=dtype, device="cuda:0", max_slots=slots, max_kv_len=max_kv,
                            max_patches=65536, max_prefill_tokens=slots * 72)


@pytest.fixture(scope="module")
def full_fp32(hip_lib):
    return _model("REC-FULL", torch.float32, 256)


@pytest.fixture(scope="module")
def full_bf16(hip_lib):
    return _model("REC-FULL", torch.bfloat16, 256)


@pytest.fixture(scope="module")
def bench_inputs():
    cfg = rec_config("REC-FULL")
    return bench_line_inputs(cfg, 256, seed=1234)


def _subset(inputs, pick):
    tiles, grids, seqs = inputs
    offs = np.cumsum([0] + [h * w for h, w in grids])
    t = torch.cat([tiles[offs[i]:offs[i + 1]] for i in pick])
    return t, [grids[i] for i in pick], [seqs[i] for i in pick]


def _check_inputs(g, tiles, grids):
    assert [tuple(x) for x in g["grids"]] == [tuple(x) for x in grids]
    # host pre-processing is deterministic numpy; allow for a different BLAS / FMA contraction on another CPU
    assert abs(float(tiles.double().sum()) - g["tiles_sum"]) <= 1e-6 * abs(g["tiles_sum"]) + 1e-3


def _free_run_fp32(cfg, m, g, tiles, grids, seqs, steps, topk):
    """Greedy free-running decode in fp32 mode against the reference's stream."""
    n = len(seqs)
    slots = list(range(n))
    m.prefill(tiles.cuda().contiguous(), grids, seqs, slots)
    alive = np.ones(n, bool)
    worst, flips, checked, ties = 0.0, 0, 0, 0
    for step in range(steps):
        if step == 1:
            m.set_active(slots)
        if step:
            m.decode(1)
        tok, sc, bb = m.read_outputs(1)
        lg = m.last_logits().cpu()
        ref_tok = g["tokens"][step].numpy()
        idx, val = g["logits_top"]["indices"][step], g["logits_top"]["values"][step]
        scale = float(g["logits_absmax"][step].max())
        live = np.nonzero(alive)[0]
        err = (torch.gather(lg, -1, idx) - val).abs()[live].max().item()
        lse_err = (torch.logsumexp(lg, -1) - g["logits_lse"][step]).abs()[live].max().item()
        worst = max(worst, err / scale, lse_err / scale)
        assert err <= 2e-4 * scale and lse_err <= 2e-4 * scale, (step, err, lse_err, scale)
        got_tok = tok[0][slots]
        for i in live:
            if got_tok[i] != ref_tok[i]:
                # the only legal token difference in fp32 mode: the reference's own top-2 margin is inside the fp32 re-ordering
                # noise (< 1e-4 x max|logit|; the fixtures hold 1-2 such positions per 1000). The line leaves the comparison.
                margin = float(val[i, 0] - val[i, 1])
                assert margin < 1e-4 * scale and int(got_tok[i]) == int(idx[i, 1]), (step, i, margin, got_tok[i], ref_tok[i])
                alive[i] = False
                ties += 1
        live = np.nonzero(alive)[0]
        raw, ints = g["bbox_raw"][step].numpy(), g["bbox_ints"][step].numpy()
        for i in live:
            for k in range(6):
                checked += 1
                if bb[0, i, k] != ints[i, k]:        # (sigmoid x 1025).long() is a truncation: +-1 only next to an integer
                    assert abs(raw[i, k] - round(raw[i, k])) < 2e-2 and abs(int(bb[0, i, k]) - int(ints[i, k])) == 1, (step, i, k, raw[i, k])
                    flips += 1
        done = np.isin(ref_tok, [cfg.eos_token_id, cfg.pad_token_id])
        ok = alive & ~done
        assert np.allclose(sc[0][slots][ok], g["scores"][step].numpy()[ok], rtol=2e-3, atol=1e-7)
        alive &= ~done                                # the device feeds <PAD> after eos; the fixture kept feeding the argmax
    # each flip was verified above to sit within 2e-2 of an integer; fp32 re-ordering noise of ~3e-6 x 1025 straddles one in <1 % of values
    assert flips <= max(2, checked // 100), (flips, checked)
    assert ties <= max(1, n // 64), ties
    return worst, flips, int(alive.sum())


DEV_CAP = 0.05      # a bf16 tolerance built on the reference's own bf16 deviation means something only while that deviation is small


def _teacher_forced_bf16(cfg, m, g, tiles, grids, seqs, steps, dev_per_step):
    """Teacher-forced bf16 logits within 2 x (the reference's own bf16 deviation) + 1e-2 x max|logit|. VERDICT r03: on the default
    synthetic weights that deviation is 12-21 % of max|logit| -- a tolerance of ~35 % accepts almost anything. The fixture, not the
    kernel, is then unfit for a parity claim: where the recorded deviation exceeds DEV_CAP x max|logit| the check degrades to an
    ENVELOPE (stated as such in the printed result and in the test's name: finite logits, inside the reference's own rounding
    envelope); the bf16 parity claim lives in tests/test_gpu_bf16_parity.py (conditioned weights, same shapes,
    deviation <= 1.8 %, enforced there with the same cap)."""
    unfit = float(max(float(dev_per_step[s_]) / float(g["logits_absmax"][s_].max()) for s_ in range(steps)))
    n = len(seqs)
    slots = list(range(n))
    m.prefill(tiles.cuda().contiguous(), grids, seqs, slots)
    m.set_active(slots)
    worst, worst_ref, mism, checked = 0.0, 0.0, 0, 0
    for step in range(steps):
        lg = m.last_logits().cpu()
        idx, val = g["logits_top"]["indices"][step], g["logits_top"]["values"][step]
        scale = float(g["logits_absmax"][step].max())
        ref_dev = float(dev_per_step[step])
        tol = 2 * ref_dev + 1e-2 * scale
        err = (torch.gather(lg, -1, idx) - val).abs().max().item()
        lse_err = (torch.logsumexp(lg, -1) - g["logits_lse"][step]).abs().max().item()
        worst, worst_ref = max(worst, err / scale), max(worst_ref, ref_dev / scale)
        assert err <= tol and lse_err <= tol, (step, err, lse_err, ref_dev, scale)
        margin = val[:, 0] - val[:, 1]
        am = lg.argmax(-1)
        for i in range(n):
            if margin[i].item() > 4 * tol:
                checked += 1
                mism += int(am[i].item() != int(g["tokens"][step][i]))
        if step + 1 < steps:
            m.set_next_tokens(slots, g["tokens"][step].tolist())
            m.decode(1)
    assert mism == 0, (mism, checked)
    if unfit > DEV_CAP:
        # the assertions above RAN and held; what they prove on this weight set is an envelope, not parity -- said in the test's name
        # (`..._envelope`) and in its printed line, not by a skip (a skip after the assertions reads as "did not run")
        print(f"ENVELOPE ONLY, no parity claim: the reference's own bf16 run deviates {unfit:.1%} of max|logit| on this weight set "
              f"(cap {DEV_CAP:.0%}); HIP bf16 worst {worst:.4f} stayed inside 2 x that + 1 %. Parity of the bf16 path: tests/test_gpu_bf16_parity.py")
    return worst, worst_ref, checked


def test_rec_full_bench8_fp32_bit_exact(full_fp32, bench_inputs):
    g = torch.load(os.path.join(GOLD, "rec_full_bench8.pt"))
    cfg, m = full_fp32
    tiles, grids, seqs = _subset(bench_inputs, g["pick"])
    _check_inputs(g, tiles, grids)
    worst, flips, alive = _free_run_fp32(cfg, m, g, tiles, grids, seqs, g["tokens"].shape[0], 32)
    print(f"REC-FULL 8 bench crops x {g['tokens'].shape[0]} tokens, fp32 mode vs the reference: tokens bit-exact, worst logit err "
          f"{worst:.2e} x max, {flips} bbox truncation flips, {alive} lines alive at the end")


def test_rec_full_bench8_bf16_teacher_forced_envelope(full_bf16, bench_inputs):
    g = torch.load(os.path.join(GOLD, "rec_full_bench8.pt"))
    cfg, m = full_bf16
    tiles, grids, seqs = _subset(bench_inputs, g["pick"])
    worst, worst_ref, checked = _teacher_forced_bf16(cfg, m, g, tiles, grids, seqs, g["tokens"].shape[0], g["bf16_dev"].amax(-1))
    print(f"REC-FULL 8 bench crops bf16 teacher-forced: worst rel logit err {worst:.4f} (reference's own bf16 path {worst_ref:.4f}), "
          f"{checked} argmax positions checked, 0 mismatches")


def test_rec_full_bench256_fp32_bit_exact(full_fp32, bench_inputs):
    """All 256 bench lines at once: the launch shapes bench.py times (M = 256)."""
    g = torch.load(os.path.join(GOLD, "rec_full_bench256.pt"))
    cfg, m = full_fp32
    tiles, grids, seqs = bench_inputs
    _check_inputs(g, tiles, grids)
    worst, flips, alive = _free_run_fp32(cfg, m, g, tiles, grids, seqs, g["tokens"].shape[0], 8)
    print(f"REC-FULL 256 bench crops x {g['tokens'].shape[0]} steps, fp32 mode vs the reference: tokens bit-exact, worst logit err "
          f"{worst:.2e} x max, {flips} bbox truncation flips")


def test_rec_full_bench256_bf16_teacher_forced_envelope(full_bf16, bench_inputs):
    g = torch.load(os.path.join(GOLD, "rec_full_bench256.pt"))
    g8 = torch.load(os.path.join(GOLD, "rec_full_bench8.pt"))
    cfg, m = full_bf16
    tiles, grids, seqs = bench_inputs
    steps = g["tokens"].shape[0]
    # the reference's bf16 deviation was recorded on the 8-line sample of the same workload (same steps)
    worst, worst_ref, checked = _teacher_forced_bf16(cfg, m, g, tiles, grids, seqs, steps, g8["bf16_dev"][:steps].amax(-1))
    print(f"REC-FULL 256 bench crops bf16 teacher-forced: worst rel logit err {worst:.4f} (reference bf16 {worst_ref:.4f}), "
          f"{checked} argmax positions checked, 0 mismatches")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_rec_small_256_lines(hip_lib, dtype):
    g = torch.load(os.path.join(GOLD, "rec_small_256.pt"))
    cfg, m = _model("REC-SMALL", dtype, 256)
    grids = [tuple(x) for x in g["grids"]]
    tiles, seqs = make_prompts(cfg, grids, seed=g["seed"])
    steps = g["tokens"].shape[0]
    if dtype == torch.float32:
        worst, flips, alive = _free_run_fp32(cfg, m, g, tiles, grids, seqs, steps, 8)
        print(f"REC-SMALL 256 lines x {steps} steps fp32: tokens bit-exact, worst logit err {worst:.2e} x max, {flips} bbox flips")
    else:
        worst, worst_ref, checked = _teacher_forced_bf16(cfg, m, g, tiles, grids, seqs, steps, g["bf16_dev"].amax(-1))
        print(f"REC-SMALL 256 lines bf16 teacher-forced: worst rel logit err {worst:.4f} (reference bf16 {worst_ref:.4f}), "
              f"{checked} argmax positions checked")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_det_default_1024_vs_reference(hip_lib, dtype):
    from surya_amd.detection.model import HipDetModel
    from oracle.det_oracle import normalise_pages            # input normalisation only (host logic)
    g = torch.load(os.path.join(GOLD, "det_default_1024.pt"))
    cfg = det_config("DET-DEFAULT")
    m = HipDetModel(cfg, make_det_weights(cfg, 0), height=1024, width=1024, dtype=dtype, device="cuda:0", max_batch=2)
    x = normalise_pages(make_pages(g["pages"], 1024, seed=g["page_seed"])[g["page"]:g["page"] + 1]).cuda().contiguous()
    heat, low = m.forward(x, want_lowres=True)
    tol = 1e-4 if dtype == torch.float32 else max(3e-2, 2 * g["bf16_dev"])
    err_low = (low.cpu() - g["logits"]).abs().max().item()
    err_up = (heat.cpu()[:, :, ::4, ::4] - g["upsampled_sample"]).abs().max().item()
    print(f"DET-DEFAULT 1024^2 {dtype}: max |heat - reference| low-res {err_low:.2e}, x4 upsampled {err_up:.2e} "
          f"(tol {tol:.1e}; reference's own bf16 deviation {g['bf16_dev']:.2e})")
    assert err_low <= tol and err_up <= tol
