"""FP8 KV cache of the decode steps (csrc/decode_attn_kv8.h; BASELINE.json configs[4]'s fp8 path beyond the weights).

No reference counterpart exists (surya has no fp8 mode), so the bounds are of two kinds:
  * bit-exact: the bytes and scales the kernels write == oracle/mx_oracle.py::kv8_quantize of the same bf16 rows (the format's
    definition, written independently in numpy integer arithmetic);
  * arithmetic: decode attention over the quantised cache vs an fp32 PyTorch attention over the DEQUANTISED cache, at the tolerance
    of the bf16 flash kernel's own test (2e-2 of max|ref|) -- dequantised values are exact in bf16, so the kernel adds no error of
    its own beyond the P rounding the bf16 kernel has too;
  * quantisation cost, stated: the same output vs attention over the UNquantised bf16 cache stays within 6e-2 of max|ref| on
    unit-variance K / V (e4m3 carries 3 mantissa bits: <= 6.25 % per element, averaged down by the softmax sum).
"""
import ctypes as C
import math

import numpy as np
import pytest
import torch

from oracle import mx_oracle as mo
from surya_amd import _lib as L
from oracle import rec_oracle as ro
from util import make_prompts, left_pad_batch
from test_gpu_attn_ops import _rope_table, _stream, DECODE_LENS
from test_gpu_rec import GRIDS, build, _oracle_run

pytestmark = pytest.mark.gpu


def _kv8_arrays(kc, vc, lens, slots, Tmax):
    """Quantise rows [0, len) of the given slots through surya_op_kv8_quant_rows; returns the device arrays."""
    n_slots, nkv, _, d = kc.shape
    T8 = (Tmax + 255) // 256 * 256
    k8 = torch.zeros(n_slots, nkv, Tmax, d, dtype=torch.uint8, device="cuda")
    v8t = torch.zeros(n_slots, nkv, T8 // 128, d, 128, dtype=torch.uint8, device="cuda")      # transposed inside each 128-token tile
    ks = torch.zeros(n_slots, nkv, T8, dtype=torch.float32, device="cuda")
    vs = torch.zeros(n_slots, nkv, T8, dtype=torch.float32, device="cuda")
    tok_slot = torch.tensor([int(s) for s, ln in zip(slots, lens) for _ in range(ln)], dtype=torch.int32, device="cuda")
    tok_pos = torch.tensor([p for ln in lens for p in range(ln)], dtype=torch.int32, device="cuda")
    if tok_slot.numel():
        rc = L.lib().surya_op_kv8_quant_rows(d, L.ptr(kc), L.ptr(vc), L.ptr(tok_slot), L.ptr(tok_pos), tok_slot.numel(), L.ptr(k8), L.ptr(v8t),
                                             L.ptr(ks), L.ptr(vs), nkv, Tmax, _stream())
        assert rc == 0, rc
    torch.cuda.synchronize()
    return k8, v8t, ks, vs


def _v_rows(v8t):
    """[slots, nkv, tiles, d, 128] -> [slots, nkv, T8, d] (token-major view of the tile-transposed V bytes)."""
    n, h, t, d, _ = v8t.shape
    return v8t.permute(0, 1, 2, 4, 3).reshape(n, h, t * 128, d)


def _case(lib, d, nq, nkv, lens, S, Tmax, seed, kv_gain=1.0):
    G, M = nq // nkv, len(lens)
    n_slots = M + 3
    dtype = torch.bfloat16
    g = torch.Generator(device="cuda").manual_seed(seed)
    qkv_d = (nq + 2 * nkv) * d
    slots = torch.randperm(n_slots, generator=torch.Generator().manual_seed(seed))[:M].to(torch.int32)
    part = torch.randn(S, M, qkv_d, device="cuda", generator=g) / math.sqrt(S)
    bias = (0.5 * torch.randn(qkv_d, device="cuda", generator=g)).to(dtype)
    # rows of very different magnitude: the per-row scale has to carry them
    gain = torch.exp2(torch.randint(-6, 7, (n_slots, nkv, Tmax, 1), device="cuda", generator=g).float()) * kv_gain
    kc = (torch.randn(n_slots, nkv, Tmax, d, device="cuda", generator=g) * gain).to(dtype)
    vc = (torch.randn(n_slots, nkv, Tmax, d, device="cuda", generator=g) * gain.flip(2)).to(dtype)
    k8, v8t, ks, vs = _kv8_arrays(kc, vc, lens, slots, Tmax)
    # (1) the prefill quantiser == the oracle's definition, bit for bit
    for r in range(M):
        ln, s = lens[r], int(slots[r])
        if ln == 0:
            continue
        qk, sk = mo.kv8_quantize(kc[s, :, :ln].float().cpu().numpy())
        qv, sv = mo.kv8_quantize(vc[s, :, :ln].float().cpu().numpy())
        assert np.array_equal(k8[s, :, :ln].cpu().numpy(), qk) and np.array_equal(ks[s, :, :ln].cpu().numpy(), sk)
        assert np.array_equal(_v_rows(v8t)[s, :, :ln].cpu().numpy(), qv) and np.array_equal(vs[s, :, :ln].cpu().numpy(), sv)
    k8_0, v8t_0 = k8.clone(), v8t.clone()
    cs = _rope_table(Tmax, d, dtype).cuda()
    out = torch.full((M, nq * d), float("nan"), device="cuda", dtype=dtype)
    act = slots.cuda()
    rl = torch.tensor(lens, dtype=torch.int32, device="cuda")
    scale = 1.0 / math.sqrt(d)
    rc = lib.surya_op_decode_attn_kv8(d, L.ptr(part), S, L.ptr(bias), L.ptr(out), L.ptr(k8), L.ptr(v8t), L.ptr(ks), L.ptr(vs), L.ptr(act),
                                      L.ptr(rl), L.ptr(cs), M, nq, nkv, Tmax, C.c_float(scale), _stream())
    assert rc == 0, rc
    torch.cuda.synchronize()
    assert not torch.isnan(out.float()).any()
    x = (part.sum(0) + bias.float()).to(dtype).float()
    qh = x[:, :nq * d].view(M, nq, d)
    kh = x[:, nq * d:(nq + nkv) * d].view(M, nkv, d)
    vh = x[:, (nq + nkv) * d:].view(M, nkv, d)
    half = d // 2
    worst_q, worst_u, ref_max = 0.0, 0.0, 1.0
    for r in range(M):
        ln, s = lens[r], int(slots[r])
        c, sn = cs[ln, :, 0], cs[ln, :, 1]

        def rope(t):
            t1, t2 = t[..., :half], t[..., half:]
            return torch.cat([(t1 * c - t2 * sn), (t2 * c + t1 * sn)], dim=-1).to(dtype).float()

        qr = (rope(qh[r]) * scale).to(dtype).float()
        kr = rope(kh[r])
        # (2) the appended row: bytes and scales == oracle quantisation of the roped k / the v row. The kernel sums the split-K slabs
        # in its own order, so a bf16 tie may round k differently from torch's sum: compare through the kernel's own dequantised row
        # against kr within one bf16 ulp + one e4m3 step instead of demanding equal bytes there.
        kq_new = mo.kv8_dequantize(k8[s, :, ln].cpu().numpy(), ks[s, :, ln].cpu().numpy())
        vq_new = mo.kv8_dequantize(_v_rows(v8t)[s, :, ln].cpu().numpy(), vs[s, :, ln].cpu().numpy())
        for got_row, want in ((kq_new, kr.cpu().numpy()), (vq_new, vh[r].cpu().numpy())):
            amax = np.abs(want).max(-1, keepdims=True)
            assert (np.abs(got_row - want) <= amax * (2.0 ** -3) * 0.5 + 1e-2).all()      # half an e4m3 step of the row's top binade + bf16 ulp
            qq, ss = mo.kv8_quantize(got_row)                                             # the stored row is a fixed point of the quantiser
            assert np.array_equal(mo.kv8_dequantize(qq, ss), got_row)
        # rows other than the appended one are untouched
        assert torch.equal(k8[s, :, :ln], k8_0[s, :, :ln]) and torch.equal(_v_rows(v8t)[s, :, ln + 1:], _v_rows(v8t_0)[s, :, ln + 1:])
        Kq = torch.from_numpy(mo.kv8_dequantize(k8[s, :, :ln + 1].cpu().numpy(), ks[s, :, :ln + 1].cpu().numpy())).cuda()
        Vq = torch.from_numpy(mo.kv8_dequantize(_v_rows(v8t)[s, :, :ln + 1].cpu().numpy(), vs[s, :, :ln + 1].cpu().numpy())).cuda()
        Ku = torch.cat([kc[s, :, :ln].float(), kr[:, None, :]], dim=1)
        Vu = torch.cat([vc[s, :, :ln].float(), vh[r][:, None, :]], dim=1)
        got = out[r].float().view(nq, d)
        for K_, V_, which in ((Kq, Vq, "q"), (Ku, Vu, "u")):
            sc = torch.einsum("hd,hkd->hk", qr, K_.repeat_interleave(G, dim=0))
            ref = torch.einsum("hk,hkd->hd", torch.softmax(sc, dim=-1), V_.repeat_interleave(G, dim=0))
            e = (got - ref).abs().max().item()
            if which == "q":
                worst_q = max(worst_q, e)
                ref_max = max(ref_max, ref.abs().max().item())
            else:
                worst_u = max(worst_u, e)
    return worst_q, worst_u, ref_max


@pytest.mark.parametrize("S", [1, 3, 8])
def test_decode_attn_kv8_d128_g5(hip_lib, S):
    wq, wu, ref_max = _case(hip_lib, 128, 10, 2, DECODE_LENS, S, 1024, seed=S)
    assert wq <= 2e-2 * ref_max, f"vs attention over the dequantised cache: {wq} (max|ref| {ref_max})"
    print(f"S={S}: err vs dequantised-cache attention {wq / ref_max:.2e}, vs unquantised bf16 cache {wu / ref_max:.2e} (x max|ref|)")


@pytest.mark.parametrize("d,nq,nkv,Tmax", [(128, 16, 2, 512), (64, 8, 2, 300), (32, 4, 2, 260)])
def test_decode_attn_kv8_other_shapes(hip_lib, d, nq, nkv, Tmax):
    wq, wu, ref_max = _case(hip_lib, d, nq, nkv, [0, 7, 64, 127, 128, 130, 257], 2, Tmax, seed=d)
    assert wq <= 2e-2 * ref_max, (wq, ref_max)


def test_decode_attn_kv8_quantisation_cost_on_unit_rows(hip_lib):
    """Unit-variance K / V (no per-row gain spread): the whole fp8 effect vs the bf16 cache."""
    torch.manual_seed(0)
    lens = [51, 202, 460, 969]
    G, M, d, nq, nkv, Tmax = 5, 4, 128, 10, 2, 1024
    dtype = torch.bfloat16
    g = torch.Generator(device="cuda").manual_seed(5)
    qkv_d = (nq + 2 * nkv) * d
    slots = torch.arange(M, dtype=torch.int32)
    part = torch.randn(1, M, qkv_d, device="cuda", generator=g)
    bias = torch.zeros(qkv_d, device="cuda", dtype=dtype)
    kc = torch.randn(M, nkv, Tmax, d, device="cuda", generator=g).to(dtype)
    vc = torch.randn(M, nkv, Tmax, d, device="cuda", generator=g).to(dtype)
    k8, v8t, ks, vs = _kv8_arrays(kc, vc, lens, slots, Tmax)
    cs = _rope_table(Tmax, d, dtype).cuda()
    out8 = torch.empty((M, nq * d), device="cuda", dtype=dtype)
    out16 = torch.empty_like(out8)
    act, rl = slots.cuda(), torch.tensor(lens, dtype=torch.int32, device="cuda")
    scale = 1.0 / math.sqrt(d)
    lib = hip_lib
    assert lib.surya_op_decode_attn_kv8(d, L.ptr(part), 1, L.ptr(bias), L.ptr(out8), L.ptr(k8), L.ptr(v8t), L.ptr(ks), L.ptr(vs), L.ptr(act),
                                        L.ptr(rl), L.ptr(cs), M, nq, nkv, Tmax, C.c_float(scale), _stream()) == 0
    assert lib.surya_op_decode_attn(L.DTYPE_BF16, d, L.ptr(part), 1, L.ptr(bias), L.ptr(out16), L.ptr(kc), L.ptr(vc), L.ptr(act), L.ptr(rl),
                                    L.ptr(cs), M, nq, nkv, Tmax, C.c_float(scale), _stream()) == 0
    torch.cuda.synchronize()
    err = (out8.float() - out16.float()).abs().max().item()
    ref = out16.float().abs().max().item()
    print(f"kv8 vs bf16 cache on unit rows: {err:.3e} of max {ref:.3e}")
    assert err <= 6e-2 * max(1.0, ref)


def test_rec_small_kv8_decode_teacher_forced(hip_lib):
    """REC-SMALL decode steps on the fp8 KV cache, teacher-forced, vs the fp32 oracle WITH the same quantisation emulated
    (rec_oracle.KV8_DECODE): within 2 x the reference's own bf16 deviation + half the format's deviation + 1e-2 x max|logit| -- the
    bound of the MXFP8 weight test (tests/test_gpu_mx.py); the GPU quantises bf16-rounded rows, which flips a few 3-bit roundings."""
    cfg, sd, m = build("REC-SMALL", torch.bfloat16)
    m.set_kv_fp8(True)
    tiles, seqs = make_prompts(cfg, GRIDS)
    T = 10
    toks_ref, _, _, logits_ref = _oracle_run(cfg, sd, tiles, seqs, T)
    ids, am, pos = left_pad_batch(cfg, seqs)
    grids = [(1, h, w) for h, w in GRIDS]
    ob = ro.OracleRecModel(cfg, {k: v.bfloat16() for k, v in sd.items()}, cfg.image_token_id)
    logits_b16 = ro.teacher_forced_logits(ob, ids, tiles, grids, am, pos, toks_ref, cfg.pad_token_id)
    ro.KV8_DECODE = True
    try:
        om = ro.OracleRecModel(cfg, sd, cfg.image_token_id)
        logits_kv = ro.teacher_forced_logits(om, ids, tiles, grids, am, pos, toks_ref, cfg.pad_token_id)
    finally:
        ro.KV8_DECODE = False
    slots = list(range(len(seqs)))
    m.prefill(tiles.cuda(), GRIDS, seqs, slots)
    m.set_active(slots)
    rep = []
    for step in range(min(T, len(logits_ref))):
        lg = m.last_logits().cpu()
        live = [i for i in range(len(seqs)) if step < len(toks_ref[i])]
        ref, emu = logits_ref[step][live], logits_kv[step][live]
        scale = ref.abs().max().item()
        b16_dev = (logits_b16[step][live] - ref).abs().max().item()
        kv_dev = (emu - ref).abs().max().item()
        err_emu = (lg[live] - emu).abs().max().item()
        err_ref = (lg[live] - ref).abs().max().item()
        rep.append((step, err_emu / scale, kv_dev / scale, b16_dev / scale, err_ref / scale))
        if step == 0:
            assert kv_dev == 0.0                            # prefill attends over the unquantised rows
        assert err_emu <= 2 * b16_dev + 0.5 * kv_dev + 1e-2 * scale, rep[-1]
        assert err_ref <= 2 * b16_dev + 1.5 * kv_dev + 1e-2 * scale, rep[-1]
        m.set_next_tokens(slots, [toks_ref[i][step] if step < len(toks_ref[i]) else cfg.pad_token_id for i in slots])
        m.decode(1)
        m.read_outputs(1)
    print("fp8 KV REC-SMALL (step, |gpu - emulation|, |emulation - fp32|, |bf16 ref - fp32|, |gpu - fp32|) / max|logit|:")
    for r in rep:
        print("   %d  %.4f  %.4f  %.4f  %.4f" % r)
    assert max(r[2] for r in rep[1:]) > 0                   # the fp8 cache really was read


def test_kv8_switch_restores_bf16_results_and_combines_with_fp8_weights(hip_lib):
    cfg, sd, m = build("REC-SMALL", torch.bfloat16)
    tiles, seqs = make_prompts(cfg, GRIDS)
    slots = list(range(len(seqs)))

    first = []                                                      # the token each run's PREFILL picks (before any decode step)

    def run():
        m.prefill(tiles.cuda(), GRIDS, seqs, slots)
        t0, _, _ = m.read_outputs(1)
        first.append(t0[0][: len(slots)].copy())
        m.set_active(slots)
        m.decode(6)
        t, s, b = m.read_outputs(6)
        return t[:, : len(slots)].copy(), s[:, : len(slots)].copy(), b[:, : len(slots)].copy()

    a = run()
    m.set_kv_fp8(True)
    f = run()
    m.set_decode_fp8(True)
    both = run()
    m.set_decode_fp8(False)
    m.set_kv_fp8(False)
    c = run()
    assert all(np.array_equal(x, y) for x, y in zip(a, c))
    assert not np.array_equal(a[1], f[1]) and not np.array_equal(f[1], both[1])
    for r in (f, both):
        assert np.isfinite(r[1]).all() and (r[0] >= 0).all() and (r[0] < cfg.decoder.vocab_size).all()
    # the token the prefill picks comes from the prefill logits (bf16 cache in every mode): identical in all four runs. (Round 4 asserted
    # this on the first DECODE step's token, which already attends over the fp8 rows of the prompt -- equal only while no slot sits on a
    # near-tie; round 5's cheaper SiLU moved slot 1 onto one.)
    assert all(np.array_equal(first[0], x) for x in first[1:])


def test_scheduler_with_slot_reuse_on_the_fp8_cache(hip_lib):
    """The continuous-batching loop (more lines than slots: slots are refilled while others keep decoding, stale fp8 rows of the
    previous owner stay behind the new context) with the FP8 KV cache: a line's tokens must not depend on the batch size or on which
    slot it lands in, every first token (computed from the prefill, bf16 cache in every mode) equals the bf16 run's, and switching the
    cache off restores the bf16 streams exactly."""
    from surya_amd.config import rec_config
    from surya_amd.recognition.predictor import RecognitionPredictor, RecognitionModelLoader
    from surya_amd.recognition.schema import TaskNames
    from surya_amd.settings import settings
    from surya_amd.synth import make_line_crops, make_rec_weights
    cfg = rec_config("REC-SMALL")
    sd = make_rec_weights(cfg, 0)

    class Loader(RecognitionModelLoader):
        def model(self, device=None, dtype_=None, **caps):
            return super().model("cuda:0", torch.bfloat16, max_slots=6, max_kv_len=300, max_patches=16384, max_prefill_tokens=4096)

    class Pred(RecognitionPredictor):
        model_loader_cls = Loader
        batch_size = 6

    old = settings.RECOGNITION_MAX_TOKENS
    settings.RECOGNITION_MAX_TOKENS = 24
    try:
        pred = Pred(checkpoint={"config": cfg, "state_dict": sd})
        crops = [c.astype(np.float32) for c in make_line_crops(17, seed=8)]
        crops.sort(key=lambda c: -c.shape[1])
        flat = {"slices": crops, "input_text": [None] * len(crops), "task_names": [TaskNames.ocr_with_boxes] * len(crops)}
        prep = pred.prepare_lines(flat, math_mode=True)
        base, _, _ = pred.generate(prep, 6)
        pred.model.set_kv_fp8(True)
        a, _, sa = pred.generate(prep, 6)
        b, _, _ = pred.generate(prep, 4)                       # other slot assignment, other refill pattern
        pred.model.set_kv_fp8(False)
        again, _, _ = pred.generate(prep, 6)
    finally:
        settings.RECOGNITION_MAX_TOKENS = old
    assert [list(t) for t in again] == [list(t) for t in base]             # switching back restores the bf16 streams
    assert [list(t) for t in a] == [list(t) for t in b]                    # scheduling-invariant under the fp8 cache too
    assert all(len(t) > 0 and t[0] == u[0] for t, u in zip(a, base))
    assert all(np.isfinite(np.asarray(s, np.float64)).all() for s in sa)
    same = sum(int(x == y) for t, u in zip(a, base) for x, y in zip(t, u))
    total = sum(min(len(t), len(u)) for t, u in zip(a, base))
    # informative only: on random REC-SMALL weights every rounding flips near-ties and a flipped token changes the rest of the line
    # (the arithmetic itself is bounded teacher-forced above)
    print(f"fp8 KV vs bf16 cache, free running: {same}/{total} tokens equal")
