"""GPU: the round-4 kernels of the decode step against the round-3 kernels they replace (same model, knobs flipped through
surya_set_tuning inside one process), and the hipGraph cache against mode changes.

  * greedy_head2_kernel (+ the next step's embedding fused into it) vs greedy_head_kernel + embed_slots_norm_kernel: the argmax is
    exact in both, so tokens must be identical; scores are the same sum in a different tree (fp32 rounding); a bbox coordinate is
    trunc(sigmoid(bf16(dot + b)) * 1025) with the dot product summed in another order, so single coordinates may move by one at a
    rounding boundary -- counted and bounded.
  * split-K reduce with slab loads sized by the slice count vs the 8-slab kernel: bit-identical by construction.
  * toggling the fp8 KV cache / MXFP8 weights / any knob while graph replay is on must not replay a graph
    captured under the other mode.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from surya_amd import _lib as L
from surya_amd.config import rec_config
from surya_amd.synth import make_rec_weights
from util import make_prompts

pytestmark = pytest.mark.gpu

GRIDS = [(6, 38), (10, 18), (8, 24), (6, 10), (12, 12), (2, 30)]
DEFAULTS = dict(graph=0, dattn=4, rnorm=2, ghead=2, fuse_embed=1, persist=1, lmhead=1, kvprefetch=0, dattn_db=0)


def tune(**kw):
    for k, v in kw.items():
        L.check(L.lib().surya_set_tuning(k.encode(), C.c_int(int(v))), f"surya_set_tuning({k})")


@pytest.fixture(autouse=True)
def _restore_tuning(hip_lib):
    yield
    tune(**DEFAULTS)


def build(cfg_name, dtype, max_slots=8, max_kv_len=256):
    from surya_amd.recognition.model import HipRecModel
    cfg = rec_config(cfg_name)
    sd = make_rec_weights(cfg, 0)
    m = HipRecModel(cfg, sd, image_token_id=cfg.image_token_id, pad_token_id=cfg.pad_token_id, eos_token_id=cfg.eos_token_id,
                    dtype=dtype, max_slots=max_slots, max_kv_len=max_kv_len, max_patches=4096, max_prefill_tokens=1024)
    return cfg, m


def run_steps(m, cfg, calls, grids=GRIDS):
    tiles, seqs = make_prompts(cfg, grids)
    slots = list(range(len(seqs)))
    m.prefill(tiles.cuda(), grids, seqs, slots)
    t0, s0, b0 = m.read_outputs(1)
    m.set_active(slots)
    toks, scs, bbs = [t0[0][slots].copy()], [s0[0][slots].copy()], [b0[0][slots].copy()]
    for n in calls:
        m.decode(n)
        t, s, b = m.read_outputs(n)
        for k in range(n):
            toks.append(t[k][slots].copy()); scs.append(s[k][slots].copy()); bbs.append(b[k][slots].copy())
    return np.stack(toks), np.stack(scs), np.stack(bbs)


@pytest.mark.parametrize("cfg_name,dtype", [("REC-TINY", torch.float32), ("REC-SMALL", torch.bfloat16), ("REC-SMALL", torch.float32)])
def test_head2_and_fused_embedding_equal_round3_kernels(hip_lib, cfg_name, dtype):
    cfg, m = build(cfg_name, dtype)
    calls = [4, 1, 3, 4]
    tune(ghead=1, fuse_embed=0)
    ref = run_steps(m, cfg, calls)
    for knobs in (dict(ghead=2, fuse_embed=0), dict(ghead=2, fuse_embed=1)):
        tune(**knobs)
        got = run_steps(m, cfg, calls)
        assert np.array_equal(got[0], ref[0]), knobs                       # tokens: exact
        assert np.allclose(got[1], ref[1], rtol=1e-5, atol=1e-7), knobs    # scores: same sum, other tree
        diff = np.abs(got[2].astype(np.int64) - ref[2].astype(np.int64))
        assert diff.max() <= 1 and (diff > 0).mean() <= 0.02, (knobs, diff.max(), (diff > 0).mean())
    # fused inner steps == the same steps as single-step calls (standalone embedding kernel every step), bit for bit
    tune(ghead=2, fuse_embed=1)
    one = run_steps(m, cfg, [1] * sum(calls))
    many = run_steps(m, cfg, calls)
    assert np.array_equal(one[0], many[0]) and np.array_equal(one[2], many[2]) and np.array_equal(one[1], many[1])


def test_reduce_norm_slab_count_variants_are_bit_identical(hip_lib):
    cfg, m = build("REC-SMALL", torch.bfloat16)
    tune(rnorm=1)
    a = run_steps(m, cfg, [4, 4])
    tune(rnorm=2)
    b = run_steps(m, cfg, [4, 4])
    assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_decode_attention_kernels_agree_inside_the_model(hip_lib):
    """dattn 3 vs 4 on REC-SMALL bf16: same tokens wherever the round-3 run is not at a near-tie; scores close."""
    cfg, m = build("REC-SMALL", torch.bfloat16)
    tune(dattn=3)
    a = run_steps(m, cfg, [4, 4, 4])
    tune(dattn=4)
    b = run_steps(m, cfg, [4, 4, 4])
    same = (a[0] == b[0]).mean()
    print(f"dattn 3 vs 4: {same:.3f} of tokens equal")
    assert same >= 0.9
    eq = a[0] == b[0]
    assert np.allclose(a[1][eq], b[1][eq], rtol=5e-2, atol=1e-3)


def test_two_buffer_decode_attention_is_picked_by_context_and_changes_nothing(hip_lib):
    """Contexts that cross one 128-key tile during the run (prompts of ~20 ... ~150 tokens + 14 steps): the model switches to the
    two-buffer decode attention by its host-side context bound (dattn_db = 0); never (-1) and always (1) must give the same tokens,
    scores and boxes bit for bit -- also with hipGraph replay on, where the bound is not consulted."""
    cfg, m = build("REC-SMALL", torch.bfloat16, max_kv_len=320)
    grids = [(12, 44), (14, 40), (6, 10), (16, 36), (10, 48)]
    calls = [4, 4, 1, 4, 1]
    runs = {}
    for db in (-1, 0, 1):
        tune(dattn_db=db)
        runs[db] = run_steps(m, cfg, calls, grids)
    for db in (0, 1):
        for x, y in zip(runs[-1], runs[db]):
            assert np.array_equal(x, y), db
    tune(dattn_db=0, graph=1)
    for rep in range(3):
        g = run_steps(m, cfg, calls, grids)
        for x, y in zip(runs[-1], g):
            assert np.array_equal(x, y), rep


def test_graph_cache_is_dropped_when_a_mode_changes(hip_lib):
    """set_kv_fp8 / set_decode_fp8 / a tuning knob flipped while hipGraph replay is on. Every run below must equal the same
    mode's run without graphs; before the fix the toggled runs replayed graphs captured with the other attention kernel."""
    cfg, m = build("REC-SMALL", torch.bfloat16)
    calls = [4, 4, 4]

    def modes():
        out = {}
        out["bf16"] = run_steps(m, cfg, calls)
        m.set_kv_fp8(True)
        out["kv8"] = run_steps(m, cfg, calls)
        m.set_decode_fp8(True)
        out["kv8+mx"] = run_steps(m, cfg, calls)
        m.set_decode_fp8(False)
        m.set_kv_fp8(False)
        out["bf16 again"] = run_steps(m, cfg, calls)
        tune(dattn=3)
        out["dattn3"] = run_steps(m, cfg, calls)
        tune(dattn=4)
        return out

    tune(graph=0)
    plain = modes()
    tune(graph=1)
    for rep in range(3):            # first sight runs eagerly, the second captures, the third replays
        g = modes()
        for k in plain:
            for x, y in zip(plain[k], g[k]):
                assert np.array_equal(x, y), (rep, k)


# ------------------------------------------------------------------------------------------------- persistent 256x256 GEMM
def _op_gemm(lib, x, w, bias, epi, res=None):
    M, K = x.shape
    N = w.shape[0]
    No = N // 2 if epi == L.EPI_SWIGLU else N
    c = torch.full((M, No), float("nan"), dtype=torch.bfloat16, device=x.device)
    rc = lib.surya_op_gemm(L.DTYPE_BF16, 0, epi, L.ptr(x), C.c_long(x.stride(0)), L.ptr(w), C.c_long(w.stride(0)), L.ptr(c), C.c_long(No),
                           L.ptr(bias), L.ptr(res), C.c_long(No if res is not None else 0), M, N, K,
                           C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc
    torch.cuda.synchronize()
    return c


# (M, N, K): encoder / prefill shapes of the bench (ragged M, N = 5 / 15 / 27 / 40 tile columns, 20 and 54 K-tiles), and a ragged N
PERSIST_SHAPES = [(46460, 1280, 1280), (46460, 3840, 1280), (15360, 10240, 1280), (46460, 1280, 3456), (33000, 2064, 256), (16700, 4096, 128)]


@pytest.mark.parametrize("epi", [L.EPI_BIAS, L.EPI_RESIDUAL, L.EPI_GELU, L.EPI_SWIGLU])
@pytest.mark.parametrize("M,N,K", PERSIST_SHAPES)
def test_persistent_tile_loop_is_bit_identical_to_one_tile_per_workgroup(hip_lib, epi, M, N, K):
    """Same K order, MFMA order and epilogue arithmetic: the persistent 256x256 loop must reproduce the one-tile kernel's bits
    (which tests/test_gpu_ops.py pins to fp32 PyTorch), including ragged last tile rows / columns and the in-place residual."""
    g = torch.Generator(device="cuda").manual_seed(M + N + K + epi)
    x = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda", generator=g).to(torch.bfloat16) if epi != L.EPI_RESIDUAL else None
    res = torch.randn(M, N, device="cuda", generator=g).to(torch.bfloat16) if epi == L.EPI_RESIDUAL else None
    tune(persist=0)
    ref = _op_gemm(hip_lib, x, w, b, epi, res)
    tune(persist=1)
    got = _op_gemm(hip_lib, x, w, b, epi, res)
    assert not torch.isnan(got.float()).any()
    assert torch.equal(got.view(torch.int16), ref.view(torch.int16)), (got.float() - ref.float()).abs().max().item()
    if epi == L.EPI_RESIDUAL:                       # in place, as the encoder calls it (C aliases R)
        c = res.clone()
        rc = hip_lib.surya_op_gemm(L.DTYPE_BF16, 0, epi, L.ptr(x), C.c_long(K), L.ptr(w), C.c_long(K), L.ptr(c), C.c_long(N), None, L.ptr(c),
                                   C.c_long(N), M, N, K, C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
        torch.cuda.synchronize()
        assert torch.equal(c.view(torch.int16), ref.view(torch.int16))


def test_encoder_features_do_not_depend_on_the_persistent_loop(hip_lib):
    """The vision qkv projection's rotary epilogue (EPI_ROPE) is reachable through the encoder only: REC-FULL bf16 image features of 96
    bench-sized lines, persistent loop on vs off, bit for bit."""
    from surya_amd.recognition.model import HipRecModel
    from util import crop_grid
    cfg = rec_config("REC-FULL")
    sd = make_rec_weights(cfg, 0)
    m = HipRecModel(cfg, sd, image_token_id=cfg.image_token_id, pad_token_id=cfg.pad_token_id, eos_token_id=cfg.eos_token_id,
                    dtype=torch.bfloat16, max_slots=8, max_kv_len=128, max_patches=65536, max_prefill_tokens=96 * 72)
    rng = np.random.default_rng(7)
    grids = [crop_grid(64, int(w)) for w in sorted(rng.integers(128, 513, size=96), reverse=True)]
    tiles, _ = make_prompts(cfg, grids, seed=5)
    tiles = tiles.cuda().contiguous()
    tune(persist=0)
    a = m.encode_only(tiles, grids).clone()
    tune(persist=1)
    b = m.encode_only(tiles, grids)
    assert torch.equal(a.view(torch.int16), b.view(torch.int16))


def test_kv_prefetch_workgroups_change_nothing(hip_lib):
    """The extra workgroups of the reduce kernels only request cache lines: tokens, scores and boxes with kvprefetch on == off."""
    cfg, m = build("REC-SMALL", torch.bfloat16)
    tune(kvprefetch=0)
    a = run_steps(m, cfg, [4, 4, 3])
    tune(kvprefetch=1)
    b = run_steps(m, cfg, [4, 4, 3])
    assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_unconsumed_look_ahead_can_be_discarded(hip_lib):
    """surya_rec_encode_ahead(n_images = 0): a second look-ahead batch is refused (SA_ERR_STATE) while the first is unconsumed, accepted
    after a discard, and the prefill that consumes it gives the tokens of a prefill fed the tiles directly."""
    cfg, m = build("REC-SMALL", torch.bfloat16)
    tiles, seqs = make_prompts(cfg, GRIDS)
    tiles = tiles.cuda()
    slots = list(range(len(seqs)))
    m.prefill(tiles, GRIDS, seqs, slots)
    ref = m.read_outputs(1)[0][0][slots].copy()
    m.encode_ahead(tiles, GRIDS)
    with pytest.raises(L.SuryaAmdError):
        m.encode_ahead(tiles, GRIDS)
    m.discard_ahead()
    m.encode_ahead(tiles, GRIDS)
    m.prefill(None, GRIDS, seqs, slots)
    got = m.read_outputs(1)[0][0][slots].copy()
    assert np.array_equal(ref, got)
    m.discard_ahead()                                    # nothing outstanding: a no-op
    m.encode_ahead(tiles, GRIDS)
    m.discard_ahead()
