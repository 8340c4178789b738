"""GPU: the attention kernels by themselves, through the C ABI, against fp32 PyTorch attention on the same (bf16-rounded) inputs.

These are the kernels the timed bf16 path runs and the fp32-mode token tests do not: `attn_mfma_kernel<80>` (vision windows and
whole-image attention, surya/common/surya/encoder/__init__.py:238-261), `attn_mfma_kernel<128>` (decoder prefill, causal GQA over
the slot KV cache, decoder/__init__.py:101-128) and `decode_attn_flash_kernel<128, 5>` (split-K combine + bias + RoPE + KV append +
flash decode, decoder/__init__.py:193-234). The fp32 twins (`attn_valu_kernel`, `decode_attn_mfma_kernel`) run the same cases at
fp32 tolerance. Tolerance for bf16: 2e-2 * max|ref| (one bf16 rounding of P and of the output, fp32 accumulation)."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

from surya_amd import _lib as L

pytestmark = pytest.mark.gpu


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _i64(a):
    a = np.ascontiguousarray(a, dtype=np.int64)
    return a, a.ctypes.data_as(C.POINTER(C.c_int64))


def _i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(C.POINTER(C.c_int32))


def _ref_segment(q, k, v, scale, causal, group):
    """q [L, H, D], k / v [Lk, Hkv, D] (fp32) -> [L, H, D]; plain softmax attention, query i sees keys <= i when causal."""
    L_, H, D = q.shape
    kk = k.repeat_interleave(group, dim=1)
    vv = v.repeat_interleave(group, dim=1)
    s = torch.einsum("qhd,khd->hqk", q, kk) * scale
    if causal:
        m = torch.ones(L_, kk.shape[0], dtype=torch.bool, device=q.device).tril()
        s = s.masked_fill(~m, float("-inf"))
    p = torch.softmax(s, dim=-1)
    return torch.einsum("hqk,khd->qhd", p, vv)


def _run_segments(lib, dtype, D, heads, seg_lens, seed, spike=False):
    """Vision layout: one packed qkv buffer [P, 3 * heads * D]; segments are consecutive row ranges (windows / images)."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    P = sum(seg_lens)
    He = heads * D
    qkv = torch.randn(P, 3 * He, device="cuda", generator=g)
    if spike:                                   # one key far above the rest in every segment: the running-max rescale path
        a = 0
        for Ls in seg_lens:
            if Ls > 70:
                qkv[a + 65, He:2 * He] *= 6.0
            a += Ls
    qkv = qkv.to(dtype)
    out = torch.full((P, He), float("nan"), device="cuda", dtype=dtype)
    starts = np.cumsum([0] + list(seg_lens))[:-1]
    sl, slp = _i32(seg_lens)
    qo, qop = _i64(starts * 3 * He)
    ko, kop = _i64(starts * 3 * He + He)
    vo, vop = _i64(starts * 3 * He + 2 * He)
    oo, oop = _i64(starts * He)
    scale = 1.0 / math.sqrt(D)
    rc = lib.surya_op_attn(L.DTYPE_BF16 if dtype == torch.bfloat16 else L.DTYPE_F32, D, L.ptr(qkv), L.ptr(qkv), L.ptr(qkv), L.ptr(out),
                           slp, qop, kop, vop, oop, len(seg_lens), heads, 1, 0, C.c_float(scale), C.c_long(3 * He), C.c_long(D),
                           C.c_long(3 * He), C.c_long(D), C.c_long(He), C.c_long(D), _stream())
    assert rc == 0, rc
    torch.cuda.synchronize()
    f = qkv.float().view(P, 3, heads, D)
    ref = torch.empty(P, heads, D, device="cuda")
    a = 0
    for Ls in seg_lens:
        ref[a:a + Ls] = _ref_segment(f[a:a + Ls, 0], f[a:a + Ls, 1], f[a:a + Ls, 2], scale, False, 1)
        a += Ls
    return out.float().view(P, heads, D), ref


# windows of the bench geometries (<= 64 patches, ragged edge windows), whole images 228 (64x512 crop), 180, 784 (texify)
VISION_CASES = [
    ("windows", [64, 64, 64, 36, 16, 64, 24, 4], False),
    ("image_228", [228, 180, 228], False),
    ("image_784", [784, 100], False),
    ("spiked", [228, 130, 71], True),
]


@pytest.mark.parametrize("name,seg_lens,spike", VISION_CASES)
def test_attn_mfma_vision_d80_vs_fp32(hip_lib, name, seg_lens, spike):
    out, ref = _run_segments(hip_lib, torch.bfloat16, 80, 16, seg_lens, seed=len(seg_lens) * 17 + seg_lens[0], spike=spike)
    assert not torch.isnan(out).any()
    err = (out - ref).abs().max().item()
    assert err <= 2e-2 * max(1.0, ref.abs().max().item()), f"{name}: max err {err} vs max|ref| {ref.abs().max().item()}"


@pytest.mark.parametrize("D", [32, 64, 80, 128])
def test_attn_valu_fp32_vs_fp32(hip_lib, D):
    out, ref = _run_segments(hip_lib, torch.float32, D, 4, [64, 37, 228, 5], seed=D)
    err = (out - ref).abs().max().item()
    assert err <= 2e-5 * max(1.0, ref.abs().max().item()), f"max err {err}"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_attn_prefill_causal_gqa_d128_vs_fp32(hip_lib, dtype):
    """Decoder prefill layout: q rows [Ttot, (nq + 2 nkv) * d] packed per sequence, K / V read from the slot cache
    [slot][kv_head][Tmax][d]; causal; 10 query heads share 2 kv heads."""
    nq, nkv, d, Tmax, n_slots = 10, 2, 128, 256, 6
    lens = [63, 51, 202, 64, 130]
    slots = [4, 0, 5, 2, 1]
    g = torch.Generator(device="cuda").manual_seed(99)
    qkv_d = (nq + 2 * nkv) * d
    Ttot = sum(lens)
    q = torch.randn(Ttot, qkv_d, device="cuda", generator=g).to(dtype)
    kc = torch.randn(n_slots, nkv, Tmax, d, device="cuda", generator=g).to(dtype)
    vc = torch.randn(n_slots, nkv, Tmax, d, device="cuda", generator=g).to(dtype)
    out = torch.full((Ttot, nq * d), float("nan"), device="cuda", dtype=dtype)
    starts = np.cumsum([0] + lens)[:-1]
    sl, slp = _i32(lens)
    qo, qop = _i64(starts * qkv_d)
    ko, kop = _i64(np.array(slots) * nkv * Tmax * d)
    oo, oop = _i64(starts * nq * d)
    scale = 1.0 / math.sqrt(d)
    rc = hip_lib.surya_op_attn(L.DTYPE_BF16 if dtype == torch.bfloat16 else L.DTYPE_F32, d, L.ptr(q), L.ptr(kc), L.ptr(vc), L.ptr(out),
                               slp, qop, kop, kop, oop, len(lens), nq, nq // nkv, 1, C.c_float(scale), C.c_long(qkv_d), C.c_long(d),
                               C.c_long(d), C.c_long(Tmax * d), C.c_long(nq * d), C.c_long(d), _stream())
    assert rc == 0, rc
    torch.cuda.synchronize()
    worst, scale_ref = 0.0, 1.0
    a = 0
    for Ls, s in zip(lens, slots):
        qs = q[a:a + Ls, :nq * d].float().view(Ls, nq, d)
        ks = kc[s, :, :Ls].float().permute(1, 0, 2)
        vs = vc[s, :, :Ls].float().permute(1, 0, 2)
        ref = _ref_segment(qs, ks, vs, scale, True, nq // nkv)
        got = out[a:a + Ls].float().view(Ls, nq, d)
        worst = max(worst, (got - ref).abs().max().item())
        scale_ref = max(scale_ref, ref.abs().max().item())
        a += Ls
    assert not torch.isnan(out.float()).any()
    tol = 2e-2 if dtype == torch.bfloat16 else 2e-5
    assert worst <= tol * scale_ref, f"max err {worst} vs max|ref| {scale_ref}"


def _rope_table(Tmax, d, dtype, theta=10000.0):
    inv = 1.0 / (theta ** (torch.arange(0, d, 2, dtype=torch.float32) / d))
    ang = torch.arange(Tmax, dtype=torch.float32)[:, None] * inv[None, :]
    cs = torch.stack([ang.cos().to(dtype).float(), ang.sin().to(dtype).float()], dim=-1)      # [Tmax, d/2, 2], rounded like the model's
    return cs.contiguous()


def _decode_case(lib, dtype, d, nq, nkv, lens, S, Tmax, seed, ret_raw=False):
    G = nq // nkv
    M = len(lens)
    n_slots = M + 3
    g = torch.Generator(device="cuda").manual_seed(seed)
    qkv_d = (nq + 2 * nkv) * d
    slots = torch.randperm(n_slots, generator=torch.Generator().manual_seed(seed))[:M].to(torch.int32)
    part = torch.randn(S, M, qkv_d, device="cuda", generator=g) / math.sqrt(S)
    bias = (0.5 * torch.randn(qkv_d, device="cuda", generator=g)).to(dtype)
    kc = torch.randn(n_slots, nkv, Tmax, d, device="cuda", generator=g).to(dtype)
    vc = torch.randn(n_slots, nkv, Tmax, d, device="cuda", generator=g).to(dtype)
    kc0, vc0 = kc.clone(), vc.clone()
    cs = _rope_table(Tmax, d, dtype).cuda()
    out = torch.full((M, nq * d), float("nan"), device="cuda", dtype=dtype)
    act = slots.cuda()
    rl = torch.tensor(lens, dtype=torch.int32, device="cuda")
    scale = 1.0 / math.sqrt(d)
    rc = lib.surya_op_decode_attn(L.DTYPE_BF16 if dtype == torch.bfloat16 else L.DTYPE_F32, d, L.ptr(part), S, L.ptr(bias), L.ptr(out),
                                  L.ptr(kc), L.ptr(vc), L.ptr(act), L.ptr(rl), L.ptr(cs), M, nq, nkv, Tmax, C.c_float(scale), _stream())
    assert rc == 0, rc
    torch.cuda.synchronize()
    if ret_raw:
        return out, kc, vc
    # fp32 PyTorch restatement on the same rounded inputs
    x = (part.sum(0) + bias.float()).to(dtype).float()                        # projection output rounded to the storage dtype
    qh = x[:, :nq * d].view(M, nq, d)
    kh = x[:, nq * d:(nq + nkv) * d].view(M, nkv, d)
    vh = x[:, (nq + nkv) * d:].view(M, nkv, d)
    half = d // 2
    worst, worst_kv, ref_max = 0.0, 0.0, 1.0
    for r in range(M):
        ln, s = lens[r], int(slots[r])
        c, sn = cs[ln, :, 0], cs[ln, :, 1]

        def rope(t):
            t1, t2 = t[..., :half], t[..., half:]
            return torch.cat([(t1 * c - t2 * sn), (t2 * c + t1 * sn)], dim=-1).to(dtype).float()

        qr = (rope(qh[r]) * scale).to(dtype).float()                          # the kernel stores q * scale in the storage dtype
        kr = rope(kh[r])
        K = torch.cat([kc0[s, :, :ln].float(), kr[:, None, :]], dim=1)         # [nkv, ln + 1, d]
        V = torch.cat([vc0[s, :, :ln].float(), vh[r][:, None, :]], dim=1)
        sc = torch.einsum("hd,hkd->hk", qr, K.repeat_interleave(G, dim=0))
        p = torch.softmax(sc, dim=-1)
        ref = torch.einsum("hk,hkd->hd", p, V.repeat_interleave(G, dim=0))
        got = out[r].float().view(nq, d)
        worst = max(worst, (got - ref).abs().max().item())
        ref_max = max(ref_max, ref.abs().max().item())
        worst_kv = max(worst_kv, (kc[s, :, ln].float() - kr).abs().max().item(), (vc[s, :, ln].float() - vh[r]).abs().max().item())
        # rows other than the appended one are untouched
        assert torch.equal(kc[s, :, :ln], kc0[s, :, :ln]) and torch.equal(vc[s, :, ln + 1:], vc0[s, :, ln + 1:])
    assert not torch.isnan(out.float()).any()
    return worst, ref_max, worst_kv


# KV lengths of the bench (prompt 51..63, +47 steps), the 128-key tile edge, the texify horizon (202 -> 970)
DECODE_LENS = [0, 1, 15, 16, 51, 63, 64, 110, 127, 128, 129, 255, 256, 300, 460, 700, 969]


@pytest.fixture(params=[4, 3], ids=["flash2", "flash"])
def dattn(request, hip_lib):
    """Both bf16 decode-attention kernels behind surya_op_decode_attn: 4 = decode_attn_flash2_kernel (round 4, the default), 3 = the
    third version it replaces (kept as the A/B arm of tools/microbench/decode_sweep.py)."""
    L.check(hip_lib.surya_set_tuning(b"dattn", C.c_int(request.param)), "surya_set_tuning")
    yield request.param
    L.check(hip_lib.surya_set_tuning(b"dattn", C.c_int(4)), "surya_set_tuning")


@pytest.mark.parametrize("S", [1, 2, 3, 4, 5, 8])
def test_decode_attn_flash_d128_g5_vs_fp32(hip_lib, dattn, S):
    worst, ref_max, worst_kv = _decode_case(hip_lib, torch.bfloat16, 128, 10, 2, DECODE_LENS, S, 1024, seed=S)
    assert worst <= 2e-2 * ref_max, f"max err {worst} vs max|ref| {ref_max}"
    assert worst_kv <= 4e-2, f"appended K/V rows differ by {worst_kv}"      # at most one bf16 ulp of |k| < 8 (fp32 contraction order)


@pytest.mark.parametrize("d,nq,nkv", [(128, 16, 2), (64, 8, 2), (32, 4, 2)])
def test_decode_attn_flash_other_shapes_vs_fp32(hip_lib, dattn, d, nq, nkv):
    worst, ref_max, worst_kv = _decode_case(hip_lib, torch.bfloat16, d, nq, nkv, [0, 7, 64, 127, 128, 130, 257], 2, 512, seed=d)
    assert worst <= 2e-2 * ref_max, f"max err {worst} vs max|ref| {ref_max}"
    assert worst_kv <= 4e-2


def test_decode_attn_flash2_agrees_with_flash(hip_lib):
    """Same slab order, rounding points, MFMA order and combine: the round-4 kernel's outputs and appended K / V rows equal the third
    version's to one bf16 ulp (the rotation's products are pinned with explicit fma / mul in the new kernel only)."""
    outs = {}
    for ver in (3, 4):
        L.check(hip_lib.surya_set_tuning(b"dattn", C.c_int(ver)), "surya_set_tuning")
        outs[ver] = _decode_case(hip_lib, torch.bfloat16, 128, 10, 2, DECODE_LENS, 3, 1024, seed=5, ret_raw=True)
    L.check(hip_lib.surya_set_tuning(b"dattn", C.c_int(4)), "surya_set_tuning")
    (o3, k3, v3), (o4, k4, v4) = outs[3], outs[4]
    assert torch.equal(v3, v4)                                   # V rows are not rotated: bit-identical
    assert (k3.float() - k4.float()).abs().max().item() <= 4e-2  # one bf16 ulp of |k| < 8
    assert (o3.float() - o4.float()).abs().max().item() <= 2e-2 * max(1.0, o3.float().abs().max().item())


@pytest.mark.parametrize("d,nq,nkv,lens,Tmax", [(128, 10, 2, DECODE_LENS, 1024), (64, 8, 2, [0, 7, 64, 127, 128, 130, 257, 383, 384, 500], 512),
                                                (32, 4, 2, [0, 127, 128, 129, 255, 256, 257, 511], 520)])
def test_decode_attn_two_tile_buffers_bit_identical_and_vs_fp32(hip_lib, d, nq, nkv, lens, Tmax):
    """decode_attn_flash2_kernel<.., DB = true> (dattn_db = 1: next tile fetched while this one is computed; the variant the model
    picks once an active context exceeds one 128-key tile) against fp32 PyTorch, and bit for bit against the single-buffered kernel:
    outputs AND the appended cache rows, over contexts of 1 ... 8 tiles incl. the tile edges and the new row opening a fresh tile."""
    outs = {}
    try:
        for db in (-1, 1):
            L.check(hip_lib.surya_set_tuning(b"dattn_db", C.c_int(db)), "surya_set_tuning")
            outs[db] = _decode_case(hip_lib, torch.bfloat16, d, nq, nkv, lens, 3, Tmax, seed=5 + d, ret_raw=True)
        worst, ref_max, worst_kv = _decode_case(hip_lib, torch.bfloat16, d, nq, nkv, lens, 2, Tmax, seed=d)
    finally:
        L.check(hip_lib.surya_set_tuning(b"dattn_db", C.c_int(0)), "surya_set_tuning")
    for a, b in zip(outs[-1], outs[1]):
        assert torch.equal(a, b)
    assert worst <= 2e-2 * ref_max, f"max err {worst} vs max|ref| {ref_max}"
    assert worst_kv <= 4e-2


def test_decode_attn_fp32_mode_vs_fp32(hip_lib):
    worst, ref_max, worst_kv = _decode_case(hip_lib, torch.float32, 128, 10, 2, DECODE_LENS, 3, 1024, seed=11)
    assert worst <= 2e-5 * ref_max and worst_kv <= 1e-5, (worst, worst_kv)
