"""CPU restatement of the MXFP8 arithmetic of the fp8 decode path (TEST INFRASTRUCTURE ONLY, like everything under oracle/).

There is no reference implementation to pin against: VikParuchuri/surya has no fp8 mode (BASELINE.json configs[4] asks
for one as an MI355X design choice). What is restated here is the published format -- OCP Microscaling Formats (MX) v1.0:
e4m3 elements (bias 7, max 448, subnormals, no infinities), one shared E8M0 scale 2^(byte - 127) per 32 consecutive
elements -- plus this repository's own scale rule (csrc/common.h mx_block_exp: smallest power of two with absmax / scale
<= 448; round to nearest even). Written independently of surya_amd/mx.py (integer / bit arithmetic in numpy instead of
torch casts) so that the two check each other in tests/test_mx_cpu.py.
"""
from __future__ import annotations

import numpy as np
import torch

BLOCK = 32


def e4m3_decode(b: np.ndarray) -> np.ndarray:
    """uint8 -> float32 value of OCP e4m3fn (0x7f / 0xff = NaN)."""
    b = b.astype(np.int32)
    s, e, m = b >> 7, (b >> 3) & 15, b & 7
    v = np.where(e == 0, np.ldexp(m / 8.0, -6), np.ldexp(1.0 + m / 8.0, e - 7))
    v = np.where((e == 15) & (m == 7), np.nan, v)
    return np.where(s == 1, -v, v).astype(np.float32)


_TABLE = None


def e4m3_encode(x: np.ndarray) -> np.ndarray:
    """float -> uint8, round to nearest even on the e4m3 grid, |x| <= 448 required (the block scale guarantees it)."""
    global _TABLE
    if _TABLE is None:
        _TABLE = e4m3_decode(np.arange(0, 127, dtype=np.uint8)).astype(np.float64)     # non-negative finite values, ascending
    a = np.abs(x.astype(np.float64))
    assert not (a > 448.0).any(), "e4m3_encode: value above the format maximum"
    hi = np.searchsorted(_TABLE, a, side="left").clip(1, 126)
    lo = hi - 1
    dl, dh = a - _TABLE[lo], _TABLE[hi] - a
    pick_hi = (dh < dl) | ((dh == dl) & (hi % 2 == 0))                                  # ties -> even mantissa = even code
    code = np.where(pick_hi, hi, lo)
    code = np.where(a == 0, 0, code)
    return (code | (np.signbit(x).astype(np.int64) << 7)).astype(np.uint8)


def block_exponent(amax: np.ndarray) -> np.ndarray:
    """Smallest e with amax / 2^e <= 448; all-zero blocks get -127."""
    amax = amax.astype(np.float64)
    with np.errstate(divide="ignore"):
        e = np.ceil(np.log2(np.where(amax > 0, amax, 1.0) / 448.0)).astype(np.int64)
    # log2 rounding guard: enforce the definition exactly
    e = np.where(np.ldexp(amax, -e) > 448.0, e + 1, e)
    e = np.where((e > -127) & (np.ldexp(amax, -(e - 1)) <= 448.0), e - 1, e)
    return np.where(amax > 0, e, -127).clip(-127, 127)


def quantize(x: np.ndarray):
    """[..., K] -> (uint8 codes [..., K], uint8 E8M0 scales [..., K / 32])."""
    K = x.shape[-1]
    assert K % BLOCK == 0
    b = x.astype(np.float32).reshape(x.shape[:-1] + (K // BLOCK, BLOCK))
    e = block_exponent(np.abs(b).max(-1))
    q = e4m3_encode(np.ldexp(b.astype(np.float64), -e[..., None]))
    return q.reshape(x.shape), (e + 127).astype(np.uint8)


def dequantize(q: np.ndarray, s: np.ndarray) -> np.ndarray:
    K = q.shape[-1]
    v = e4m3_decode(q).reshape(q.shape[:-1] + (K // BLOCK, BLOCK)).astype(np.float64)
    return np.ldexp(v, s.astype(np.int64)[..., None] - 127).reshape(q.shape).astype(np.float32)


def fake_quant(x: torch.Tensor) -> torch.Tensor:
    """Quantise-dequantise along the last dimension (what an MXFP8 GEMM operand carries)."""
    q, s = quantize(x.detach().float().cpu().numpy())
    return torch.from_numpy(dequantize(q, s)).to(x.dtype)


# ---- "KV8": the fp8 KV cache of the decode steps (csrc/decode_attn_kv8.h, include/surya_amd.h surya_rec_set_kv_fp8) -----------------
# One MX-style block per (token, kv head) that spans the whole head dim: a power-of-two scale (block_exponent above, stored as the
# fp32 value 2^e) and round-to-nearest-even e4m3 elements. No reference counterpart (see the header comment): this IS the definition.

def kv8_quantize(x: np.ndarray):
    """[..., D] rows -> (uint8 codes [..., D], float32 scales [...])."""
    x = x.astype(np.float32)
    e = block_exponent(np.abs(x).max(-1))
    q = e4m3_encode(np.ldexp(x.astype(np.float64), -e[..., None]))
    return q, np.ldexp(np.float32(1.0), e).astype(np.float32)


def kv8_dequantize(q: np.ndarray, scale: np.ndarray) -> np.ndarray:
    return (e4m3_decode(q).astype(np.float64) * scale.astype(np.float64)[..., None]).astype(np.float32)
