"""CPU: the host-side MXFP8 weight quantiser (surya_amd/mx.py, torch casts) against the independent restatement of the OCP
MX format in oracle/mx_oracle.py (integer arithmetic), and the layout of the MXFP8 weight table handed to
surya_rec_set_mx_weights. No reference counterpart exists (surya has no fp8 mode); the format is the published one."""
import numpy as np
import torch

from oracle import mx_oracle as mo
from surya_amd import _lib as L
from surya_amd import mx
from surya_amd.config import rec_config
from surya_amd.recognition.weights import repack_rec_weights
from surya_amd.synth import make_rec_weights


def _cases():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal((96, 256)) * np.exp(rng.standard_normal((96, 1)) * 4)).astype(np.float32)
    x[3, :32] = 0                                   # an all-zero block
    x[4, 32:64] = 448.0                             # exactly the format maximum
    x[5, 64:96] = np.float32(449.0)                 # just above: needs the next scale
    x[6, 96:128] = np.float32(1e-30)                # tiny values
    x[7, :32] = np.ldexp(np.float32(1.0625), np.arange(32) - 20).astype(np.float32)    # ties and a wide in-block range
    x[8, :32] = -x[7, :32]
    return x


def test_all_e4m3_codes_round_trip():
    codes = np.arange(256, dtype=np.uint8)
    v = mo.e4m3_decode(codes)
    ok = ~np.isnan(v)
    assert ok.sum() == 254 and np.nanmax(v) == 448.0
    assert np.array_equal(mo.e4m3_encode(v[ok]), codes[ok])
    assert np.array_equal(v[ok], torch.from_numpy(codes).view(torch.float8_e4m3fn).float().numpy()[ok])


def test_host_quantiser_matches_oracle_bit_for_bit():
    x = _cases()
    q, s = mo.quantize(x)
    q2, s2 = mx.quantize_mx(torch.from_numpy(x))
    assert np.array_equal(q, q2.numpy()) and np.array_equal(s, s2.numpy())
    d = mo.dequantize(q, s)
    assert np.array_equal(d, mx.dequantize_mx(q2, s2).numpy())
    # the scale rule: nothing saturates, and the block maximum lands in (224, 448] (one binade below the format maximum)
    amax = np.abs(x.reshape(96, 8, 32)).max(-1)
    scaled = amax / np.ldexp(1.0, s.astype(np.int64) - 127)
    nz = amax > 1e-20
    assert (scaled[nz] <= 448).all() and (scaled[nz] > 224).all()
    assert (s[~(amax > 0)] == 0).all()
    # element error: half a step of the 3-bit mantissa relative to the element's own binade, or half a subnormal step
    err = np.abs(d - x).reshape(96, 8, 32)
    step = np.maximum(np.abs(x).reshape(96, 8, 32) * 2.0 ** -4, np.ldexp(1.0, s.astype(np.int64) - 127 - 10)[..., None])
    assert (err <= step * 1.0001).all()


def test_mx_weight_table_layout():
    cfg = rec_config("REC-SMALL")
    sd = make_rec_weights(cfg, 0)
    w = repack_rec_weights(cfg, sd, torch.bfloat16, "cpu")
    t = mx.repack_rec_mx_weights(cfg, w, "cpu")
    d = cfg.decoder
    assert len(t) == d.num_hidden_layers * 8 + 2
    qkv = (d.num_attention_heads + 2 * d.num_key_value_heads) * d.head_dim
    shapes = [(qkv, d.hidden_size), (d.hidden_size, d.num_attention_heads * d.head_dim), (2 * d.intermediate_size, d.hidden_size),
              (d.hidden_size, d.intermediate_size)]
    for l in range(d.num_hidden_layers):
        for k, (n, kk) in enumerate(shapes):
            q, s = t[l * 8 + 2 * k], t[l * 8 + 2 * k + 1]
            assert q.dtype == torch.uint8 and tuple(q.shape) == (n, kk) and tuple(s.shape) == (kk // 128, n, 4)   # K-tile-major scales
    assert tuple(t[-2].shape) == (d.vocab_size, d.hidden_size)
    # a weight row survives the round trip to within the format's precision, from the SAME (bf16, kernel-layout) source
    base = L.RW_GLOBALS + cfg.encoder.depth * L.RE_COUNT
    src = w[base + L.RD_GU_W].float()
    s_rows = t[5].permute(1, 0, 2).reshape(src.shape[0], -1)
    assert torch.equal(mx.tile_major_scales(s_rows), t[5])
    back = mx.dequantize_mx(t[4], s_rows)
    assert (back - src).abs().max() <= src.abs().max() * 2.0 ** -4


def test_kv8_row_format_properties():
    """The fp8 KV cache row format (oracle/mx_oracle.py::kv8_quantize; csrc/decode_attn_kv8.h): power-of-two scale, nothing
    saturates, error <= 1/16 of the row's absmax per element, dequantised rows are fixed points in value and exact in bf16."""
    import torch
    from oracle import mx_oracle as mo
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((7, 3, 128)) * np.exp2(rng.integers(-8, 9, (7, 3, 1)))).astype(np.float32)
    x[0, 0] = 0.0                                            # an all-zero row
    x[1, 1, 5] = 448.0 * 4                                   # a row whose absmax sits exactly on the format maximum after scaling
    q, s = mo.kv8_quantize(x)
    assert q.dtype == np.uint8 and s.shape == x.shape[:-1]
    assert (np.log2(s[s > 0]) % 1 == 0).all()
    d = mo.kv8_dequantize(q, s)
    assert np.isfinite(d).all() and (d[0, 0] == 0).all()
    amax = np.abs(x).max(-1, keepdims=True)
    assert (np.abs(d - x) <= amax / 16 + 1e-30).all()
    assert np.array_equal(mo.kv8_dequantize(*mo.kv8_quantize(d)), d)
    assert np.array_equal(torch.from_numpy(d).bfloat16().float().numpy(), d)
    # torch's own e4m3 cast agrees with the table-based encoder on scaled rows
    scaled = torch.from_numpy(x / s[..., None].clip(min=1e-38))
    assert np.array_equal(scaled.to(torch.float8_e4m3fn).view(torch.uint8).numpy() & 0x7F, q & 0x7F)
