"""GPU: the detector's fused forms (csrc/det_fused.h, sa::Tuning det_fuse) against the op list they replace, on the SAME engine and
weights, and against the fp32 oracle.

det_fuse = 0 runs detection/plan.py's op list as written (round 5's path: every intermediate through HBM); each bit folds one pair:
  1  LiteMLA depthwise 5x5 + grouped 1x1      2  LiteMLA kv + out (fp32 MFMA)      4  z0 inside the head's sum + classify pass
  8  MBConv depthwise 3x3 + projection        16 FusedMBConv 3x3 + Hardswish + projection (the two Cout = 64 blocks of stage 0)
  32 the three 32-channel stem convolutions on the patch-in-LDS kernel (a kernel choice, not a fusion)
  64 whole MBConv blocks (expand 1x1 + depthwise 3x3 + projection 1x1; csrc/det_mbconv.h) -- the two stride-2 transitions
  256 (with 32) the first convolution reads the caller's pixels itself: no input-layout launch (same conversions: bit-identical)
  512 the stem's residual block (two 32 -> 32 3x3 convolutions) in one launch, the tensor between them in LDS only (bit-identical)
  128 (with 4) the folded head entirely on the matrix cores: bilinear up-samplings as a constant K = 96 map on z0's accumulators (csrc/det_head.h)
Expectations written into the asserts:
  * bits 4, 8, 16, 32, 64, 256 and 512 repeat the op list's arithmetic exactly (same MFMA, same K order, same rounding points): heat maps bit-identical;
  * bit 1 runs the grouped 1x1 on the bf16 MFMA instead of an fp32 fma chain, bit 2 sums tokens on the fp32 MFMA in another order and bit 128
    interpolates on the MFMA (and does not round z0 to bf16 on its own):
    fp32-accumulation re-association only, but a bf16 rounding step of an intermediate may flip and the flips travel through the six
    LiteMLA blocks and the head -- measured 1.1e-2 max / 1e-3 mean on the [0, 1] maps at 1024^2 (the bf16 tolerance against the fp32 oracle is
    3e-2 / 4e-3): <= 2e-2 max and <= 2e-3 mean here, and no further from the fp32 oracle than the op list is (test_fused_forms_vs_oracle);
"""
import ctypes as C

import pytest
import torch

from oracle import det_oracle as do
from surya_amd.config import det_config
from surya_amd.synth import make_det_weights, make_pages

pytestmark = pytest.mark.gpu

ALL = 1023


def _set(lib, v):
    from surya_amd import _lib as L
    L.check(lib.surya_set_tuning(b"det_fuse", C.c_int(v)), "surya_set_tuning")


def _default(lib):
    _set(lib, DEFAULT)


DEFAULT = 1023


def build(name, size, dtype, max_batch):
    from surya_amd.detection.model import HipDetModel
    cfg = det_config(name)
    sd = make_det_weights(cfg, 0)
    return cfg, sd, HipDetModel(cfg, sd, height=size, width=size, dtype=dtype, max_batch=max_batch)


@pytest.mark.parametrize("pages_n,size", [(8, 1024), (3, 672), (1, 256)])
def test_fused_forms_vs_op_list_bf16(hip_lib, pages_n, size):
    """DET-DEFAULT bf16 at the BASELINE page size (8 pages: a whole-MBConv variant on v_dot2c_f32_bf16 repeated the op list's bits on pages 0-6 of this
    input and left them on page 7 -- few pages are not enough to call a form bit-identical), at a ragged size (672 = 21 x 32: every stage has partial tiles) and at the size the
    oracle tests use. Each bit alone and all together against det_fuse = 0."""
    cfg, sd, m = build("DET-DEFAULT", size, torch.bfloat16, pages_n)
    x = do.normalise_pages(list(make_pages(pages_n, size, seed=99))).cuda().contiguous()
    try:
        _set(hip_lib, 0)
        base = m.forward(x).clone()
        again = m.forward(x).clone()
        assert torch.equal(base, again)
        assert torch.isfinite(base).all() and base.std().item() > 0.02
        for bit in (1, 2, 4, 8, 16, 32, 64, 72, 127, 132, 288, 512, 800, ALL):
            _set(hip_lib, bit)
            h = m.forward(x).clone()
            h2 = m.forward(x).clone()
            assert torch.equal(h, h2), f"det_fuse={bit}: not run-to-run identical"
            d = (h - base).abs().max().item()
            print(f"{size}^2 x {pages_n}: det_fuse={bit:2d} vs op list: max abs diff {d:.3e}, identical {torch.equal(h, base)}")
            assert torch.isfinite(h).all()
            if bit in (4, 8, 16, 32, 64, 72, 288, 512, 800):
                assert torch.equal(h.view(torch.int32), base.view(torch.int32)), f"det_fuse={bit} must repeat the op list's bits"
            else:
                assert d <= 2e-2 and (h - base).abs().mean().item() <= 2e-3, (bit, d)
    finally:
        _default(hip_lib)


def test_fused_forms_vs_oracle(hip_lib):
    """All fused forms on (the default) vs the fp32 oracle: inside the bf16 tolerance of tests/test_gpu_det.py and no further away than
    the op list."""
    cfg, sd, m = build("DET-DEFAULT", 256, torch.bfloat16, 2)
    pages = make_pages(2, 256, seed=5)
    x = do.normalise_pages(pages)
    ref = do.heatmaps(sd, cfg, x)
    try:
        _set(hip_lib, 0)
        e0 = (m.forward(x.cuda()).cpu() - ref).abs()
        _set(hip_lib, ALL)
        e1 = (m.forward(x.cuda()).cpu() - ref).abs()
    finally:
        _default(hip_lib)
    print(f"vs oracle: op list max {e0.max():.4f} mean {e0.mean():.5f} | fused max {e1.max():.4f} mean {e1.mean():.5f}")
    assert e1.max().item() <= max(3e-2, 1.5 * e0.max().item())
    assert e1.mean().item() <= max(4e-3, 1.5 * e0.mean().item())


@pytest.mark.parametrize("size", [256, 352])
def test_litemla_one_launch_fp32_vs_two_launch(hip_lib, size):
    """Reference mode: LiteMLA's kv + out in one launch on the fp32 MFMA (the only fused form fp32 takes) against the two-launch
    kernels: fp32 sums in another order, <= 2e-5 on the maps; both <= 1e-4 from the oracle. 352 = 11 x 32: 121 tokens, a ragged chunk."""
    cfg, sd, m = build("DET-DEFAULT", size, torch.float32, 1)
    x = do.normalise_pages(make_pages(1, size, seed=3))
    ref = do.heatmaps(sd, cfg, x)
    try:
        _set(hip_lib, 0)
        a = m.forward(x.cuda()).cpu()
        _set(hip_lib, 2)
        b = m.forward(x.cuda()).cpu()
    finally:
        _default(hip_lib)
    d = (a - b).abs().max().item()
    print(f"fp32 {size}: one launch vs two: {d:.2e}; vs oracle {(a - ref).abs().max():.2e} / {(b - ref).abs().max():.2e}")
    assert d <= 2e-5
    assert (b - ref).abs().max().item() <= 1e-4


def test_forward_timed_reports_every_op(hip_lib):
    cfg, sd, m = build("DET-TINY", 128, torch.bfloat16, 2)
    x = do.normalise_pages(make_pages(2, 128, seed=1)).cuda().contiguous()
    ref = m.forward(x).clone()
    heat, rows = m.forward_timed(x)
    assert torch.equal(heat, ref)
    assert len(rows) == len(m.plan_ops) and all(ms >= 0 for _, ms in rows) and sum(ms for _, ms in rows) > 0


@pytest.mark.parametrize("h,w,pages_n", [(384, 672, 2), (736, 288, 3)])
def test_fused_forms_non_square_pages(hip_lib, h, w, pages_n):
    """Non-square pages (tile rasters with tiles_x != tiles_y; 672 / 4 = 168 and 288 / 4 = 72 are odd multiples of 8: the head kernel's half-outside
    last tile): the bit-identical forms repeat the op list's bits, the re-associating ones stay inside their bound."""
    from surya_amd.detection.model import HipDetModel
    cfg = det_config("DET-DEFAULT")
    sd = make_det_weights(cfg, 0)
    m = HipDetModel(cfg, sd, height=h, width=w, dtype=torch.bfloat16, max_batch=pages_n)
    g = torch.Generator().manual_seed(7)
    x = (torch.randn(pages_n, 3, h, w, generator=g) * 0.8).cuda().contiguous()
    try:
        _set(hip_lib, 0)
        base = m.forward(x).clone()
        assert torch.isfinite(base).all() and base.std().item() > 0.01
        for bit in (8, 16, 32, 64, 72, 288, 512, 800, 132, ALL):
            _set(hip_lib, bit)
            hm = m.forward(x).clone()
            d = (hm - base).abs()
            print(f"{h}x{w} x {pages_n}: det_fuse={bit:3d} vs op list: max {d.max().item():.3e} mean {d.mean().item():.3e}")
            if bit in (8, 16, 32, 64, 72, 288, 512, 800):
                assert torch.equal(hm.view(torch.int32), base.view(torch.int32)), f"det_fuse={bit} must repeat the op list's bits"
            else:
                assert d.max().item() <= 2e-2 and d.mean().item() <= 2e-3, (bit, d.max().item())
    finally:
        _default(hip_lib)


@pytest.mark.parametrize("pix", [3, 4])
def test_input_fused_stem_u8_pages(hip_lib, pix):
    """surya_det_forward_u8 with the first convolution reading the uint8 pages itself (bits 32 + 256: rescale + normalise in the patch loader, RGB
    and RGBX) against the op list (input-layout launch + implicit-GEMM convolution): the same conversions, bit-identical maps; 672 x 416 page."""
    import numpy as np
    from surya_amd.detection.model import HipDetModel
    cfg = det_config("DET-DEFAULT")
    sd = make_det_weights(cfg, 0)
    h, w, n = 416, 672, 3
    m = HipDetModel(cfg, sd, height=h, width=w, dtype=torch.bfloat16, max_batch=n)
    rng = np.random.default_rng(11)
    u8 = torch.from_numpy(rng.integers(0, 256, size=(n, h, w, pix), dtype=np.uint8)).cuda().contiguous()
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    try:
        _set(hip_lib, 0)
        base = m.forward_u8(u8, mean, std).clone()
        for bit in (32, 288, 800, ALL):
            _set(hip_lib, bit)
            hm = m.forward_u8(u8, mean, std).clone()
            d = (hm - base).abs().max().item()
            print(f"u8 pix={pix}: det_fuse={bit} vs op list: max abs diff {d:.3e}")
            if bit != ALL:
                assert torch.equal(hm.view(torch.int32), base.view(torch.int32)), bit
            else:
                assert d <= 2e-2
    finally:
        _default(hip_lib)


@pytest.mark.parametrize("h,w,pages_n", [(1024, 1024, 2), (416, 672, 3)])
def test_x4_output_upsample_repeats_the_generic_kernel(hip_lib, h, w, pages_n):
    """sa::Tuning det_up4: the x4 output up-sampling with a 4 x 4 output block per thread (upsample_planes_x4_kernel) against the generic float4 kernel
    (det_up4 = 0): same coefficients, same operands, same expression -- bit-identical heat maps, also on a page whose quarter-resolution map is not a
    multiple of the 64 x 4 thread block."""
    from surya_amd import _lib as L
    from surya_amd.detection.model import HipDetModel
    cfg = det_config("DET-DEFAULT")
    sd = make_det_weights(cfg, 0)
    m = HipDetModel(cfg, sd, height=h, width=w, dtype=torch.bfloat16, max_batch=pages_n)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(pages_n, 3, h, w, generator=g).cuda().contiguous()
    try:
        L.check(hip_lib.surya_set_tuning(b"det_up4", C.c_int(0)), "surya_set_tuning")
        base = m.forward(x).clone()
        L.check(hip_lib.surya_set_tuning(b"det_up4", C.c_int(1)), "surya_set_tuning")
        got = m.forward(x).clone()
        assert base.std().item() > 0.01
        assert torch.equal(got.view(torch.int32), base.view(torch.int32))
    finally:
        L.check(hip_lib.surya_set_tuning(b"det_up4", C.c_int(1)), "surya_set_tuning")
