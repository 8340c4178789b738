"""GPU: detection forward pass through the C ABI vs the CPU oracle (bit-identical to the reference) and the golden
fixture recorded from the reference module.

Tolerances: fp32 reference mode <= 1e-4 abs on the [0,1] maps (re-ordered fp32 sums through ~60 conv
layers; the reference itself is only reproducible to that level across conv algorithms); bf16 <= 3e-2 abs, mean <= 4e-3.
"""
import os

import pytest
import torch

from oracle import det_oracle as do
from surya_amd.config import det_config
from surya_amd.synth import make_det_weights, make_pages

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def build(name, size, dtype, max_batch=4):
    from surya_amd.detection.model import HipDetModel
    cfg = det_config(name)
    sd = make_det_weights(cfg, 0)
    return cfg, sd, HipDetModel(cfg, sd, height=size, width=size, dtype=dtype, max_batch=max_batch)


@pytest.mark.parametrize("name,size,n", [("DET-TINY", 128, 2), ("DET-TINY", 256, 3), ("DET-DEFAULT", 256, 1)])
def test_det_fp32_vs_oracle(hip_lib, name, size, n):
    cfg, sd, m = build(name, size, torch.float32)
    x = do.normalise_pages(make_pages(n, size, seed=1234))
    ref_low = do.forward(sd, cfg, x)
    ref_up = do.heatmaps(sd, cfg, x)
    heat, low = m.forward(x.cuda(), want_lowres=True)
    e_low = (low.cpu() - ref_low).abs().max().item()
    e_up = (heat.cpu() - ref_up).abs().max().item()
    assert e_low <= 1e-4 and e_up <= 1e-4, (e_low, e_up)
    assert ref_low.std().item() > 0.05                  # the synthetic init gives a non-trivial map


def test_det_fp32_vs_reference_fixture(hip_lib):
    g = torch.load(os.path.join(GOLD, "det_tiny.pt"))
    cfg, sd, m = build(g["config"], g["size"], torch.float32)
    x = do.normalise_pages(make_pages(g["n"], g["size"], seed=g["page_seed"]))
    heat, low = m.forward(x.cuda(), want_lowres=True)
    assert (low.cpu() - g["logits"]).abs().max().item() <= 1e-4
    assert (heat.cpu()[:, :, ::8, ::8] - g["upsampled_sample"]).abs().max().item() <= 1e-4


@pytest.mark.parametrize("name,size,n", [("DET-TINY", 128, 2), ("DET-DEFAULT", 256, 1)])
def test_det_bf16_vs_oracle(hip_lib, name, size, n):
    cfg, sd, m = build(name, size, torch.bfloat16)
    x = do.normalise_pages(make_pages(n, size, seed=1234))
    ref = do.heatmaps(sd, cfg, x)
    # the reference's own rounding model: same oracle with bf16 weights / activations
    ref_b16 = do.heatmaps({k: v.bfloat16() for k, v in sd.items()}, cfg, x.bfloat16()).float()
    heat = m.forward(x.cuda()).cpu()
    err, ref_err = (heat - ref).abs(), (ref_b16 - ref).abs()
    print(f"bf16 det {name}: max {err.max():.4f} mean {err.mean():.5f} | torch-bf16 path max {ref_err.max():.4f} mean {ref_err.mean():.5f}")
    assert err.max().item() <= max(3e-2, 2 * ref_err.max().item())
    assert err.mean().item() <= max(4e-3, 2 * ref_err.mean().item())


@pytest.mark.parametrize("name,size,n,dtype", [("DET-TINY", 256, 2, torch.float32), ("DET-DEFAULT", 256, 1, torch.float32),
                                               ("DET-DEFAULT", 256, 1, torch.bfloat16), ("DET-TINY", 160, 2, torch.bfloat16)])
def test_det_folded_head_vs_reference_order(hip_lib, name, size, n, dtype):
    """The decode head as sum_s up(A_s x_s) + c (detection/plan.py; SA_DET_UPSUM_CLASSIFY) against the SAME engine running the
    reference's op order (linear_c -> upsample -> concat -> linear_fuse, DETECTOR_HEAD_UNFOLDED=1) and against the oracle: the two
    forms agree to fp32 re-association in reference mode, and in bf16 the folded form is no further from the fp32 oracle than the
    unfolded one (it has fewer rounding points)."""
    from surya_amd.settings import settings
    x = do.normalise_pages(make_pages(n, size, seed=77))
    ref = do.heatmaps(sd := make_det_weights(det_config(name), 0), det_config(name), x)
    maps = {}
    for unfolded in (True, False):
        settings.DETECTOR_HEAD_UNFOLDED = unfolded
        try:
            cfg, _, m = build(name, size, dtype)
        finally:
            settings.DETECTOR_HEAD_UNFOLDED = False
        from surya_amd.detection.plan import OP_UPCAT, OP_UPSUM_CLASSIFY, build_det_plan
        types = [o["type"] for o in build_det_plan(cfg, sd, size, size, folded_head=not unfolded).ops]
        assert (OP_UPCAT in types) == unfolded and (OP_UPSUM_CLASSIFY in types) == (not unfolded)
        heat, low = m.forward(x.cuda(), want_lowres=True)
        maps[unfolded] = (heat.cpu(), low.cpu())
    d_up = (maps[True][0] - maps[False][0]).abs().max().item()
    d_low = (maps[True][1] - maps[False][1]).abs().max().item()
    e_f = (maps[False][0] - ref).abs().max().item()
    e_u = (maps[True][0] - ref).abs().max().item()
    print(f"{name} {dtype}: folded vs unfolded {d_up:.2e} / {d_low:.2e}; vs oracle folded {e_f:.2e} unfolded {e_u:.2e}")
    if dtype == torch.float32:
        assert d_up <= 2e-5 and d_low <= 2e-5 and e_f <= 1e-4
    else:
        assert e_f <= max(3e-2, 1.5 * e_u)
    # the per-pixel form of the sum + classify pass (det_head_blk = 0; also what odd ratios take) against the register-blocked one
    import ctypes as C
    from surya_amd import _lib as L
    L.check(L.lib().surya_set_tuning(b"det_head_blk", C.c_int(0)), "surya_set_tuning")
    try:
        heat_px, low_px = m.forward(x.cuda(), want_lowres=True)
    finally:
        L.check(L.lib().surya_set_tuning(b"det_head_blk", C.c_int(1)), "surya_set_tuning")
    d_px = (low_px.cpu() - maps[False][1]).abs().max().item()
    print(f"   per-pixel vs blocked kernel: {d_px:.2e}")
    assert d_px <= (2e-6 if dtype == torch.float32 else 8e-3)      # bf16: one rounding of y = T(relu(v)) may flip at a tie of the fp32 sums


def test_det_batch_and_capacity(hip_lib):
    cfg, sd, m = build("DET-TINY", 128, torch.float32, max_batch=3)
    x = do.normalise_pages(make_pages(3, 128, seed=7)).cuda()
    full = m.forward(x)
    one = m.forward(x[1:2].contiguous())
    assert torch.equal(full[1:2], one)                  # pages are independent: batch composition changes nothing
    with pytest.raises(AssertionError):
        m.forward(torch.zeros(4, 3, 128, 128, device="cuda"))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_det_forward_from_uint8_pages_is_bit_identical(hip_lib, dtype):
    """surya_det_forward_u8 (rescale + normalise inside the first layout kernel) == surya_det_forward on the host-normalised
    pixel_values of the same pages (SegformerImageProcessor arithmetic, surya/detection/processor.py:126-146)."""
    import numpy as np
    from surya_amd.detection.predictor import SegformerImageProcessor
    cfg, sd, m = build("DET-TINY", 128, dtype)
    pages = make_pages(3, 128, seed=11)
    proc = SegformerImageProcessor({"height": 128, "width": 128})
    x = torch.from_numpy(np.stack([proc(p)["pixel_values"][0] for p in pages])).cuda().contiguous()
    ref = m.forward(x)
    u8 = torch.from_numpy(np.stack(pages)).cuda().contiguous()
    got = m.forward_u8(u8, proc.image_mean, proc.image_std)
    assert torch.equal(got, ref)
    # the same pages in PIL's RGBX memory layout (pixel stride 4, junk in the fourth byte)
    x4 = np.concatenate([np.stack(pages), np.random.default_rng(0).integers(0, 256, size=(3, 128, 128, 1), dtype=np.uint8)], 3)
    got4 = m.forward_u8(torch.from_numpy(x4).cuda().contiguous(), proc.image_mean, proc.image_std)
    assert torch.equal(got4, ref)


@pytest.mark.parametrize("pages_n,size", [(2, 1024), (3, 672)])
def test_persistent_conv_loop_is_bit_identical_to_the_one_tile_kernel(hip_lib, pages_n, size):
    """The 3 x 3 convolutions of the large stages run on the persistent 8-phase loop with the gather in its request stream (sa::Tuning
    conv_persist: bit 0 = the Cin % 64 == 0 convolutions, bit 1 = the Cin = 32 one with two taps per K-tile; default 3) or on the one-tile
    2-stage kernel (0): same K order, same MFMA, the virtual zero K-tile of an odd K-tile count adds 0 * 0 -- DET-DEFAULT bf16 heat maps
    must agree bit for bit, run to run and across the kernels. 672^2 x 3 pages leaves the last 256-row tile of every such convolution ragged."""
    import ctypes as C
    from surya_amd import _lib as L
    cfg, sd, m = build("DET-DEFAULT", size, torch.bfloat16, max_batch=pages_n)
    x = do.normalise_pages(list(make_pages(pages_n, size, seed=99))).cuda().contiguous()
    outs = {}
    try:
        for v in (0, 1, 3, 0, 1, 3):
            L.check(hip_lib.surya_set_tuning(b"conv_persist", C.c_int(v)), "surya_set_tuning")
            h = m.forward(x).clone()
            torch.cuda.synchronize()
            if v in outs:
                assert torch.equal(outs[v].view(torch.int32), h.view(torch.int32)), f"conv_persist={v}: not run-to-run identical"
            outs[v] = h
    finally:
        L.check(hip_lib.surya_set_tuning(b"conv_persist", C.c_int(3)), "surya_set_tuning")
    assert torch.isfinite(outs[3]).all()
    assert torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32))
    assert torch.equal(outs[0].view(torch.int32), outs[3].view(torch.int32))
