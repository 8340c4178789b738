"""GPU: the layout model family (Donut-Swin encoder + ADETR decoder) through the C ABI against fixtures recorded
from the REAL reference modules (oracle/make_golden_layout.py; the oracle restates them bit for bit, tests/test_oracle_golden.py).

fp32 reference mode: encoder output <= 2e-4 x max, class logits <= 2e-4 x max and sigmoid boxes <= 1e-5 at every teacher-forced decode
step, argmax classes bit-exact, and the free-running greedy loop reproduces the reference's fed-back tokens. bf16: encoder <= 3e-2 x
max, class logits within 4e-2 x max (teacher forced)."""
import os

import numpy as np
import pytest
import torch

from surya_amd.layout.config import layout_config
from surya_amd.synth import make_layout_weights

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = [("LAYOUT-TINY", "layout_tiny.pt"), ("LAYOUT-SMALL", "layout_small.pt"), ("LAYOUT-DEFAULT", "layout_default.pt"),
         ("LAYOUT-PAD", "layout_pad.pt")]          # LAYOUT-PAD: 176 x 208 pixels, stage grids 44 x 52 / 22 x 26 / 11 x 13 -- every Swin block pads to whole windows


def _pixels(cfg, batch, seed):
    return torch.randn(batch, 3, *cfg.encoder.image_size, generator=torch.Generator().manual_seed(seed))      # = make_golden_layout.layout_pixels


def _model(name, dtype, batch):
    from surya_amd.layout.model import HipLayoutModel
    cfg = layout_config(name)
    return cfg, HipLayoutModel(cfg, make_layout_weights(cfg, 0), dtype=dtype, max_batch=batch, max_boxes=32)


@pytest.mark.parametrize("name,fixture", CASES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_layout_encoder_and_teacher_forced_decoder(hip_lib, name, fixture, dtype):
    g = torch.load(os.path.join(GOLD, fixture))
    cfg, m = _model(name, dtype, g["batch"])
    d = cfg.decoder
    m.encode(_pixels(cfg, g["batch"], g["seed"]).cuda().contiguous())
    enc = m.encoder_states().float().cpu()
    tol_e = (2e-4 if dtype == torch.float32 else 3e-2) * g["encoder_absmax"]
    err_e = (enc[:, ::g["enc_stride"]] - g["encoder_out"]).abs().max().item()
    assert err_e <= tol_e, (err_e, tol_e)
    boxes = np.full((g["batch"], 7), d.bos_token_id, np.int32)
    worst_c = worst_b = 0.0
    for step in range(g["steps"]):
        cls, box = m.decode_step(boxes, step)
        ref_c, ref_b = g["class_logits"][step].numpy(), g["bbox_logits"][step].numpy()
        scale = max(1.0, float(np.abs(ref_c).max()))
        ec, eb = float(np.abs(cls - ref_c).max()), float(np.abs(box - ref_b).max())
        worst_c, worst_b = max(worst_c, ec / scale), max(worst_b, eb)
        if dtype == torch.float32:
            assert ec <= 2e-4 * scale and eb <= 1e-5, (step, ec, eb)
            assert np.array_equal(cls.argmax(-1), ref_c.argmax(-1)), step
        else:
            assert ec <= 4e-2 * scale and eb <= 2e-2, (step, ec, eb)
        boxes = g["fed_tokens"][step].numpy().astype(np.int32)
    print(f"{name} {dtype}: encoder err {err_e:.2e} (max {g['encoder_absmax']:.2f}), worst class-logit err {worst_c:.2e} x max, worst box err {worst_b:.2e}")


@pytest.mark.parametrize("name,fixture", CASES[:2] + CASES[3:])
def test_layout_free_running_tokens_fp32(hip_lib, name, fixture):
    """The greedy loop of LayoutPredictor (feed back box * bbox_size and the argmax class): fp32 mode reproduces the reference's tokens."""
    g = torch.load(os.path.join(GOLD, fixture))
    cfg, m = _model(name, torch.float32, g["batch"])
    d = cfg.decoder
    m.encode(_pixels(cfg, g["batch"], g["seed"]).cuda().contiguous())
    boxes = np.full((g["batch"], 7), d.bos_token_id, np.int32)
    for step in range(g["steps"]):
        cls, box = m.decode_step(boxes, step)
        nxt = np.concatenate([(torch.from_numpy(box) * d.bbox_size).numpy(), cls.argmax(-1)[:, None].astype(np.float32)], -1).astype(np.int64)
        ref = g["fed_tokens"][step].numpy()
        same = nxt == ref
        # a box coordinate may differ by one where box * 1024 sits within 2e-2 of an integer (truncation of an fp32-reordered value)
        raw = box * d.bbox_size
        for b, k in zip(*np.nonzero(~same)):
            assert k < 6 and abs(raw[b, k] - round(raw[b, k])) < 2e-2 and abs(int(nxt[b, k]) - int(ref[b, k])) == 1, (step, b, k, raw[b, k])
        boxes = ref.astype(np.int32)


def test_layout_predictor_call_and_schema(hip_lib):
    """LayoutPredictor end to end on synthetic pages (LAYOUT-SMALL, fp32): signature / schema of surya.layout.LayoutPredictor, one
    result per page with boxes inside the page, a large page sliced and re-joined, and the model-facing part of the loop equal to the
    oracle's greedy loop on the processor's own pixel_values."""
    from PIL import Image
    from oracle import layout_oracle as lo
    from surya_amd.layout.predictor import LayoutPredictor, LayoutModelLoader
    from surya_amd.layout.schema import LayoutResult
    from surya_amd.synth import make_pages
    cfg = layout_config("LAYOUT-SMALL")
    sd = make_layout_weights(cfg, 0)

    class Loader(LayoutModelLoader):
        def model(self, device=None, dtype=None, max_batch=None):
            return super().model("cuda:0", torch.float32, max_batch=4)

    class Pred(LayoutPredictor):
        model_loader_cls = Loader
        batch_size = 4

    pred = Pred(checkpoint={"config": cfg, "state_dict": sd})
    pages = [Image.fromarray(p) for p in make_pages(3, 512, seed=3)]
    pages.append(Image.fromarray(np.vstack(make_pages(2, 1024, seed=5))[:1800]))          # 1024 x 1800: two slices of 1200 / 600 rows
    out = pred(pages, top_k=3)
    assert len(out) == 4 and all(isinstance(r, LayoutResult) for r in out)
    assert out[3].sliced and out[3].image_bbox == [0, 0, 1024, 1800] and not out[0].sliced
    for r, p in zip(out, pages):
        for b in r.bboxes:
            assert b.label in set(__import__("surya_amd.layout.config", fromlist=["ID_TO_LABEL"]).ID_TO_LABEL.values())
            assert len(b.polygon) == 4 and 0 <= b.confidence <= 1 and len(b.top_k) <= 3
    assert sum(len(r.bboxes) for r in out) > 0
    # the tokens the loop fed back == the oracle's on the same pixel_values (first page batch, 6 steps)
    px = torch.from_numpy(np.stack(pred.processor(pages[:3])["pixel_values"]))
    _, steps = lo.generate(sd, cfg, px, 6)
    pred.model.encode(px.cuda().contiguous())
    boxes = np.full((3, 7), cfg.decoder.bos_token_id, np.int32)
    for k, st in enumerate(steps):
        cls, box = pred.model.decode_step(boxes, k)
        assert np.array_equal(cls.argmax(-1), st["class_preds"].numpy())
        nxt = torch.cat([st["box_preds"], st["class_preds"][:, None].float()], -1).to(torch.long).numpy()
        boxes = nxt.astype(np.int32)
