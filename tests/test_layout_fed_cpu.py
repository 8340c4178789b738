"""CPU: the layout predictor's greedy loop on device-fed runs when pages finish at different steps.

The device applies the PageHeader / PageFooter re-label rule to EVERY row of the batch on every step; the host -- like the reference,
which `continue`s on finished rows before the rule (surya/layout/__init__.py:150-157) -- applies it to unfinished pages only. A page that
already emitted </S> and later produces a header class with a box in the page middle therefore gets different fed tokens on the two
sides. Nobody reads either (rows are independent sequences), so the call must go through; before the fix FedRuns compared all rows
and the whole call raised 'device-fed token differs'. The stand-in model below serves scripted logits and implements the KERNEL's
rule (csrc/layout_kernels.h, all rows) for the tokens it feeds itself."""
import numpy as np
import pytest
import torch
from PIL import Image

from surya_amd import _lib as L
from surya_amd.layout import predictor as lp
from surya_amd.layout.config import ID_TO_LABEL, layout_config
from surya_amd.layout.model import FedRuns


class ScriptedFedModel:
    """decode_steps / wait_steps / set_feedback / encode of HipLayoutModel on scripted per-step outputs."""
    device, dtype, max_batch = torch.device("cpu"), torch.float32, 8

    def __init__(self, cfg, cls_script, box_script, rule_on_all_rows=True):
        self.config = cfg
        self.cls, self.box = cls_script, box_script            # [steps, rows, labels], [steps, rows, 6]
        self.pos = 0
        self.runs = {}
        self.sizes = None
        self.rule_on_all_rows = rule_on_all_rows

    def encode(self, px):
        pass

    def set_feedback(self, page_sizes=None):
        self.sizes = np.asarray(page_sizes, np.int64)

    def _fed_token(self, step):
        d = self.config.decoder
        cls, box = self.cls[step], self.box[step]
        bp = box * d.bbox_size
        label = cls.argmax(-1)
        sp = d.special_token_count
        hf = [k + sp for k, v in ID_TO_LABEL.items() if v in ("PageHeader", "PageFooter")]
        nxt = np.concatenate([bp, label[:, None].astype(np.float32)], -1)
        polys = lp.polygons_of_predictions(nxt, self.sizes, d.bbox_size, d.skew_scaler)
        w, h = self.sizes[:, 0], self.sizes[:, 1]
        mid = (np.isin(label, hf) & (polys[:, 0, 1] < h * .8) & (polys[:, 2, 1] > h * .2) & (polys[:, 0, 0] < w * .8) & (polys[:, 2, 0] > w * .2))
        for r in np.nonzero(mid)[0]:
            lg = cls[r].copy()
            lg[label[r]] = 0
            nxt[r, 6] = lg.argmax()
        return nxt.astype(np.int64).astype(np.int32)

    def decode_steps(self, boxes, position, n_steps, ring=0):
        assert position == self.pos
        self.runs[ring] = (self.pos, n_steps)
        self.pos += n_steps

    def wait_steps(self, n_steps, ring=0):
        p0, n = self.runs.pop(ring)
        assert n == n_steps
        toks = np.stack([self._fed_token(p0 + i) for i in range(n)])
        return self.cls[p0:p0 + n], self.box[p0:p0 + n], toks


def _script(cfg, steps=48):      # longer than two device-fed runs: the run behind the current one is always enqueued
    d = cfg.decoder
    sp = d.special_token_count
    header = [k + sp for k, v in ID_TO_LABEL.items() if v == "PageHeader"][0]
    text = [k + sp for k, v in ID_TO_LABEL.items() if v == "Text"][0]
    cls = np.zeros((steps, 2, d.label_count), np.float32)
    box = np.full((steps, 2, 6), 0.5, np.float32)              # centre of the page, half its size: "in the page middle"
    box[..., 4:] = d.skew_scaler / d.bbox_size                 # no skew
    # page 0: </S> at step 0, then a PageHeader in the page middle at step 1 (second-best label: Text), then </S> again
    cls[0, 0, d.eos_token_id] = 5
    cls[1, 0, header] = 5; cls[1, 0, text] = 3
    cls[2:, 0, d.eos_token_id] = 5
    # page 1: three text boxes, a header in the middle (re-labelled on BOTH sides: the page is still live), then </S>
    cls[0:3, 1, text] = 5
    cls[3, 1, header] = 5; cls[3, 1, text] = 3
    cls[4:, 1, d.eos_token_id] = 5
    return cls, box, header, text


def _predictor(model):
    pred = lp.LayoutPredictor.__new__(lp.LayoutPredictor)
    pred.model = model
    pred.processor = lambda images: {"pixel_values": [np.zeros((3, 8, 8), np.float32) for _ in images]}
    return pred


def test_page_that_finished_early_then_emits_a_header_does_not_abort_the_call():
    cfg = layout_config("LAYOUT-TINY")
    cls, box, header, text = _script(cfg)
    pred = _predictor(ScriptedFedModel(cfg, cls, box))
    pages = [Image.new("RGB", (100, 100)), Image.new("RGB", (100, 100))]
    out = pred._detect_chunk(pages, [p.size for p in pages], cfg.decoder, top_k=3)
    assert [len(r.bboxes) for r in out] == [0, 4]
    assert [b.label for b in out[1].bboxes] == ["Text", "Text", "Text", "Text"]     # the live page's header was re-labelled to its second-best class


def test_a_live_row_that_differs_still_raises():
    """The comparison is only narrowed to live rows, not switched off: a stand-in whose rule differs on a LIVE page is caught."""
    cfg = layout_config("LAYOUT-TINY")
    cls, box, header, text = _script(cfg)
    m = ScriptedFedModel(cfg, cls, box)
    orig = m._fed_token
    def wrong(step):
        t = orig(step)
        if step == 1:
            t[1, 0] += 1                                    # page 1 is live at step 1
        return t
    m._fed_token = wrong
    pred = _predictor(m)
    pages = [Image.new("RGB", (100, 100)), Image.new("RGB", (100, 100))]
    with pytest.raises(L.SuryaAmdError, match="device-fed token differs"):
        pred._detect_chunk(pages, [p.size for p in pages], cfg.decoder, top_k=3)
