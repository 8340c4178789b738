"""CPU: host-side logic of the recognition boundary (tokenizer, processor, post-processing, geometry, ABI surface)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from surya_amd.common.geometry import PolygonBox
from surya_amd.common import imageops
from surya_amd.recognition import postprocess as pp
from surya_amd.recognition.processor import SuryaOCRProcessor
from surya_amd.recognition.schema import TextChar, TextLine, TaskNames
from surya_amd.recognition.tokenizer import OCRTokenizer, ByteMathTokenizer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_exports_every_declared_symbol(hip_lib):
    hdr = open(os.path.join(ROOT, "include", "surya_amd.h")).read()
    names = set(re.findall(r"\b(surya_[a-z_0-9]+)\s*\(", hdr))
    assert len(names) >= 14
    for n in names:
        assert hasattr(hip_lib, n), f"{n} declared in include/surya_amd.h but not exported"


def test_create_rejects_bad_config(hip_lib):
    from surya_amd import _lib as L
    cfg = L.RecConfigC(dtype=7)
    h = C.c_void_p()
    assert hip_lib.surya_rec_create(C.byref(cfg), None, 0, C.byref(h)) == -3        # SA_ERR_UNSUPPORTED
    assert hip_lib.surya_rec_workspace_bytes(C.byref(cfg)) == 0


def test_tokenizer_roundtrip_three_ranges():
    tk = OCRTokenizer(None, ByteMathTokenizer(256), reserve_special=64)
    text = 'Hi <b>x</b> <math display="inline">a^2</math> \U0001F600'
    ids = tk(text, TaskNames.ocr_with_boxes)["input_ids"][0]
    assert any(i < tk.qwen_offset for i in ids)                                   # math bytes
    assert any(tk.qwen_offset <= i < tk.special_token_offset for i in ids)        # tags
    assert sum(i >= tk.special_token_offset for i in ids) >= 7                    # UTF-16 units (emoji = 2)
    assert tk.decode(ids, task=TaskNames.ocr_with_boxes) == text
    assert tk.vocab_size == tk.special_token_offset + 65536
    assert tk.decode([tk.special_token_offset - 1], task=TaskNames.ocr_with_boxes) == "<RSV-63>"   # reserved placeholder tag


def test_processor_tile_order_and_prompt():
    tk = OCRTokenizer(None, ByteMathTokenizer(256), reserve_special=64)
    proc = SuryaOCRProcessor(tk)
    # index image: pixel value = patch row-major index; 56 x 84 -> grid 4 x 6 (no resize needed)
    gh, gw = 4, 6
    img = np.zeros((gh * 14, gw * 14, 3), np.float32)
    for r in range(gh):
        for c in range(gw):
            img[r * 14:(r + 1) * 14, c * 14:(c + 1) * 14] = r * gw + c
    tiles, grid = proc.process_and_tile(img)
    assert grid == (gh, gw) and tiles.shape == (24, 588)
    raw = np.rint((tiles[:, 0] * 0.229 + 0.485) * 255).astype(int).tolist()
    # merge-block-major: (0,0),(0,1),(1,0),(1,1),(0,2),(0,3),(1,2),(1,3) ...
    assert raw[:8] == [0, 1, 6, 7, 2, 3, 8, 9]
    ids = proc.prompt_ids(6, TaskNames.ocr_with_boxes, "", math_mode=False)
    s = tk.system_tokens
    assert ids == [s["<IMAGE>"]] * 6 + [s["<REG1>"], s["<REG2>"], s["<REG3>"], s["<REG4>"], s["<OCR-WB>"], s["<NO-MATH>"], s["<EOI>"]]


def test_scale_to_fit_area_budget():
    f = SuryaOCRProcessor.scale_to_fit
    a = f(np.zeros((64, 512, 3), np.float32), (1024, 256))
    assert a.shape[:2] == (64, 512)                       # 32768 px: inside [168^2, 1024*256]
    b = f(np.zeros((64, 128, 3), np.float32), (1024, 256))
    assert b.shape[:2] == (119, 238)                      # ceil(64*s), ceil(128*s), s = sqrt(28224/8192)
    c = f(np.zeros((600, 1000, 3), np.float32), (1024, 256))
    assert c.shape[0] * c.shape[1] <= 1024 * 256


def test_resize_identity_and_constant():
    img = np.random.default_rng(0).random((20, 30, 3)).astype(np.float32)
    assert np.array_equal(imageops.resize(img, 30, 20, "cubic"), img)
    const = np.full((10, 12, 3), 7.0, np.float32)
    for kind in ("cubic", "lanczos4"):
        out = imageops.resize(const, 25, 31, kind)
        assert out.shape == (31, 25, 3) and np.allclose(out, 7.0, atol=1e-5)


def test_fill_poly_mask_rectangle_and_triangle():
    m = imageops.fill_poly_mask(10, 10, [(2, 2), (7, 2), (7, 6), (2, 6)])
    assert m[2:7, 2:8].all() and m.sum() == 5 * 6
    t = imageops.fill_poly_mask(10, 10, [(0, 0), (9, 0), (0, 9)])
    assert t[0, 0] and t[0, 9] and t[9, 0] and not t[9, 9]


def test_detect_repeat_token_rule():
    assert not pp.detect_repeat_token([1] * 39)
    assert pp.detect_repeat_token([1] * 40)
    assert pp.detect_repeat_token([1, 2] * 20)
    assert not pp.detect_repeat_token(list(range(40)))
    assert not pp.detect_repeat_token([1, 2, 3, 4, 5, 6] * 7)


def test_clean_math_tags_reference_cases():
    # the two weight-free cases of the reference's own tests (tests/test_recognition.py:55-67)
    assert pp.clean_math_tags("text <math><b>x</b></math> </math>") == "text <math>x</math> "
    assert pp.clean_math_tags("a <math> </math> b") == "a  b"
    assert pp.unwrap_math("<math>hello</math>") == "hello"
    assert pp.unwrap_math("<math>a+b</math>") == "<math>a+b</math>"


def test_prediction_to_polygon_batch_centre_box():
    pred = np.zeros((1, 1, 6), np.float32)
    pred[0, 0] = [512, 512, 100, 50, 512, 512]            # no skew (skew centred at bbox_size // 2)
    poly = pp.prediction_to_polygon_batch(pred, [(64, 512, 3)], 1025, 512)
    w, h = 512 / 1025, 64 / 1025
    assert np.allclose(poly[0, 0], [[462 * w, 487 * h], [562 * w, 487 * h], [562 * w, 537 * h], [462 * w, 537 * h]], atol=1e-4)


def test_polygon_box_semantics():
    b = PolygonBox(polygon=[10, 20, 110, 60])
    assert b.bbox == [10, 20, 110, 60] and b.width == 100 and b.height == 40
    b.expand(0, 0.05)
    assert b.bbox == [10, 18, 110, 62]
    b.rescale((100, 100), (50, 200))
    assert b.polygon[0] == [5, 36]
    with pytest.raises(ValueError):
        PolygonBox(polygon=[1, 2, 3])


def test_words_from_chars_and_fix_tags():
    line = PolygonBox(polygon=[0, 0, 100, 10])
    chars = [TextChar(text=c, polygon=[i * 10, 0, i * 10 + 8, 10], confidence=0.9) for i, c in enumerate("ab cd")]
    words = pp.words_from_chars(chars, line)
    assert [w.text for w in words] == ["ab", "cd"]
    st = {"formatting": ["<b>", "</b>", "<br>"], "math_external": ["<math>", "</math>"]}
    cs = [TextChar(text="<b>", polygon=[0, 0, 1, 1], bbox_valid=False), TextChar(text="x", polygon=[0, 0, 1, 1])]
    out = pp.fix_unbalanced_tags(cs, st)
    assert out[-1].text == "</b>"


# ------------------------------------------------------------------------------------------- detection host logic
def test_heatmap_boxes_on_synthetic_map():
    from surya_amd.detection import heatmap as hm
    m = np.full((128, 256), 0.05, np.float32)
    m[20:32, 30:200] = 0.9                      # a wide line
    m[60:70, 40:90] = 0.8                       # a short line
    m[100:103, 10:13] = 0.9                     # 9 px: below the area filter
    res = hm.parallel_get_boxes([m, m], (512, 256))
    assert len(res.bboxes) == 2 and res.image_bbox == [0, 0, 512, 256]
    b = sorted(res.bboxes, key=lambda b: b.bbox[1])[0]
    # map (256 wide, 128 tall) -> image (512, 256): x2 both ways; dilation grows the box by ~sqrt(12)+1 px before rescale
    assert 50 <= b.bbox[0] <= 60 and 398 <= b.bbox[2] <= 410 and 28 <= b.bbox[1] <= 40 and 64 <= b.bbox[3] <= 76
    assert b.polygon[0][0] < b.polygon[1][0] and b.polygon[1][1] < b.polygon[2][1]      # TL, TR, BR, BL order
    assert max(bb.confidence for bb in res.bboxes) == 1.0


def test_dilate_and_min_area_rect():
    from surya_amd.detection import heatmap as hm
    m = np.zeros((9, 9), bool)
    m[4, 4] = True
    assert hm.dilate_rect(m, 3).sum() == 9 and hm.dilate_rect(m, 3)[3:6, 3:6].all()
    d2 = hm.dilate_rect(m, 2)                   # even kernel: anchor k//2 = 1 -> window covers offsets [-1, 0]
    assert d2.sum() == 4 and d2[4:6, 4:6].all()
    pts = np.array([[0, 0], [10, 10], [12, 8], [2, -2]])        # a 45-degree rectangle
    box = hm.min_area_rect_points(pts)
    area = np.linalg.norm(box[0] - box[1]) * np.linalg.norm(box[1] - box[2])
    assert abs(area - 40.0) < 1e-3


def test_split_image_and_packing():
    from PIL import Image
    from surya_amd.detection.predictor import split_image, get_total_splits
    img = Image.new("RGB", (300, 2500), "white")
    parts, heights = split_image(img, 1024)
    assert [p.size for p in parts] == [(300, 1024)] * 3 and heights == [1024, 1024, 452]
    assert get_total_splits((300, 2500), 1024) == 3 and get_total_splits((300, 1400), 1024) == 1


def test_pair_interleave_keeps_qk_dot_products():
    """Vision qkv rows are stored with every q / k head pair-interleaved for the GEMM's RoPE epilogue; rotate_half on the
    original layout and the adjacent-pair rotation on the permuted layout give the same q.k scores, v is untouched."""
    import torch
    from surya_amd.recognition.weights import _pair_interleave_qk
    heads, d, n = 3, 16, 5
    g = torch.Generator().manual_seed(0)
    qkv = torch.randn(n, 3 * heads * d, generator=g, dtype=torch.float64)
    ang = torch.randn(n, d // 2, generator=g, dtype=torch.float64)
    cos, sin = ang.cos(), ang.sin()

    def rot_ref(x):                                   # rotate_half pairs element i with i + d/2
        x = x.reshape(n, heads, d)
        x1, x2 = x[..., : d // 2], x[..., d // 2:]
        return torch.cat([x1 * cos[:, None] - x2 * sin[:, None], x2 * cos[:, None] + x1 * sin[:, None]], -1)

    def rot_pairs(x):                                 # permuted layout: (2j, 2j + 1) = old (j, j + d/2)
        x = x.reshape(n, heads, d // 2, 2)
        a, b = x[..., 0], x[..., 1]
        return torch.stack([a * cos[:, None] - b * sin[:, None], b * cos[:, None] + a * sin[:, None]], -1).reshape(n, heads, d)

    hd = heads * d
    perm_cols = _pair_interleave_qk(torch.arange(3 * hd), heads, d)
    p = qkv[:, perm_cols]                             # what the GEMM with permuted weight rows produces
    assert torch.equal(p[:, 2 * hd:], qkv[:, 2 * hd:])
    q_ref, k_ref = rot_ref(qkv[:, :hd]), rot_ref(qkv[:, hd:2 * hd])
    q_new, k_new = rot_pairs(p[:, :hd]), rot_pairs(p[:, hd:2 * hd])
    s_ref = torch.einsum("ihd,jhd->hij", q_ref, k_ref)
    s_new = torch.einsum("ihd,jhd->hij", q_new, k_new)
    assert torch.allclose(s_ref, s_new, atol=1e-12)


def test_pmc_bucket_names_match_bench():
    """tools/rocpd_pmc.py maps mangled kernel names to the bucket labels bench.py prints (profiles/hbm_traffic.json keys)."""
    import importlib.util, os, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("rocpd_pmc", os.path.join(root, "tools", "rocpd_pmc.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    bench_src = open(os.path.join(root, "bench.py")).read()
    for name in ("_ZN2sa14gemm_nt_kernelIttLi64ELi64ELi2ELi2ELi0ELb1ELi4EEEvNS_8GemmArgsIT_T0_EE.kd",
                 "_ZN2sa14gemm_nt_kernelIttLi256ELi256ELi4ELi2ELi3ELb0ELi2EEEvNS_8GemmArgsIT_T0_EE.kd",
                 "_ZN2sa14gemm_nt_kernelItfLi128ELi128ELi2ELi2ELi6ELb0ELi2EEEvNS_8GemmArgsIT_T0_EE.kd",
                 "_ZN2sa16conv_gemm_kernelItLi128ELi128ELi2ELi2ELb1EEEvNS_8ConvArgsIT_EE.kd"):
        b = mod.bucket(name)
        assert b is not None and b in bench_src, (name, b)
    assert mod.bucket("_ZN2sa14rmsnorm_kernelItEEvPKT_lS3_PS1_lPKiiif.kd") is None
    import json
    keys = json.load(open(os.path.join(root, "profiles", "hbm_traffic.json"))).keys()
    assert all(k in bench_src for k in keys)
