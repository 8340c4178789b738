"""A checkpoint DIRECTORY through RecognitionModelLoader (surya_amd/recognition/predictor.py; mirrors the reference's
surya/recognition/loader.py:25-82): config.json + *.safetensors + the Qwen2 BPE files + special_ocr_tokens. No real checkpoint is
available offline, so the directory is synthetic (tests/ckpt_util.py; its writer is checked against the reference's own
save_pretrained in tests/test_oracle_vs_reference.py). CPU: config, token-id layout, tensors; GPU: the loaded model's tokens == the
oracle's on the same weights, through the predictor's own scheduler."""
import warnings

import numpy as np
import pytest
import torch

import ckpt_util as cu
from surya_amd.synth import make_rec_weights, make_line_crops


@pytest.fixture(scope="module")
def ckpt(tmp_path_factory):
    cfg = cu.tiny_checkpoint_config()
    sd = make_rec_weights(cfg, 3)
    return cfg, sd, cu.write_rec_checkpoint(str(tmp_path_factory.mktemp("rec_ckpt")), cfg, sd)


def test_loader_reads_config_tokenizer_and_tensors(ckpt):
    from surya_amd.recognition.predictor import RecognitionModelLoader
    cfg, sd, path = ckpt
    ld = RecognitionModelLoader(path)
    ld._resolve()
    assert ld._cfg.encoder == cfg.encoder and ld._cfg.decoder == cfg.decoder
    assert (ld._cfg.bbox_size, ld._cfg.num_register_tokens, ld._cfg.image_embed_encoding_size) == (1025, 4, 1024)
    assert set(ld._sd) == set(sd) and all(torch.equal(ld._sd[k], sd[k]) for k in sd)
    tok = ld.tokenizer()
    # the id layout is defined by the files: BPE ids first, then the tags in the order of special_ocr_tokens["all"], then UTF-16
    n_tags = len(cu.special_ocr_tokens()["all"])
    assert tok.qwen_offset == cu.N_BPE and tok.special_token_offset == cu.N_BPE + n_tags and tok.vocab_size == cfg.decoder.vocab_size
    assert tok.system_tokens["</S>"] == cu.N_BPE and tok.system_tokens["<IMAGE>"] == cu.N_BPE + 2
    text = 'a<b>x</b><math display="inline">\\frac{1}{2} x^2</math>é\U0001d11e'
    ids = tok([text], ["ocr_with_boxes"])["input_ids"][0]
    assert tok.decode(ids) == text
    math = [i for i in ids if i < tok.qwen_offset]
    assert 259 in math                                   # "\frac" went through the checkpoint's BPE merges, not byte by byte
    proc = ld.processor()
    assert proc.ocr_tokenizer.special_token_offset == tok.special_token_offset


def test_padded_and_short_lm_head(tmp_path):
    """decoder.vocab_size may exceed the tokenizer's id count (padded lm_head: accepted with a warning; ids beyond the tokenizer
    decode to nothing instead of raising); a head SMALLER than the tokenizer is a layout error."""
    from surya_amd.recognition.predictor import RecognitionModelLoader
    cfg = cu.tiny_checkpoint_config(pad_vocab=12)
    path = cu.write_rec_checkpoint(str(tmp_path / "padded"), cfg, {"x": torch.zeros(1)})
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        tok = RecognitionModelLoader(path).tokenizer()
    assert any("padded lm_head" in str(x.message) for x in w) and tok.vocab_size == cfg.decoder.vocab_size - 12
    small = cu.tiny_checkpoint_config(pad_vocab=-8)
    path = cu.write_rec_checkpoint(str(tmp_path / "short"), small, {"x": torch.zeros(1)})
    with pytest.raises(ValueError):
        RecognitionModelLoader(path).tokenizer()


@pytest.mark.gpu
def test_predictor_from_checkpoint_dir_matches_oracle(hip_lib, ckpt):
    from surya_amd.recognition.predictor import RecognitionPredictor, RecognitionModelLoader
    from surya_amd.settings import settings
    from test_gpu_predictors import oracle_tokens
    cfg, sd, path = ckpt

    class Loader(RecognitionModelLoader):
        def model(self, device=None, dtype_=None, **caps):
            return super().model("cuda:0", torch.float32, max_slots=4, max_kv_len=256, max_patches=8192, max_prefill_tokens=1024)

    class Pred(RecognitionPredictor):
        model_loader_cls = Loader
        batch_size = 4

    settings.RECOGNITION_MAX_TOKENS = 10
    try:
        pred = Pred(checkpoint=path)
        assert pred.model.vocab == cfg.decoder.vocab_size
        crops = [c.astype(np.float32) for c in make_line_crops(6, seed=9)]
        crops.sort(key=lambda c: -c.shape[1])
        prep, ref = oracle_tokens(cfg, sd, pred, crops, 10)
        toks, boxes, scores = pred.generate(prep, 4)
        assert [list(t) for t in toks] == ref
    finally:
        settings.RECOGNITION_MAX_TOKENS = None


# ------------------------------------------------------------------------------------------------------------------ detector
@pytest.fixture(scope="module")
def det_ckpt(tmp_path_factory):
    from surya_amd.config import det_config
    from surya_amd.synth import make_det_weights
    cfg = det_config("DET-TINY")
    sd = make_det_weights(cfg, 5)
    return cfg, sd, cu.write_det_checkpoint(str(tmp_path_factory.mktemp("det_ckpt")), cfg, sd, size=256)


def test_det_loader_reads_directory(det_ckpt):
    import dataclasses
    from surya_amd.detection.predictor import DetectionModelLoader
    cfg, sd, path = det_ckpt
    ld = DetectionModelLoader(path)
    assert dataclasses.replace(ld._cfg, name=cfg.name) == cfg and ld._size == 256
    assert set(ld._sd) == set(sd) and all(torch.equal(ld._sd[k], sd[k]) for k in sd)
    p = ld.processor()
    assert p.size == {"height": 256, "width": 256} and np.allclose(p.image_mean, [0.485, 0.456, 0.406])


@pytest.mark.gpu
def test_detection_predictor_from_checkpoint_dir(hip_lib, det_ckpt):
    """The directory-loaded detector yields the same boxes as the dict-loaded one (same weights)."""
    from PIL import Image
    from surya_amd.detection.predictor import DetectionPredictor
    from surya_amd.synth import make_pages
    cfg, sd, path = det_ckpt
    pages = [Image.fromarray(p) for p in make_pages(2, 256, seed=13)]
    a = DetectionPredictor(checkpoint=path, dtype=torch.float32)(pages)
    b = DetectionPredictor(checkpoint={"config": cfg, "state_dict": sd, "size": 256}, dtype=torch.float32)(pages)
    assert [[x.polygon for x in r.bboxes] for r in a] == [[x.polygon for x in r.bboxes] for r in b]
    assert sum(len(r.bboxes) for r in a) > 0


def _write_family_dir(path, cfg, sd, preprocessor=None):
    """config.json with `encoder` / `decoder` sub-dicts + model.safetensors, the layout / table-rec on-disk format
    (surya/layout/loader.py:31-50, surya/table_rec/loader.py:33-46); checked against the reference's own writer in
    tests/test_oracle_vs_reference.py::test_live_layout_and_table_checkpoint_directories_written_by_the_reference."""
    import dataclasses, json, os
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    raw = {"model_type": "vision-encoder-decoder", "is_encoder_decoder": True,
           "encoder": {k: (list(v) if isinstance(v, tuple) else v) for k, v in dataclasses.asdict(cfg.encoder).items()},
           "decoder": {k: (list(v) if isinstance(v, tuple) else v) for k, v in dataclasses.asdict(cfg.decoder).items()}}
    raw["encoder"]["model_type"] = "donut-swin"
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(raw, f)
    save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(path, "model.safetensors"))
    if preprocessor:
        with open(os.path.join(path, "preprocessor_config.json"), "w") as f:
            json.dump(preprocessor, f)
    return path


def test_layout_and_table_loaders_read_directories(tmp_path):
    import dataclasses
    import numpy as np
    from surya_amd.layout.config import layout_config
    from surya_amd.layout.predictor import LayoutModelLoader
    from surya_amd.synth import make_layout_weights, make_table_weights
    from surya_amd.table_rec.config import table_config
    from surya_amd.table_rec.predictor import TableRecModelLoader
    cfg = layout_config("LAYOUT-TINY")
    sd = make_layout_weights(cfg, 1)
    ld = LayoutModelLoader(_write_family_dir(str(tmp_path / "layout"), cfg, sd))
    assert dataclasses.replace(ld._cfg, name=cfg.name) == cfg
    assert set(ld._sd) == set(sd) and all(torch.equal(ld._sd[k], sd[k]) for k in sd)
    assert ld.processor().max_size == {"height": 128, "width": 128}
    tcfg = table_config("TABLE-TINY")
    tsd = make_table_weights(tcfg, 2)
    tl = TableRecModelLoader(_write_family_dir(str(tmp_path / "table"), tcfg, tsd,
                                               {"image_mean": [0.4, 0.5, 0.6], "image_std": [0.2, 0.25, 0.3], "image_processor_type": "SuryaEncoderImageProcessor"}))
    assert dataclasses.replace(tl._cfg, name=tcfg.name) == tcfg
    assert all(torch.equal(tl._sd[k], tsd[k]) for k in tsd)
    p = tl.processor()
    assert np.allclose(p.image_processor.image_mean, [0.4, 0.5, 0.6]) and np.allclose(p.image_processor.image_std, [0.2, 0.25, 0.3])
    with pytest.raises(FileNotFoundError):
        import os
        os.remove(str(tmp_path / "table" / "model.safetensors"))
        TableRecModelLoader(str(tmp_path / "table"))


@pytest.mark.gpu
def test_table_predictor_from_checkpoint_dir(hip_lib, tmp_path):
    """Directory -> loader -> HIP model -> TableRecPredictor.__call__ (fp32): the same results as the in-memory checkpoint."""
    import numpy as np
    from PIL import Image
    from surya_amd.synth import make_table_weights
    from surya_amd.table_rec import predictor as tp
    from surya_amd.table_rec.config import table_config
    cfg = table_config("TABLE-TINY")
    sd = make_table_weights(cfg, 0)
    sd["decoder.box_property_heads.category.weight"][5 + 1] *= 3.0
    sd["decoder.box_property_heads.category.weight"][5 + 2] *= 2.5
    path = _write_family_dir(str(tmp_path / "table"), cfg, sd)
    rng = np.random.default_rng(3)
    pages = [Image.fromarray(rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)) for h, w in ((200, 320), (128, 128))]
    old = tp.TABLE_REC_MAX_BOXES
    tp.TABLE_REC_MAX_BOXES = 10
    try:
        a = tp.TableRecPredictor(checkpoint=path, dtype=torch.float32)(pages)
        b = tp.TableRecPredictor(checkpoint={"config": cfg, "state_dict": sd}, dtype=torch.float32)(pages)
    finally:
        tp.TABLE_REC_MAX_BOXES = old
    assert [r.model_dump() for r in a] == [r.model_dump() for r in b]


def test_config_json_architecture_switches_are_checked_not_dropped():
    """A config.json key the HIP engine does not read must carry the value it implements: a decoder without cross attention
    in some layer, a non-causal decoder, a Swin without qkv bias or an encoder config without encoder_length raise a clear ValueError
    instead of loading and computing something else; inference no-ops (dropout rates) and matching values pass."""
    import dataclasses
    from surya_amd.layout.config import layout_config, layout_config_from_reference_json
    from surya_amd.table_rec.config import table_config, table_config_from_reference_json
    cfg = layout_config("LAYOUT-TINY")
    enc, dec = dataclasses.asdict(cfg.encoder), dataclasses.asdict(cfg.decoder)
    n = dec["num_hidden_layers"]
    ok = {"encoder": {**enc, "drop_path_rate": 0.1, "qkv_bias": True, "hidden_act": "gelu"},
          "decoder": {**dec, "cross_attn_layers": list(range(n)), "double_residual_flow": True, "attention_dropout": 0.1, "causal": True}}
    assert dataclasses.replace(layout_config_from_reference_json(ok), name=cfg.name) == cfg
    for where, key, value in (("decoder", "cross_attn_layers", [0]), ("decoder", "causal", False), ("decoder", "double_residual_flow", False),
                              ("decoder", "hidden_activation", "silu"), ("encoder", "qkv_bias", False), ("encoder", "use_absolute_embeddings", True)):
        bad = {"encoder": dict(ok["encoder"]), "decoder": dict(ok["decoder"])}
        bad[where][key] = value
        with pytest.raises(ValueError, match=key):
            layout_config_from_reference_json(bad)
    # keys the reference's config class stores but its forward pass never reads: a checkpoint carrying other values there
    # runs identically in the reference, so it must load here too
    stored_only = {"encoder": dict(ok["encoder"]), "decoder": {**ok["decoder"], "aux_heads": 2, "max_pause_tokens": 3, "tie_word_embeddings": True,
                                                               "block_types": ["attention", "recurrent"], "encoder_cross_attn_layers": [0]}}
    assert dataclasses.replace(layout_config_from_reference_json(stored_only), name=cfg.name) == cfg
    no_len = {"encoder": {k: v for k, v in enc.items() if k != "encoder_length"}, "decoder": dec}
    with pytest.raises(ValueError, match="encoder_length"):
        layout_config_from_reference_json(no_len)
    tcfg = table_config("TABLE-TINY")
    tenc, tdec = dataclasses.asdict(tcfg.encoder), dataclasses.asdict(tcfg.decoder)
    with pytest.raises(ValueError, match="double_residual_flow"):
        table_config_from_reference_json({"encoder": tenc, "decoder": {**tdec, "double_residual_flow": True}})
