"""Configuration of the layout model family: Donut-Swin window-attention encoder + ADETR decoder with
cross- and self-attention. Field names follow the reference's config classes so a checkpoint's config.json maps 1:1:
  encoder -> surya/layout/model/config.py:77-131 (DonutSwinLayoutConfig)
  decoder -> surya/layout/model/config.py:134-226 (SuryaLayoutDecoderConfig)
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Tuple

ID_TO_LABEL = {0: "Blank", 1: "Text", 2: "TextInlineMath", 3: "Code", 4: "SectionHeader", 5: "Caption", 6: "Footnote", 7: "Equation",
               8: "ListItem", 9: "PageFooter", 10: "PageHeader", 11: "Picture", 12: "Figure", 13: "Table", 14: "Form",
               15: "TableOfContents", 16: "Handwriting"}        # surya/layout/model/config.py:16-34


@dataclass(frozen=True)
class SwinConfig:
    image_size: Tuple[int, int] = (768, 768)
    patch_size: int = 4
    num_channels: int = 3
    embed_dim: int = 128
    depths: Tuple[int, ...] = (2, 2, 16, 2)
    num_heads: Tuple[int, ...] = (4, 8, 16, 32)
    num_kv_heads: Tuple[int, ...] = (4, 8, 16, 32)
    window_size: int = 8
    mlp_ratio: float = 4.0
    layer_norm_eps: float = 1e-5
    encoder_length: int = 768
    use_positional_embeddings: bool = True

    @property
    def hidden_size(self) -> int:
        return int(self.embed_dim * 2 ** (len(self.depths) - 1))

    @property
    def grid(self) -> Tuple[int, int]:
        return self.image_size[0] // self.patch_size, self.image_size[1] // self.patch_size


@dataclass(frozen=True)
class LayoutDecoderConfig:
    num_hidden_layers: int = 8
    vocab_size: int = 1025
    bbox_size: int = 1024
    label_count: int = 20
    skew_scaler: int = 512
    special_token_count: int = 3
    hidden_size: int = 1024
    intermediate_size: int = 4096
    encoder_hidden_size: int = 1024
    num_attention_heads: int = 16
    num_key_value_heads: int = 4
    rms_norm_eps: float = 1e-6
    layer_norm_eps: float = 1e-5
    rope_theta: float = 10000.0
    pad_token_id: int = 0
    eos_token_id: int = 1
    bos_token_id: int = 1
    pause_token_id: int = 2
    pause_token_count: int = 0
    ASSUMED_EXTRA = {"double_residual_flow": True}      # (class attribute, not a field) surya/layout/model/config.py:221

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads


@dataclass(frozen=True)
class LayoutConfig:
    name: str = "LAYOUT-DEFAULT"
    encoder: SwinConfig = field(default_factory=SwinConfig)
    decoder: LayoutDecoderConfig = field(default_factory=LayoutDecoderConfig)


def layout_config(name: str) -> LayoutConfig:
    name = name.upper()
    if name == "LAYOUT-DEFAULT":
        return LayoutConfig()
    if name == "LAYOUT-SMALL":      # all four stages, shifted windows with masks, GQA decoder at reduced widths: the GPU parity config
        enc = SwinConfig(image_size=(256, 256), embed_dim=64, depths=(2, 2, 4, 2), num_heads=(2, 4, 8, 16), num_kv_heads=(2, 4, 8, 16),
                         encoder_length=96)
        dec = LayoutDecoderConfig(num_hidden_layers=3, hidden_size=256, intermediate_size=512, encoder_hidden_size=512,
                                  num_attention_heads=4, num_key_value_heads=2)
        return LayoutConfig(name="LAYOUT-SMALL", encoder=enc, decoder=dec)
    if name == "LAYOUT-TINY":
        enc = SwinConfig(image_size=(128, 128), embed_dim=64, depths=(2, 2), num_heads=(2, 4), num_kv_heads=(2, 2), encoder_length=256)
        dec = LayoutDecoderConfig(num_hidden_layers=2, hidden_size=64, intermediate_size=128, encoder_hidden_size=128,
                                  num_attention_heads=2, num_key_value_heads=1)
        return LayoutConfig(name="LAYOUT-TINY", encoder=enc, decoder=dec)
    if name == "LAYOUT-PAD":        # stage grids 44 x 52, 22 x 26, 11 x 13: none a multiple of the window -- every block pads (maybe_pad, donut/encoder.py:588-596)
        enc = SwinConfig(image_size=(176, 208), embed_dim=64, depths=(2, 2, 2), num_heads=(2, 4, 8), num_kv_heads=(2, 4, 4), encoder_length=160)
        dec = LayoutDecoderConfig(num_hidden_layers=2, hidden_size=128, intermediate_size=256, encoder_hidden_size=256,
                                  num_attention_heads=4, num_key_value_heads=2)
        return LayoutConfig(name="LAYOUT-PAD", encoder=enc, decoder=dec)
    raise KeyError(name)


# Architecture switches of the reference's config classes that the HIP engine does NOT read because it implements exactly one
# setting of each (csrc/layout_model.hip: every decoder layer has cross + self attention, all of it global and causal; GELU(tanh)
# gated MLP; bias-free attention projections; Swin with qkv bias, exact GELU, no absolute position embedding). A checkpoint that says
# otherwise would load and silently compute something else -- so a key that is present must carry the assumed value.
_ASSUMED_ENCODER = {"qkv_bias": True, "hidden_act": "gelu", "use_absolute_embeddings": False}     # (dropout / drop-path rates are inference no-ops)
# Only switches the reference's FORWARD PASS reads are hard checks. Keys its config class merely stores (aux_heads, max_pause_tokens,
# block_types, encoder_cross_attn_layers: no use outside surya/layout/model/config.py; tie_word_embeddings: overwritten with False by
# encoderdecoder.py:35-36) cannot change what the reference computes, so a checkpoint that carries other values there loads and runs
# identically in the reference -- refusing it here would not be drop-in behaviour.
_ASSUMED_DECODER = {"hidden_activation": "gelu_pytorch_tanh", "attention_bias": False, "causal": True}
_ALL_LAYERS = ("cross_attn_layers", "self_attn_layers", "global_attn_layers")


def _norm(v):
    return tuple(_norm(x) for x in v) if isinstance(v, (list, tuple)) else v


def check_assumed(raw: dict, assumed: dict, what: str, n_layers: int | None = None, extra: dict | None = None):
    """Raise a clear ValueError when a config.json key the engine ignores carries a value other than the one it implements."""
    bad = []
    for k, want in {**assumed, **(extra or {})}.items():
        if k in raw and raw[k] is not None and _norm(raw[k]) != _norm(want) and not (isinstance(want, (int, float)) and float(raw[k]) == float(want)):
            bad.append(f"{k} = {raw[k]!r} (the HIP engine implements {want!r})")
    if n_layers is not None:
        for k in _ALL_LAYERS:
            if k in raw and raw[k] is not None and not set(range(n_layers)) <= {int(x) for x in raw[k]}:
                bad.append(f"{k} = {list(raw[k])!r} (the HIP engine runs this in every one of the {n_layers} layers)")
    if bad:
        raise ValueError(f"{what}: unsupported architecture switches in config.json: " + "; ".join(bad))


def _pick(cls, raw: dict, **extra):
    """Keep the keys of a reference sub-config dict that are fields of our dataclass (lists -> tuples). Keys that change the
    architecture are not dropped silently: see check_assumed."""
    if cls is SwinConfig:
        check_assumed(raw, _ASSUMED_ENCODER, "Swin encoder")
    elif "num_hidden_layers" in raw or "hidden_activation" in raw:
        check_assumed(raw, _ASSUMED_DECODER, cls.__name__, n_layers=int(raw.get("num_hidden_layers", cls.__dataclass_fields__["num_hidden_layers"].default)),
                      extra=getattr(cls, "ASSUMED_EXTRA", None))
    kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in raw.items() if k in cls.__dataclass_fields__}
    kw.update(extra)
    cfg = cls(**kw)
    if cls is SwinConfig and "encoder_length" not in raw:
        raise ValueError("Swin encoder: config.json has no encoder_length (the position-embedding length of the encoder output)")
    return cfg


def layout_config_from_reference_json(raw: dict) -> LayoutConfig:
    """config.json of a layout checkpoint (SuryaLayoutConfig.to_dict: `encoder` = DonutSwinLayoutConfig, `decoder` =
    SuryaLayoutDecoderConfig sub-dicts; surya/layout/loader.py:31-40 reads it the same way)."""
    return LayoutConfig(name="checkpoint", encoder=_pick(SwinConfig, raw["encoder"]), decoder=_pick(LayoutDecoderConfig, raw["decoder"]))


def read_checkpoint_dir(path: str):
    """(config.json as a dict, merged tensors of every *.safetensors file, preprocessor_config.json as a dict or None) of a directory
    in the reference's on-disk format (surya/common/s3.py:50-65)."""
    import json
    import os
    from safetensors.torch import load_file
    with open(os.path.join(path, "config.json")) as f:
        raw = json.load(f)
    sd = {}
    for fn in sorted(os.listdir(path)):
        if fn.endswith(".safetensors"):
            sd.update(load_file(os.path.join(path, fn)))
    if not sd:
        raise FileNotFoundError(f"{path}: no *.safetensors file")
    pp = None
    if os.path.exists(os.path.join(path, "preprocessor_config.json")):
        with open(os.path.join(path, "preprocessor_config.json")) as f:
            pp = json.load(f)
    return raw, sd, pp
