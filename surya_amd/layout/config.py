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
    raise KeyError(name)


def _pick(cls, raw: dict, **extra):
    """Keep the keys of a reference sub-config dict that are fields of our dataclass (lists -> tuples)."""
    kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in raw.items() if k in cls.__dataclass_fields__}
    kw.update(extra)
    return cls(**kw)


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
