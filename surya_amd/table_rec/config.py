"""Configuration and label vocabulary of the table-recognition model (second caller of the Donut-Swin + ADETR
family). Field names follow the reference's config classes so a checkpoint's config.json maps 1:1:
  constants -> surya/table_rec/model/config.py:11-44
  encoder   -> :72-137 (DonutSwinTableRecConfig)      decoder -> :140-230 (SuryaTableRecDecoderConfig)
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Tuple

from ..layout.config import SwinConfig

BOX_DIM = 1024
SPECIAL_TOKENS = 5
MAX_BOXES = 150
MERGE_KEYS = {"none": 0, "merge_up": 1, "merge_down": 2, "merge_both": 3}
MERGE_VALUES = [MERGE_KEYS["merge_up"], MERGE_KEYS["merge_down"], MERGE_KEYS["merge_both"]]
ID_TO_CATEGORY = {0: "Blank", 1: "Table-row", 2: "Table-column", 3: "Table-cell", 4: "Table"}
CATEGORY_TO_ID = {v: k for k, v in ID_TO_CATEGORY.items()}
ID_TO_HEADER = {0: "None", 1: "Header"}
HEADER_TO_ID = {v: k for k, v in ID_TO_HEADER.items()}
# (key, width, mode): the order of a token's 10 numbers and of the decoder's property heads
BOX_PROPERTIES = [("bbox", 6, "regression"), ("category", len(ID_TO_CATEGORY), "classification"), ("merges", len(MERGE_KEYS), "classification"),
                  ("colspan", 1, "regression"), ("is_header", len(ID_TO_HEADER), "classification")]
TOKEN_WIDTH = 10


@dataclass(frozen=True)
class TableDecoderConfig:
    num_hidden_layers: int = 6
    vocab_size: int = BOX_DIM + 1
    bbox_size: int = BOX_DIM
    hidden_size: int = 512
    property_embed_size: int = 64
    box_embed_size: int = 512 - 64
    intermediate_size: int = 4 * 512
    encoder_hidden_size: int = 1024
    num_attention_heads: int = 8
    num_key_value_heads: int = 4
    rms_norm_eps: float = 1e-6
    layer_norm_eps: float = 1e-5
    rope_theta: float = 10000.0
    pad_token_id: int = 0
    eos_token_id: int = 1
    bos_token_id: int = 1
    pause_token_id: int = 2
    query_end_token_id: int = 4
    special_token_count: int = SPECIAL_TOKENS
    ASSUMED_EXTRA = {"double_residual_flow": False}     # (class attribute, not a field) surya/table_rec/model/config.py:224

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    def head_widths(self):
        """(key, rows) of the decoder's box_property_heads in BOX_PROPERTIES order (classification heads carry the special tokens,
        table_rec/shaper.py:69-76)."""
        return [(k, n + (self.special_token_count if mode == "classification" else 0)) for k, n, mode in BOX_PROPERTIES]

    @property
    def category_count(self) -> int:          # rows of category_embed: the head width + special_token_count again (decoder.py:36-41)
        return dict(self.head_widths())["category"] + self.special_token_count

    @property
    def merge_count(self) -> int:
        return dict(self.head_widths())["merges"] + self.special_token_count


@dataclass(frozen=True)
class TableRecConfig:
    name: str = "TABLE-DEFAULT"
    encoder: SwinConfig = field(default_factory=lambda: SwinConfig(depths=(2, 2, 12, 2), encoder_length=1024))
    decoder: TableDecoderConfig = field(default_factory=TableDecoderConfig)


def table_config(name: str) -> TableRecConfig:
    name = name.upper()
    if name == "TABLE-DEFAULT":
        return TableRecConfig()
    if name == "TABLE-SMALL":       # all four Swin stages, GQA decoder, the concat embedding at reduced widths: the GPU parity config
        enc = SwinConfig(image_size=(256, 256), embed_dim=64, depths=(2, 2, 4, 2), num_heads=(2, 4, 8, 16), num_kv_heads=(2, 4, 8, 16),
                         encoder_length=96)
        dec = TableDecoderConfig(num_hidden_layers=3, hidden_size=256, property_embed_size=64, box_embed_size=192, intermediate_size=512,
                                 encoder_hidden_size=512, num_attention_heads=4, num_key_value_heads=2)
        return TableRecConfig(name="TABLE-SMALL", encoder=enc, decoder=dec)
    if name == "TABLE-TINY":
        enc = SwinConfig(image_size=(128, 128), embed_dim=64, depths=(2, 2), num_heads=(2, 4), num_kv_heads=(2, 2), encoder_length=256)
        dec = TableDecoderConfig(num_hidden_layers=2, hidden_size=64, property_embed_size=16, box_embed_size=48, intermediate_size=128,
                                 encoder_hidden_size=128, num_attention_heads=2, num_key_value_heads=1)
        return TableRecConfig(name="TABLE-TINY", encoder=enc, decoder=dec)
    raise KeyError(name)


def table_config_from_reference_json(raw: dict) -> TableRecConfig:
    """config.json of a table-recognition checkpoint (SuryaTableRecConfig.to_dict: `encoder` = DonutSwinTableRecConfig, `decoder` =
    SuryaTableRecDecoderConfig sub-dicts; surya/table_rec/loader.py:33-41 reads it the same way)."""
    from ..layout.config import _pick
    return TableRecConfig(name="checkpoint", encoder=_pick(SwinConfig, raw["encoder"]), decoder=_pick(TableDecoderConfig, raw["decoder"]))
