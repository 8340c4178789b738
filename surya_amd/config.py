"""Model configurations for the recognition / detection hot path.

Field names follow the reference's config classes so that a real checkpoint's ``config.json`` maps 1:1:
  * encoder  -> surya/common/surya/encoder/config.py:18-53  (SuryaEncoderConfig, Qwen2.5-VL style ViT)
  * decoder  -> surya/common/surya/decoder/config.py:28-85  (SuryaDecoderConfig, Qwen2 style LM)
  * model    -> surya/common/surya/config.py:12-71          (SuryaModelConfig)
  * detector -> surya/detection/model/config.py:12-52       (EfficientViTConfig)

The real checkpoints are not available offline, so the named synthetic configs below
(REC-TINY / REC-SMALL / REC-FULL, DET-TINY / DET-DEFAULT) are what tests and bench.py run.
"""
from __future__ import annotations

from dataclasses import dataclass, field, asdict
from typing import Tuple


@dataclass(frozen=True)
class EncoderConfig:
    depth: int = 8
    hidden_size: int = 1280
    intermediate_size: int = 3420
    num_heads: int = 16
    in_channels: int = 3
    patch_size: int = 14
    spatial_merge_size: int = 2
    temporal_patch_size: int = 1
    window_size: int = 112
    out_hidden_size: int = 1280
    fullatt_block_indexes: Tuple[int, ...] = (3, 7)
    rms_norm_eps: float = 1e-6

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_heads

    @property
    def patch_dim(self) -> int:
        return self.in_channels * self.temporal_patch_size * self.patch_size * self.patch_size


@dataclass(frozen=True)
class DecoderConfig:
    vocab_size: int = 81920
    hidden_size: int = 1280
    intermediate_size: int = 5120
    num_hidden_layers: int = 16
    num_attention_heads: int = 10
    num_key_value_heads: int = 2
    head_dim: int = 128
    rope_theta: float = 10000.0
    rms_norm_eps: float = 1e-6


@dataclass(frozen=True)
class RecConfig:
    name: str = "REC-FULL"
    encoder: EncoderConfig = field(default_factory=EncoderConfig)
    decoder: DecoderConfig = field(default_factory=DecoderConfig)
    bbox_size: int = 1025
    image_embed_encoding_size: int = 1024
    image_embed_encoding_multiplier: int = 256
    num_register_tokens: int = 4
    # Token-id layout mirrors SuryaOCRTokenizer (processor/tokenizer.py:224-260):
    # [0, qwen_offset) math BPE | [qwen_offset, special_token_offset) tags | [special_token_offset, +65536) UTF-16 units
    qwen_offset: int = 16256
    num_special_tokens: int = 128

    @property
    def special_token_offset(self) -> int:
        return self.qwen_offset + self.num_special_tokens

    @property
    def vocab_size(self) -> int:
        return self.decoder.vocab_size

    @property
    def hidden_size(self) -> int:
        return self.decoder.hidden_size

    # System tokens occupy the first special-token ids in this order (synthetic tokenizer; a real checkpoint's
    # ids come from config.special_ocr_tokens via SuryaOCRTokenizer.system_tokens, tokenizer.py:239-242).
    SYSTEM_TOKENS = ("</S>", "<PAD>", "<IMAGE>", "<REG1>", "<REG2>", "<REG3>", "<REG4>", "<EOI>", "<NOP>", "<ROT>",
                     "<NO-MATH>", "<OCR-WB>", "<OCR-WOB>", "<BLOCKS-WOB>")

    def token_id(self, tag: str) -> int:
        return self.qwen_offset + self.SYSTEM_TOKENS.index(tag)

    @property
    def eos_token_id(self) -> int:
        return self.token_id("</S>")

    @property
    def pad_token_id(self) -> int:
        return self.token_id("<PAD>")

    @property
    def image_token_id(self) -> int:
        return self.token_id("<IMAGE>")

    @property
    def nop_token_id(self) -> int:
        return self.token_id("<NOP>")

    def to_dict(self):
        return asdict(self)


@dataclass(frozen=True)
class DetConfig:
    name: str = "DET-DEFAULT"
    num_channels: int = 3
    widths: Tuple[int, ...] = (32, 64, 128, 256, 512)
    depths: Tuple[int, ...] = (1, 1, 1, 6, 6)
    strides: Tuple[int, ...] = (2, 2, 2, 2, 2)
    head_dim: int = 32
    layer_norm_eps: float = 1e-6
    decoder_layer_hidden_size: int = 128
    decoder_hidden_size: int = 512
    num_labels: int = 2

    def to_dict(self):
        return asdict(self)


def rec_config(name: str) -> RecConfig:
    """Named synthetic recognition configs."""
    name = name.upper()
    if name == "REC-FULL":
        return RecConfig()
    if name == "REC-SMALL":  # same op mix as FULL (head_dim 80 / 128, GQA 5:1, odd intermediate) at ~1/30 the weights
        enc = EncoderConfig(depth=4, hidden_size=320, intermediate_size=860, num_heads=4, out_hidden_size=256,
                            fullatt_block_indexes=(1, 3))
        dec = DecoderConfig(vocab_size=4096 + 65536, hidden_size=256, intermediate_size=768, num_hidden_layers=3,
                            num_attention_heads=10, num_key_value_heads=2, head_dim=128)
        return RecConfig(name="REC-SMALL", encoder=enc, decoder=dec, qwen_offset=4096 - 128)
    if name == "REC-TINY":
        enc = EncoderConfig(depth=2, hidden_size=128, intermediate_size=256, num_heads=4, out_hidden_size=128,
                            fullatt_block_indexes=(1,))
        dec = DecoderConfig(vocab_size=1024, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                            num_attention_heads=4, num_key_value_heads=2, head_dim=32)
        # tiny vocab: only the first (1024-256-64) UTF-16 units are representable
        return RecConfig(name="REC-TINY", encoder=enc, decoder=dec, qwen_offset=256, num_special_tokens=64)
    raise KeyError(name)


def det_config(name: str) -> DetConfig:
    name = name.upper()
    if name == "DET-DEFAULT":
        return DetConfig()
    if name == "DET-TINY":
        return DetConfig(name="DET-TINY", widths=(8, 16, 32, 64, 128), depths=(1, 1, 1, 2, 2), head_dim=16,
                         decoder_layer_hidden_size=32, decoder_hidden_size=64)
    raise KeyError(name)
