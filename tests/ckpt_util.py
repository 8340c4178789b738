"""Test helper: a synthetic recognition CHECKPOINT DIRECTORY in the reference's on-disk format (recognition/loader.py:25-82):
config.json (SuryaModelConfig.to_dict layout: vision_encoder / decoder sub-dicts, special_ocr_tokens, bbox_size ...),
model.safetensors (the reference's parameter names) and the Qwen2 BPE files the OCR tokenizer loads from the same directory
(processor/tokenizer.py:234: Qwen2Tokenizer.from_pretrained(checkpoint)): vocab.json, merges.txt, tokenizer_config.json.

The BPE is hand made and tiny (256 byte tokens of the GPT-2 / Qwen2 byte alphabet + a few merges + <|endoftext|>), which is what
makes the id layout testable: qwen_offset = len(tokenizer) = 265, then the tags of special_ocr_tokens["all"], then 65536 UTF-16
units. tests/test_oracle_vs_reference.py checks this writer against the reference's OWN save_pretrained (build container only)."""
from __future__ import annotations

import json
import os

import torch

from surya_amd.config import DecoderConfig, EncoderConfig, RecConfig
from surya_amd.recognition.tokenizer import DEFAULT_SPECIAL_TOKENS

MERGES = [("\\", "f"), ("r", "a"), ("\\f", "ra"), ("\\fra", "c"), ("x", "^"), ("Ġ", "+"), ("{", "1"), ("}", "{")]
N_BPE = 256 + len(MERGES) + 1            # byte tokens + merges + <|endoftext|>


def bytes_to_unicode():
    """The GPT-2 / Qwen2 byte -> printable character table."""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, [chr(c) for c in cs]))


def write_tokenizer_files(path: str):
    b2u = bytes_to_unicode()
    vocab = {b2u[i]: i for i in range(256)}
    lines = []
    for a, b in MERGES:
        lines.append(f"{a} {b}")
        vocab[a + b] = len(vocab)
    with open(os.path.join(path, "vocab.json"), "w") as f:
        json.dump(vocab, f)
    with open(os.path.join(path, "merges.txt"), "w") as f:
        f.write("#version: 0.2\n" + "\n".join(lines) + "\n")
    with open(os.path.join(path, "tokenizer_config.json"), "w") as f:
        json.dump({"tokenizer_class": "Qwen2Tokenizer", "model_max_length": 4096, "unk_token": None, "eos_token": "<|endoftext|>",
                   "pad_token": "<|endoftext|>",
                   "added_tokens_decoder": {str(len(vocab)): {"content": "<|endoftext|>", "special": True, "lstrip": False,
                                                             "rstrip": False, "normalized": False, "single_word": False}}}, f)


def special_ocr_tokens() -> dict:
    return {k: list(v) for k, v in DEFAULT_SPECIAL_TOKENS.items()}


def tiny_checkpoint_config(pad_vocab: int = 0) -> RecConfig:
    """REC-TINY's layers with the vocabulary a real id layout needs: N_BPE + tags + 65536 UTF-16 units (+ optional lm_head padding)."""
    n_tags = len(dict.fromkeys(DEFAULT_SPECIAL_TOKENS["all"]))
    vocab = N_BPE + n_tags + 65536 + pad_vocab
    enc = EncoderConfig(depth=2, hidden_size=128, intermediate_size=256, num_heads=4, out_hidden_size=128, fullatt_block_indexes=(1,))
    dec = DecoderConfig(vocab_size=vocab, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4,
                        num_key_value_heads=2, head_dim=32)
    return RecConfig(name="checkpoint", encoder=enc, decoder=dec, qwen_offset=N_BPE, num_special_tokens=n_tags)


def reference_config_dict(cfg: RecConfig, special: dict) -> dict:
    """config.json as SuryaModelConfig.to_dict() lays it out (surya/common/surya/config.py:12-71), restricted to what defines the
    model; the reference writes more bookkeeping keys (transformers version, architectures ...), which no loader reads."""
    e, d = cfg.encoder, cfg.decoder
    return {
        "model_type": "surya-multimodal-foundation", "vocab_size": d.vocab_size, "bbox_size": cfg.bbox_size,
        "num_register_tokens": cfg.num_register_tokens, "image_embed_encoding_size": cfg.image_embed_encoding_size,
        "image_embed_encoding_multiplier": cfg.image_embed_encoding_multiplier, "special_ocr_tokens": special,
        "vision_encoder": {"depth": e.depth, "hidden_size": e.hidden_size, "intermediate_size": e.intermediate_size,
                           "num_heads": e.num_heads, "in_channels": e.in_channels, "patch_size": e.patch_size,
                           "spatial_merge_size": e.spatial_merge_size, "temporal_patch_size": e.temporal_patch_size,
                           "window_size": e.window_size, "out_hidden_size": e.out_hidden_size,
                           "fullatt_block_indexes": list(e.fullatt_block_indexes)},     # (the encoder's eps is a constant, :424)
        "decoder": {"vocab_size": d.vocab_size, "hidden_size": d.hidden_size, "intermediate_size": d.intermediate_size,
                    "num_hidden_layers": d.num_hidden_layers, "num_attention_heads": d.num_attention_heads,
                    "num_key_value_heads": d.num_key_value_heads, "head_dim": d.head_dim, "rope_theta": d.rope_theta,
                    "rms_norm_eps": d.rms_norm_eps},
    }


def write_rec_checkpoint(path: str, cfg: RecConfig, sd: dict, special: dict | None = None) -> str:
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    special = special or special_ocr_tokens()
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(reference_config_dict(cfg, special), f, indent=1)
    save_file({k: v.contiguous() for k, v in sd.items() if torch.is_tensor(v)}, os.path.join(path, "model.safetensors"))
    write_tokenizer_files(path)
    return path


def write_det_checkpoint(path: str, cfg, sd: dict, size: int = 256) -> str:
    """Detector directory as the reference reads it (surya/detection/loader.py:23-63): config.json = EfficientViTConfig.to_dict()
    keys that define the network, model.safetensors, preprocessor_config.json = SegformerImageProcessor's (size, mean, std)."""
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump({"model_type": "efficientvit", "num_channels": cfg.num_channels, "widths": list(cfg.widths), "depths": list(cfg.depths),
                   "strides": list(cfg.strides), "head_dim": cfg.head_dim, "layer_norm_eps": cfg.layer_norm_eps,
                   "decoder_layer_hidden_size": cfg.decoder_layer_hidden_size, "decoder_hidden_size": cfg.decoder_hidden_size,
                   # (HF keeps the label count as id2label and omits it at the default of 2 labels -- the detector's own count)
                   **({} if cfg.num_labels == 2 else {"id2label": {str(i): f"LABEL_{i}" for i in range(cfg.num_labels)}})}, f, indent=1)
    save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(path, "model.safetensors"))
    with open(os.path.join(path, "preprocessor_config.json"), "w") as f:
        json.dump({"do_resize": True, "size": {"height": size, "width": size}, "do_rescale": True, "rescale_factor": 1 / 255,
                   "do_normalize": True, "image_mean": [0.485, 0.456, 0.406], "image_std": [0.229, 0.224, 0.225],
                   "image_processor_type": "SegformerImageProcessor"}, f, indent=1)
    return path
