"""Record layout-model fixtures from the REAL reference modules (build container only).

    python oracle/make_golden_layout.py

Imports VikParuchuri/surya @ v0.14.6's DonutSwinLayoutModel / SuryaLayoutDecoder from /root/reference through oracle/ref_shim,
loads the seeded synthetic weights (surya_amd.synth.make_layout_weights) into them, and records, per configuration
(LAYOUT-TINY, LAYOUT-SMALL, LAYOUT-DEFAULT, LAYOUT-PAD): the encoder output (a strided sample for the large one), and for N greedy steps of
the reference's own decode loop (surya/layout/__init__.py:95-131 call shapes: prefill=True at step 0, cache_position, the fed-back
(box * bbox_size, class) tokens) the class logits, the sigmoid bbox outputs and the fed-back tokens.
-> tests/golden/layout_{tiny,small,default}.pt. tests/test_oracle_golden.py pins oracle/layout_oracle.py to them on the CPU,
tests/test_gpu_layout.py the HIP path on the GPU box (where /root/reference does not exist)."""
from __future__ import annotations

import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch

from oracle import ref_shim

GOLD = os.path.join(ROOT, "tests", "golden")


def build_reference_layout(cfg, sd):
    cfgm, encm, decm = ref_shim.import_layout_modules()
    e, d = cfg.encoder, cfg.decoder
    enc_cfg = cfgm.DonutSwinLayoutConfig(image_size=e.image_size, embed_dim=e.embed_dim, depths=list(e.depths), num_heads=list(e.num_heads),
                                         num_kv_heads=list(e.num_kv_heads), window_size=e.window_size, encoder_length=e.encoder_length,
                                         layer_norm_eps=e.layer_norm_eps)
    L = tuple(range(d.num_hidden_layers))
    dec_cfg = cfgm.SuryaLayoutDecoderConfig(num_hidden_layers=d.num_hidden_layers, hidden_size=d.hidden_size, intermediate_size=d.intermediate_size,
                                            encoder_hidden_size=d.encoder_hidden_size, num_attention_heads=d.num_attention_heads,
                                            num_key_value_heads=d.num_key_value_heads, cross_attn_layers=L, encoder_cross_attn_layers=L,
                                            self_attn_layers=L, global_attn_layers=L, rms_norm_eps=d.rms_norm_eps,
                                            layer_norm_eps=d.layer_norm_eps, rope_theta=d.rope_theta)
    enc = encm.DonutSwinLayoutModel(enc_cfg).eval()
    dec = decm.SuryaLayoutDecoder(dec_cfg).eval()
    m1 = enc.load_state_dict({k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}, strict=False)
    m2 = dec.load_state_dict({k[len("decoder."):]: v for k, v in sd.items() if k.startswith("decoder.")}, strict=False)
    assert not [k for k in m1.missing_keys if "relative_position_index" not in k] and not m2.missing_keys, (m1, m2)
    assert not m1.unexpected_keys and not m2.unexpected_keys
    return enc, dec, dec_cfg


def layout_pixels(cfg, batch: int, seed: int) -> torch.Tensor:
    """Deterministic inputs at the processor boundary: normalised pixel_values [B, 3, H, W]."""
    return torch.randn(batch, 3, *cfg.encoder.image_size, generator=torch.Generator().manual_seed(seed))


def record(name: str, batch: int, steps: int, seed: int, enc_stride: int):
    from surya_amd.layout.config import layout_config
    from surya_amd.synth import make_layout_weights
    cfg = layout_config(name)
    d = cfg.decoder
    sd = make_layout_weights(cfg, 0)
    enc, dec, dec_cfg = build_reference_layout(cfg, sd)
    x = layout_pixels(cfg, batch, seed)
    t0 = time.time()
    with torch.inference_mode():
        h = enc(pixel_values=x)[0]
        dec.model._setup_cache(dec_cfg, batch, "cpu", torch.float32)
        boxes = torch.tensor([[[d.bos_token_id] * 7] + [[d.pause_token_id] * 7] * d.pause_token_count] * batch, dtype=torch.long)
        pos = torch.ones_like(boxes[0, :, 0]).cumsum(0) - 1
        cls_log, box_log, fed = [], [], []
        for step in range(steps):
            out = dec(input_boxes=boxes, encoder_hidden_states=h, cache_position=pos, use_cache=True, prefill=(step == 0))
            pos = pos[-1:] + 1
            bl, cl = out.bbox_logits[:, -1, :], out.class_logits[:, -1, :]
            cls_log.append(cl.clone()); box_log.append(bl.clone())
            boxes = torch.cat([(bl * d.bbox_size).unsqueeze(1), cl.argmax(-1).unsqueeze(1).unsqueeze(1)], dim=-1).to(torch.long)
            fed.append(boxes[:, 0].clone())
    print(f"{name}: reference encoder + {steps} decode steps, batch {batch}: {time.time() - t0:.1f}s; classes of image 0: "
          f"{[int(c[0].argmax()) for c in cls_log]}", flush=True)
    g = {"config": name, "batch": batch, "seed": seed, "steps": steps, "enc_stride": enc_stride, "encoder_out": h[:, ::enc_stride].clone(),
         "encoder_absmax": float(h.abs().max()), "class_logits": torch.stack(cls_log), "bbox_logits": torch.stack(box_log),
         "fed_tokens": torch.stack(fed)}
    torch.save(g, os.path.join(GOLD, "layout_" + name.split("-")[1].lower() + ".pt"))


def main():
    ref_shim.install_layout()
    record("LAYOUT-TINY", 3, 12, 11, 1)
    record("LAYOUT-SMALL", 4, 16, 12, 1)
    record("LAYOUT-DEFAULT", 2, 8, 13, 8)
    record("LAYOUT-PAD", 3, 8, 14, 1)          # stage grids that are not whole windows: the reference pads inside every block
    for f in sorted(os.listdir(GOLD)):
        if f.startswith("layout_"):
            print(f, os.path.getsize(os.path.join(GOLD, f)))


if __name__ == "__main__":
    main()
