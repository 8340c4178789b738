"""Generate tests/golden/*.pt from the REAL reference modules (run in the build container only).

    python oracle/make_golden.py

Imports VikParuchuri/surya @ v0.14.6 from /root/reference through oracle/ref_shim, loads the seeded synthetic weights
(surya_amd.synth) into the reference's own nn.Modules and records their CPU fp32 outputs on seeded inputs. The
fixtures pin oracle/rec_oracle.py and oracle/det_oracle.py (tests/test_oracle_golden.py) and travel to the GPU box,
where /root/reference does not exist.
"""
from __future__ import annotations

import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch
import torch.nn.functional as F

from oracle import ref_shim

ref_shim.install()

from surya_amd.config import rec_config, det_config            # noqa: E402
from surya_amd.synth import make_rec_weights, make_det_weights, make_pages   # noqa: E402
from util import make_prompts, left_pad_batch                   # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
REC_GRIDS = [(6, 38), (10, 18), (8, 24), (4, 4), (2, 30)]
REC_STEPS = 10


def build_reference_rec(cfg, sd, attn="eager"):
    from surya.common.surya import SuryaModel
    from surya.common.surya.config import SuryaModelConfig
    from surya.common.surya.decoder.config import SuryaDecoderConfig
    from surya.common.surya.encoder.config import SuryaEncoderConfig
    e, d = cfg.encoder, cfg.decoder
    enc = SuryaEncoderConfig(depth=e.depth, hidden_size=e.hidden_size, intermediate_size=e.intermediate_size,
                             num_heads=e.num_heads, out_hidden_size=e.out_hidden_size,
                             fullatt_block_indexes=e.fullatt_block_indexes, window_size=e.window_size)
    dec = SuryaDecoderConfig(vocab_size=d.vocab_size, hidden_size=d.hidden_size, intermediate_size=d.intermediate_size,
                             num_hidden_layers=d.num_hidden_layers, num_attention_heads=d.num_attention_heads,
                             num_key_value_heads=d.num_key_value_heads, rope_theta=d.rope_theta, pad_token_id=cfg.pad_token_id,
                             head_dim=d.head_dim, rms_norm_eps=d.rms_norm_eps)
    mc = SuryaModelConfig(vocab_size=d.vocab_size, vision_encoder=enc, decoder=dec, image_token_id=cfg.image_token_id)
    mc.decoder._attn_implementation = attn
    mc.vision_encoder._attn_implementation = attn
    m = SuryaModel(mc).eval()
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    return m


def golden_rec(name: str, attn: str):
    from transformers import DynamicCache
    cfg = rec_config(name)
    sd = make_rec_weights(cfg, 0)
    ref = build_reference_rec(cfg, sd, attn)
    tiles, seqs = make_prompts(cfg, REC_GRIDS)
    ids, am, pos = left_pad_batch(cfg, seqs)
    grid = torch.tensor([(1, h, w) for h, w in REC_GRIDS])
    logits, bboxes, tokens = [], [], []
    with torch.inference_mode():
        emb = ref.get_image_embeddings(tiles, grid, 4096)
        cache = DynamicCache()
        out = ref(input_ids=ids, image_tiles=tiles, grid_thw=grid, attention_mask=am, position_ids=pos,
                  past_key_values=cache, use_cache=True, logits_to_keep=1, encoder_chunk_size=4096)
        for step in range(REC_STEPS):
            lm, bb = out.lm_logits[:, -1].float(), out.bbox_logits[:, -1].float()
            logits.append(lm.clone()); bboxes.append((bb * cfg.bbox_size).to(torch.long))
            nxt = lm.argmax(-1, keepdim=True)
            tokens.append(nxt[:, 0].clone())
            am = F.pad(am, (0, 1), value=1)
            pos = pos[:, -1:] + 1
            out = ref(input_ids=nxt, attention_mask=am, position_ids=pos, use_cache=True, past_key_values=cache, logits_to_keep=1)
    return {"config": name, "attn": attn, "grids": REC_GRIDS, "seed": 5, "image_embeddings": emb,
            "logits": torch.stack(logits), "bbox_ints": torch.stack(bboxes), "tokens": torch.stack(tokens)}


def golden_det(name: str, size: int, n: int):
    from surya.detection.model.config import EfficientViTConfig
    from surya.detection.model.encoderdecoder import EfficientViTForSemanticSegmentation
    from oracle.det_oracle import normalise_pages
    c = det_config(name)
    rc = EfficientViTConfig(widths=c.widths, depths=c.depths, head_dim=c.head_dim,
                            decoder_layer_hidden_size=c.decoder_layer_hidden_size, decoder_hidden_size=c.decoder_hidden_size,
                            num_labels=c.num_labels)
    m = EfficientViTForSemanticSegmentation(rc).eval()
    sd = make_det_weights(c, 0)
    m.load_state_dict(sd, strict=True)
    pages = make_pages(n, size, seed=1234)
    x = normalise_pages(pages)
    from surya.detection.processor import SegformerImageProcessor            # the reference's own rescale + normalise
    rp = SegformerImageProcessor(size={"height": size, "width": size})
    assert all(np.array_equal(rp(pg)["pixel_values"][0], x[i].numpy()) for i, pg in enumerate(pages)), "normalise_pages != reference processor"
    with torch.inference_mode():
        out = m(pixel_values=x)
        up = F.interpolate(out.logits, size=(size, size), mode="bilinear", align_corners=False)   # detection/__init__.py:121-129
    return {"config": name, "size": size, "n": n, "page_seed": 1234, "logits": out.logits.clone(),
            "stage_means": [float(h.mean()) for h in out.hidden_states], "upsampled_sample": up[:, :, ::8, ::8].clone()}


def golden_processor():
    """_process_and_tile of the reference processor on a size that needs no cv2 resize (multiple of 28)."""
    from surya.common.surya.processor import SuryaOCRProcessor
    p = object.__new__(SuryaOCRProcessor)
    p.patch_size, p.merge_size = 14, 2
    p.image_mean = np.array(SuryaOCRProcessor.image_mean, dtype=np.float32)
    p.image_std = np.array(SuryaOCRProcessor.image_std, dtype=np.float32)
    img = np.random.default_rng(7).integers(0, 256, size=(56, 84, 3)).astype(np.float32)
    tiles, grid = p._process_and_tile(img)
    return {"image": torch.from_numpy(img), "tiles": tiles.clone(), "grid_thw": tuple(int(g) for g in grid)}


def main():
    os.makedirs(GOLD, exist_ok=True)
    torch.save(golden_rec("REC-TINY", "eager"), os.path.join(GOLD, "rec_tiny_eager.pt"))
    torch.save(golden_rec("REC-TINY", "sdpa"), os.path.join(GOLD, "rec_tiny_sdpa.pt"))
    small = golden_rec("REC-SMALL", "eager")        # FULL's op mix (encoder head_dim 80, GQA 5:1, odd intermediate) at ~1/30 the weights
    lg = small.pop("logits")                         # [steps, B, V]: keep the top 32 per row + logsumexp (fixture size)
    tk = torch.topk(lg, 32, dim=-1)
    small["logits_top"] = {"values": tk.values.clone(), "indices": tk.indices.clone()}
    small["logits_lse"] = torch.logsumexp(lg, -1)
    small["logits_absmax"] = lg.abs().amax(-1)
    torch.save(small, os.path.join(GOLD, "rec_small_eager.pt"))
    torch.save(golden_det("DET-TINY", 128, 2), os.path.join(GOLD, "det_tiny.pt"))
    torch.save(golden_processor(), os.path.join(GOLD, "processor_tiles.pt"))
    for f in sorted(os.listdir(GOLD)):
        print(f, os.path.getsize(os.path.join(GOLD, f)))


if __name__ == "__main__":
    main()
