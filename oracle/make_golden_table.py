"""Record table-recognition fixtures from the REAL reference modules (build container only).

    python oracle/make_golden_table.py

Imports VikParuchuri/surya @ v0.14.6's table_rec DonutSwinModel / SuryaTableRecDecoder from /root/reference through oracle/ref_shim,
loads the seeded synthetic weights (surya_amd.synth.make_table_weights) into them and records, per configuration (TABLE-TINY,
TABLE-SMALL, TABLE-DEFAULT): the encoder output (a strided sample for the large one) and the reference's own inference loop
(surya/table_rec/__init__.py:35-131) on a 3-token prompt [bos, query, query_end] + a few column tokens: the multi-token prefill
(prefill=True, cache_position 0..T-1), then N single-token steps fed back through LabelShaper.dict_to_labels exactly as the loop
does; per step the property logits of the last position and the fed-back tokens.
-> tests/golden/table_{tiny,small,default}.pt. tests/test_oracle_golden.py pins oracle/layout_oracle.py to them on the CPU,
tests/test_gpu_table.py the HIP path on the GPU box (where /root/reference does not exist)."""
from __future__ import annotations

import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch

from oracle import ref_shim

GOLD = os.path.join(ROOT, "tests", "golden")


def build_reference_table(cfg, sd):
    cfgm, encm, decm, _ = ref_shim.import_table_modules()
    e, d = cfg.encoder, cfg.decoder
    enc_cfg = cfgm.DonutSwinTableRecConfig(image_size=e.image_size, embed_dim=e.embed_dim, depths=list(e.depths), num_heads=list(e.num_heads),
                                           num_kv_heads=list(e.num_kv_heads), window_size=e.window_size, encoder_length=e.encoder_length,
                                           layer_norm_eps=e.layer_norm_eps)
    L = tuple(range(d.num_hidden_layers))
    dec_cfg = cfgm.SuryaTableRecDecoderConfig(num_hidden_layers=d.num_hidden_layers, hidden_size=d.hidden_size,
                                              property_embed_size=d.property_embed_size, box_embed_size=d.box_embed_size,
                                              intermediate_size=d.intermediate_size, encoder_hidden_size=d.encoder_hidden_size,
                                              num_attention_heads=d.num_attention_heads, num_key_value_heads=d.num_key_value_heads,
                                              cross_attn_layers=L, encoder_cross_attn_layers=L, self_attn_layers=L, global_attn_layers=L,
                                              rms_norm_eps=d.rms_norm_eps, layer_norm_eps=d.layer_norm_eps, rope_theta=d.rope_theta)
    enc = encm.DonutSwinModel(enc_cfg).eval()
    dec = decm.SuryaTableRecDecoder(dec_cfg).eval()
    m1 = enc.load_state_dict({k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}, strict=False)
    m2 = dec.load_state_dict({k[len("decoder."):]: v for k, v in sd.items() if k.startswith("decoder.")}, strict=False)
    assert not [k for k in m1.missing_keys if "relative_position_index" not in k] and not m2.missing_keys, (m1, m2)
    assert not m1.unexpected_keys and not m2.unexpected_keys
    return enc, dec, dec_cfg


def table_pixels(cfg, batch: int, seed: int) -> torch.Tensor:
    return torch.randn(batch, 3, *cfg.encoder.image_size, generator=torch.Generator().manual_seed(seed))


def table_prompt(cfg, batch: int, n_columns: int, seed: int) -> torch.Tensor:
    """[B, 3 + n_columns, 10] decoder prompt as SuryaTableRecProcessor builds it (table_rec/processor.py:62-85): bos row, the query
    (a table / row box with its properties), query_end row, then column boxes -- the same column list for every row of the batch."""
    from surya_amd.table_rec.config import SPECIAL_TOKENS
    d = cfg.decoder
    g = torch.Generator().manual_seed(seed)
    ri = lambda lo, hi, n=1: torch.randint(lo, hi, (n,), generator=g).tolist()

    def box_token(cat):
        cx, cy = ri(100, 900)[0], ri(100, 900)[0]
        w, h = ri(20, 600)[0], ri(10, 300)[0]
        return [cx, cy, w, h, 512 + ri(-6, 7)[0], 512 + ri(-6, 7)[0], cat + SPECIAL_TOKENS, ri(0, 4)[0] + SPECIAL_TOKENS, ri(0, 4)[0],
                ri(0, 2)[0] + SPECIAL_TOKENS]

    cols = [box_token(2) for _ in range(n_columns)]
    rows = []
    for b in range(batch):
        rows.append([[d.bos_token_id] * 10, box_token(4 if n_columns == 0 else 1), [d.query_end_token_id] * 10] + cols)
    return torch.tensor(rows, dtype=torch.long)


def record(name: str, batch: int, n_columns: int, steps: int, seed: int, enc_stride: int):
    from surya_amd.table_rec.config import table_config, BOX_PROPERTIES, SPECIAL_TOKENS, BOX_DIM
    from surya_amd.synth import make_table_weights
    _, _, _, shaper_mod = ref_shim.import_table_modules()
    shaper = shaper_mod.LabelShaper()
    cfg = table_config(name)
    d = cfg.decoder
    sd = make_table_weights(cfg, 0)
    enc, dec, dec_cfg = build_reference_table(cfg, sd)
    x = table_pixels(cfg, batch, seed)
    ids = table_prompt(cfg, batch, n_columns, seed + 100)
    t0 = time.time()
    with torch.inference_mode():
        h = enc(pixel_values=x).last_hidden_state
        dec.model._setup_cache(dec_cfg, batch, "cpu", torch.float32)
        pos = torch.ones_like(ids[0, :, 0], dtype=torch.int64).cumsum(0) - 1
        cur = ids
        logs, fed = [], []
        for step in range(steps):
            out = dec(input_ids=cur, encoder_hidden_states=h, cache_position=pos, use_cache=True, prefill=(step == 0))
            pos = pos[-1:] + 1
            last = {k: out["box_property_logits"][k][:, -1, :].clone() for k, _, _ in BOX_PROPERTIES}
            logs.append(last)
            # the loop's post-processing (:78-117)
            props = []
            for j in range(batch):
                bp = {}
                for k, _, mode in BOX_PROPERTIES:
                    if mode == "classification":
                        bp[k] = int(last[k][j].argmax(-1)) - SPECIAL_TOKENS
                    elif k == "bbox":
                        bp[k] = (last[k][j] * BOX_DIM).tolist()
                    else:
                        bp[k] = int(torch.round(torch.clamp(last[k][j], min=1)).item())
                props.append(bp)
            cur = torch.tensor(shaper.dict_to_labels(props), dtype=torch.long).unsqueeze(1)
            fed.append(cur[:, 0].clone())
    print(f"{name}: reference encoder + prefill of {ids.shape[1]} tokens + {steps - 1} steps, batch {batch}: {time.time() - t0:.1f}s; "
          f"categories of image 0: {[int(l['category'][0].argmax()) for l in logs]}", flush=True)
    g = {"config": name, "batch": batch, "seed": seed, "steps": steps, "enc_stride": enc_stride, "prompt": ids,
         "encoder_out": h[:, ::enc_stride].clone(), "encoder_absmax": float(h.abs().max()),
         "logits": {k: torch.stack([l[k] for l in logs]) for k, _, _ in BOX_PROPERTIES}, "fed_tokens": torch.stack(fed)}
    torch.save(g, os.path.join(GOLD, "table_" + name.split("-")[1].lower() + ".pt"))


def main():
    ref_shim.install_layout()
    record("TABLE-TINY", 3, 2, 10, 21, 1)
    record("TABLE-SMALL", 4, 3, 12, 22, 1)
    record("TABLE-DEFAULT", 2, 4, 8, 23, 8)
    for f in sorted(os.listdir(GOLD)):
        if f.startswith("table_"):
            print(f, os.path.getsize(os.path.join(GOLD, f)))


if __name__ == "__main__":
    main()
