"""CPU, build container only: live cross-check of the oracle against the real reference modules imported from
/root/reference through oracle/ref_shim (skipped where the reference is absent, e.g. on the GPU box)."""
import pytest
import torch

from oracle import ref_shim

pytestmark = pytest.mark.skipif(not ref_shim.available(), reason="/root/reference not present")


def test_live_rec_small_head_dim_80_and_gqa():
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
    from oracle.make_golden import build_reference_rec
    from oracle import rec_oracle as ro
    from surya_amd.config import rec_config
    from surya_amd.synth import make_rec_weights
    from util import make_prompts, left_pad_batch
    from transformers import DynamicCache
    cfg = rec_config("REC-SMALL")
    sd = make_rec_weights(cfg, 0)
    ref = build_reference_rec(cfg, sd, "sdpa")
    grids = [(6, 38), (10, 18)]
    tiles, seqs = make_prompts(cfg, grids)
    ids, am, pos = left_pad_batch(cfg, seqs)
    with torch.inference_mode():
        out = ref(input_ids=ids, image_tiles=tiles, grid_thw=torch.tensor([(1, h, w) for h, w in grids]), attention_mask=am,
                  position_ids=pos, past_key_values=DynamicCache(), use_cache=True, logits_to_keep=1, encoder_chunk_size=4096)
    om = ro.OracleRecModel(cfg, sd, cfg.image_token_id)
    lm, bb = om.prefill(ids, tiles, [(1, h, w) for h, w in grids], am, pos)
    assert (lm - out.lm_logits).abs().max().item() <= 1e-4 * out.lm_logits.abs().max().item()
    assert (bb - out.bbox_logits).abs().max().item() <= 1e-5
    assert torch.equal(lm.argmax(-1), out.lm_logits.argmax(-1))
