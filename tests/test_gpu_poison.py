"""GPU: the engines must not depend on what hipMalloc hands out. With SURYA_AMD_POISON=1 every arena is filled with 0xFF bytes (NaN as
bf16 / fp32) right after its allocation (csrc/common.h poison_arena); an engine that reads memory it never wrote then produces NaN on
every box instead of on the one whose last tenant left NaN patterns in HBM (gpurun r04f: the layout decoder's first step read cache
row 0 behind its masked key columns -- 0 x NaN). The flag is read once per process, hence the subprocess."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import sys
sys.path.insert(0, %r)
sys.path.insert(0, %r + "/tests")
import numpy as np, torch
import __graft_entry__ as ge
from surya_amd.layout.config import layout_config
from surya_amd.layout.model import HipLayoutModel
from surya_amd.synth import make_layout_weights, make_table_weights
from surya_amd.table_rec.config import table_config
ge.smoke()                                                       # recognition: prefill + 7 decode steps == the oracle's tokens
for fam, cfg, sd in (("layout", layout_config("LAYOUT-TINY"), None), ("table", table_config("TABLE-TINY"), None)):
    sd = make_layout_weights(cfg, 0) if fam == "layout" else make_table_weights(cfg, 0)
    for dtype in (torch.float32, torch.bfloat16):
        m = HipLayoutModel(cfg, sd, dtype=dtype, max_batch=3, max_boxes=16)
        px = torch.randn(3, 3, *cfg.encoder.image_size, generator=torch.Generator().manual_seed(1)).cuda().contiguous()
        m.encode(px)
        assert torch.isfinite(m.encoder_states().float()).all(), (fam, dtype, "encoder")
        tok = np.full((3, m.tok_width), cfg.decoder.bos_token_id, np.int32)
        for k in range(4):
            cls, box = m.decode_step(tok, k)
            assert np.isfinite(cls).all() and np.isfinite(box).all(), (fam, dtype, "decode step", k)
        m.encode(px)
        m.set_feedback(None if fam == "table" else np.array([[612, 792]] * 3, np.int32))
        m.decode_steps(tok, 0, 6, 0)
        cls, box, fed = m.wait_steps(6, 0)
        assert np.isfinite(cls).all() and np.isfinite(box).all(), (fam, dtype, "device-fed run")
print("poison ok")
"""


def test_engines_do_not_read_unwritten_arena_memory(hip_lib):
    env = dict(os.environ, SURYA_AMD_POISON="1")
    r = subprocess.run([sys.executable, "-c", SCRIPT % (ROOT, ROOT)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "poison ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
