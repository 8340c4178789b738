"""Debug: determinism / NaN check of the layout decode steps (gpurun r04g)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import numpy as np, torch
from surya_amd import _lib as L
from surya_amd.layout.config import layout_config
from surya_amd.layout.model import HipLayoutModel
from surya_amd.synth import make_layout_weights

if len(sys.argv) > 1:
    L.check(L.lib().surya_set_tuning(b"graph", int(sys.argv[1])), "tuning")
for name, dtype, B, mb in (("LAYOUT-DEFAULT", torch.bfloat16, 2, 16), ("LAYOUT-DEFAULT", torch.bfloat16, 32, 104), ("LAYOUT-SMALL", torch.float32, 4, 32)):
    cfg = layout_config(name)
    d = cfg.decoder
    sd = make_layout_weights(cfg, 0)
    m = HipLayoutModel(cfg, sd, dtype=dtype, max_batch=B, max_boxes=mb)
    px = torch.randn(B, 3, *cfg.encoder.image_size, generator=torch.Generator().manual_seed(5)).cuda().contiguous()
    ref = None
    for rep in range(3):
        m.encode(px)
        boxes = np.full((B, 7), d.bos_token_id, np.int32)
        rec = []
        for k in range(10):
            cls, box = m.decode_step(boxes, k)
            rec.append((cls, box))
            boxes = np.concatenate([box * d.bbox_size, cls.argmax(-1)[:, None].astype(np.float32)], -1).astype(np.int64).astype(np.int32)
        nan = [int(np.isnan(c).sum() + np.isnan(b).sum()) for c, b in rec]
        if ref is None:
            ref = rec
        same = [bool(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])) for a, b in zip(ref, rec)]
        print(name, dtype, B, "host-fed rep", rep, "nan per step", nan, "same as rep 0", same, flush=True)
    for rep in range(3):
        m.encode(px)
        m.set_feedback(np.tile(np.array([[612, 792]], np.int32), (B, 1)))
        boxes = np.full((B, 7), d.bos_token_id, np.int32)
        m.decode_steps(boxes, 0, 10, 0)
        cls, box, tok = m.wait_steps(10, 0)
        nan = [int(np.isnan(cls[k]).sum() + np.isnan(box[k]).sum()) for k in range(10)]
        same = [bool(np.array_equal(ref[k][0], cls[k]) and np.array_equal(ref[k][1], box[k])) for k in range(10)]
        print(name, dtype, B, "device-fed rep", rep, "nan per step", nan, "same as host-fed", same, flush=True)
    del m
