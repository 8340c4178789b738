import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from test_gpu_rec import GRIDS, build
from util import make_prompts
cfg, sd, m = build("REC-SMALL", torch.bfloat16)
tiles, seqs = make_prompts(cfg, GRIDS)
slots = list(range(len(seqs)))
def pre():
    m.prefill(tiles.cuda(), GRIDS, seqs, slots)
    t, _, _ = m.read_outputs(1)
    lg = m.last_logits().cpu().clone()
    return t[0][:len(slots)].copy(), lg
def full():
    m.prefill(tiles.cuda(), GRIDS, seqs, slots); m.set_active(slots); m.decode(6)
    t, s, b = m.read_outputs(6)
    return t[:, :len(slots)].copy()
t0, l0 = pre(); t1, l1 = pre()
print("bf16 prefill twice: tokens equal", np.array_equal(t0, t1), "logits bitwise equal", bool((l0.view(torch.int32) == l1.view(torch.int32)).all()))
m.set_kv_fp8(True)
t2, l2 = pre()
print("kv8 prefill vs bf16: tokens equal", np.array_equal(t0, t2), "logits bitwise equal", bool((l0.view(torch.int32) == l2.view(torch.int32)).all()), "max diff", float((l0 - l2).abs().max()))
f = full()
m.set_kv_fp8(False)
a = full()
print("prefill tokens", t0.tolist()); print("a[0]", a[0].tolist()); print("f[0]", f[0].tolist())
top = torch.topk(l0, 2, dim=-1)
print("prefill top2 margins", (top.values[:, 0] - top.values[:, 1]).tolist(), top.indices.tolist())
