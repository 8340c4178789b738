"""usage: det_conv_ab.py [knob]   (default conv_persist)
Detector forward with the 3 x 3 convolutions on the one-tile 2-stage kernel (sa::Tuning conv_persist = 0) vs on the persistent 8-phase
loop with the gather in its request stream (conv_persist = 1): heat maps must be BIT-identical (same K order, same MFMA; the virtual zero
K-tile of an odd K-tile count adds + 0 * 0), then interleaved timing of the whole forward."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from surya_amd import _lib as L
from surya_amd.config import det_config
from surya_amd.detection.model import HipDetModel
from surya_amd.synth import make_det_weights, make_pages
from oracle import det_oracle as do          # input normalisation only (host logic)

KNOB = (sys.argv[1] if len(sys.argv) > 1 else "conv_persist").encode()        # or dwconv_pipe, conv_lean, ...
VALUES = tuple(int(v) for v in sys.argv[2].split(",")) if len(sys.argv) > 2 else (0, 1)
lib = L.lib()
cfg = det_config("DET-DEFAULT")
sd = make_det_weights(cfg, 0)
for pages_n, size in ((16, 1024), (3, 640)):
    m = HipDetModel(cfg, sd, height=size, width=size, dtype=torch.bfloat16, device="cuda:0", max_batch=pages_n)
    x = do.normalise_pages(make_pages(pages_n, size, seed=99)).cuda().contiguous()
    outs = {}
    for v in VALUES:
        L.check(lib.surya_set_tuning(KNOB, C.c_int(v)), "tuning")
        for rep in range(3):
            h = m.forward(x).clone()
            torch.cuda.synchronize()
            if v in outs:
                assert torch.equal(outs[v].view(torch.int32), h.view(torch.int32)), "not run-to-run identical"
            outs[v] = h
    same = all(torch.equal(outs[VALUES[0]].view(torch.int32), outs[v].view(torch.int32)) for v in VALUES[1:])
    print(f"{pages_n} pages of {size}^2: heat maps bit-identical across {KNOB.decode()} {VALUES}: {same}   (max |diff| {max(float((outs[VALUES[0]] - outs[v]).abs().max()) for v in VALUES[1:]):.3e})", flush=True)
    times = {v: [] for v in VALUES}
    for rnd in range(5):
        for v in VALUES:
            L.check(lib.surya_set_tuning(KNOB, C.c_int(v)), "tuning")
            m.forward(x); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5): m.forward(x)
            torch.cuda.synchronize()
            times[v].append((time.perf_counter() - t0) / 5 * 1e3)
    for v in VALUES:
        ms = sorted(times[v])[2]
        print(f"    {KNOB.decode()}={v}: {ms:.3f} ms per forward, {pages_n / ms * 1e3:.1f} pages/s", flush=True)
    del m
