"""usage: det_lib_ab.py <libA.so> <libB.so> [rounds]
A/B of two BUILDS of the library on the detector forward (DET-DEFAULT bf16, 16 pages of 1024^2): one subprocess per (round, build),
alternating, each loading its build through SURYA_AMD_LIB; prints the median forward time and a CRC of the heat maps per run.
(For knobs inside one build use det_conv_ab.py; this is for changes that are compile-time, e.g. -DSA_CONV_TRUE_DIV=1.)"""
import os, subprocess, sys

CHILD = r'''
import os, sys, time, zlib
sys.path.insert(0, os.environ["SA_ROOT"])
import torch
from surya_amd.config import det_config
from surya_amd.detection.model import HipDetModel
from surya_amd.synth import make_det_weights, make_pages
from oracle import det_oracle as do
cfg = det_config("DET-DEFAULT")
m = HipDetModel(cfg, make_det_weights(cfg, 0), height=1024, width=1024, dtype=torch.bfloat16, device="cuda:0", max_batch=16)
x = do.normalise_pages(make_pages(16, 1024, seed=99)).cuda().contiguous()
for _ in range(3): h = m.forward(x)
torch.cuda.synchronize()
crc = zlib.crc32(h.cpu().numpy().tobytes())
ts = []
for _ in range(7):
    t0 = time.perf_counter()
    for _ in range(5): m.forward(x)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / 5 * 1e3)
print(f"{sorted(ts)[3]:.3f} {crc:08x}")
'''
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
libs = [os.path.abspath(sys.argv[1]), os.path.abspath(sys.argv[2])]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ, SURYA_AMD_LIB=l, SA_ROOT=root)
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
        line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else "FAILED " + out.stderr[-300:]
        print(f"round {r} {os.path.basename(l)}: {line}", flush=True)
        res[l].append(line)
