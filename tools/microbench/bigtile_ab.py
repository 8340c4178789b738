"""A/B of the large-GEMM tile variants (sa::Tuning bigtile) on the encoder / prefill shapes through surya_op_gemm: bit equality of
the outputs (same K order, same MFMA -> identical expected) and event-timed TFLOP/s.

    python tools/microbench/bigtile_ab.py [variants...]      (default: 0 1 = 128x128 only vs 256x256 where whole rounds allow)
    python tools/microbench/bigtile_ab.py persist            (round 4: one 256x256 tile per workgroup vs the persistent tile loop,
                                                              interleaved rounds: 5 x (10 launches each), median reported)
"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from surya_amd import _lib as L

lib = L.lib()
KNOB = b"bigtile"
if len(sys.argv) > 1 and sys.argv[1] == "persist":
    KNOB, variants = b"persist", [0, 1]
else:
    L.check(lib.surya_set_tuning(b"persist", C.c_int(0)), "tuning")          # the bigtile variants are compared one tile per workgroup
    variants = [int(v) for v in sys.argv[1:]] or [0, 1]
shapes = [(46460, 6912, 1280, 3, "enc gate|up"), (46460, 3840, 1280, 0, "enc qkv"), (46460, 1280, 1280, 1, "enc proj"),
          (46460, 1280, 3456, 1, "enc down"), (15360, 10240, 1280, 3, "dec prefill gate|up"), (15360, 1280, 5120, 1, "dec prefill down"),
          (15360, 1792, 1280, 0, "dec prefill qkv"), (8192, 8192, 8192, 0, "square 8k"), (4096, 4096, 4096, 0, "square 4k")]
for M, N, K, epi, name in shapes:
    torch.manual_seed(0)
    x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    b = torch.randn(N, device="cuda").bfloat16()
    No = N // 2 if epi == 3 else N
    r = torch.randn(M, No, device="cuda").bfloat16() if epi == 1 else None
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ref = None
    c = torch.empty(M, No, device="cuda", dtype=torch.bfloat16)
    def run():
        rc = lib.surya_op_gemm(1, 0, epi, L.ptr(x), C.c_long(K), L.ptr(w), C.c_long(K), L.ptr(c), C.c_long(No), L.ptr(b), L.ptr(r),
                               C.c_long(No), M, N, K, st)
        assert rc == 0, rc
    times = {v: [] for v in variants}
    same = {}
    for rnd in range(5):                                   # interleaved rounds: a clock / thermal drift hits every variant alike
        for v in variants:
            L.check(lib.surya_set_tuning(KNOB, C.c_int(v)), "tuning")
            for _ in range(2): run()
            torch.cuda.synchronize()
            if ref is None: ref = c.clone()
            same[v] = bool((c.view(torch.int16) == ref.view(torch.int16)).all())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 10
            e0.record()
            for _ in range(n): run()
            e1.record(); torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / n)
    for v in variants:
        ms = sorted(times[v])[len(times[v]) // 2]
        print(f"{name:22s} M={M:6d} N={N:6d} K={K:5d} {KNOB.decode()}={v}: {ms*1e3:9.1f} us  {2.0*M*N*K/ms/1e9:8.1f} TFLOP/s  (min {min(times[v])*1e3:.1f})  identical={same[v]}", flush=True)
L.check(lib.surya_set_tuning(KNOB, C.c_int(1 if KNOB == b"persist" else 3)), "tuning")      # back to the defaults
