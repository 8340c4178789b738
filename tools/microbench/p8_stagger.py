"""Experiment: do the 256 workgroups of a round lose time because they all store their 128 KiB tiles at the same moment? sa::Tuning
p8_stagger delays the first-round workgroups by (blockIdx / 8 % 4) x N cycles, which leaves the CUs a quarter tile apart for the rest
of the launch. Per-shape time at stagger 0 / a quarter of the tile time."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from surya_amd import _lib as L
lib = L.lib()
shapes = [(46460, 6912, 1280, 3, "enc gate|up"), (46460, 3840, 1280, 0, "enc qkv"), (46460, 1280, 1280, 1, "enc proj"),
          (46460, 1280, 3456, 1, "enc down"), (15360, 10240, 1280, 3, "dec prefill gate|up")]
for M, N, K, epi, name in shapes:
    x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    b = torch.randn(N, device="cuda").bfloat16()
    No = N // 2 if epi == 3 else N
    c = torch.empty(M, No, device="cuda", dtype=torch.bfloat16)
    r = torch.randn(M, No, device="cuda").bfloat16() if epi == 1 else None
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    def run():
        rc = lib.surya_op_gemm(1, 0, epi, L.ptr(x), C.c_long(K), L.ptr(w), C.c_long(K), L.ptr(c), C.c_long(No), L.ptr(b), L.ptr(r),
                               C.c_long(No), M, N, K, st)
        assert rc == 0
    res = {}
    variants = [0, 5000, 10000, 20000]
    for rnd in range(5):
        for v in variants:
            L.check(lib.surya_set_tuning(b"p8_stagger", C.c_int(v)), "tuning")
            for _ in range(2): run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(10): run()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(v, []).append(e0.elapsed_time(e1) / 10)
    print(f"{name:22s} M={M:6d} N={N:6d} K={K:5d}: " + "  ".join(f"stagger {v}: {sorted(t)[2]*1e3:7.1f} us" for v, t in res.items()), flush=True)
L.check(lib.surya_set_tuning(b"p8_stagger", C.c_int(0)), "tuning")
