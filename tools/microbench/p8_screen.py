"""Race screen of the 8-phase 256x256 bf16 GEMM schedule (sa::Tuning bigtile = 3) against the 2-stage loop (bigtile = 1).

Same K order, same MFMA, same accumulator assignment -> outputs must be BIT-identical. A misplaced counted wait / barrier shows up as
rare wrong tiles that come and go with shape and memory load, so every shape is run `REPS` times against fresh operands, with a
bandwidth hog on a second stream for half of the runs (DMA landing order changes under load).

    python tools/microbench/p8_screen.py [reps] [persist]       (persist: screen the PERSISTENT 8-phase loop, sa::Tuning persist = 1, instead)
"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from surya_amd import _lib as L

lib = L.lib()
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 6
PERSIST = len(sys.argv) > 2 and sys.argv[2] == "persist"
# (M, N, K, epi): small / ragged / one round / several rounds; K-tile counts 2, 4, 6, 20, 54, 64; epilogues bias, residual, gelu, swiglu
shapes = [(512, 512, 128, 0), (512, 512, 256, 1), (300, 264, 384, 0), (1024, 512, 256, 3), (70000, 256, 512, 1), (777, 1280, 1280, 1), (4096, 4096, 4096, 0), (2048, 6912, 1280, 3),
          (46460, 1280, 1280, 1), (15360, 1792, 1280, 0), (8192, 1280, 3456, 2), (46460, 3840, 1280, 0)]
L.check(lib.surya_set_tuning(b"bigtile_any", C.c_int(1)), "tuning")
hog_stream = torch.cuda.Stream()
hog_a = torch.empty(64 << 20, device="cuda", dtype=torch.float32)
hog_b = torch.empty_like(hog_a)
bad = 0
for M, N, K, epi in shapes:
    No = N // 2 if epi == 3 else N
    mism = 0
    for rep in range(REPS):
        torch.manual_seed(rep * 7919 + M)
        x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
        b = torch.randn(N, device="cuda").bfloat16()
        r = torch.randn(M, No, device="cuda").bfloat16() if epi == 1 else None
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        outs = {}
        for v in (1, 3):
            L.check(lib.surya_set_tuning(b"bigtile", C.c_int(v)), "tuning")
            L.check(lib.surya_set_tuning(b"persist", C.c_int(1 if (PERSIST and v == 3) else 0)), "tuning")
            c = torch.full((M, No), float("nan"), device="cuda", dtype=torch.bfloat16)
            torch.cuda.synchronize()
            if rep % 2 == 1:
                with torch.cuda.stream(hog_stream):
                    for _ in range(4): hog_b.copy_(hog_a)
            for _ in range(3 if v == 3 else 1):          # the screened variant three times back to back (warm / cold L2 mix)
                rc = lib.surya_op_gemm(1, 0, epi, L.ptr(x), C.c_long(K), L.ptr(w), C.c_long(K), L.ptr(c), C.c_long(No), L.ptr(b), L.ptr(r),
                                       C.c_long(No), M, N, K, st)
                assert rc == 0, rc
            torch.cuda.synchronize()
            outs[v] = c
        ne = int((outs[1].view(torch.int16) != outs[3].view(torch.int16)).sum())
        if rep == 0:
            ref = (x.float() @ w.float().t())
            # loose sanity of the baseline itself against fp32 (bias / epilogue aside, on the plain-bias shapes)
            if epi == 0:
                err = float(((outs[1].float() - (ref + b.float())).abs().max()))
                assert err < 0.25, err
        mism += ne
    bad += mism
    print(f"M={M:6d} N={N:5d} K={K:5d} epi={epi}: {REPS} runs, mismatching elements {mism}", flush=True)
L.check(lib.surya_set_tuning(b"bigtile_any", C.c_int(0)), "tuning")
L.check(lib.surya_set_tuning(b"bigtile", C.c_int(3)), "tuning")
L.check(lib.surya_set_tuning(b"persist", C.c_int(1)), "tuning")
print("RACE SCREEN (persistent loop)" if PERSIST else "RACE SCREEN", "CLEAN" if bad == 0 else f"FAILED ({bad} elements)")
sys.exit(0 if bad == 0 else 1)
