"""Time surya_op_gemm on the shapes of the recognition path (hipEvents on torch's current stream)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from surya_amd import _lib as L
lib = L.lib()
shapes = [(46460, 6912, 1280, 3, "enc gate|up"), (46460, 3840, 1280, 0, "enc qkv"), (46460, 1280, 1280, 1, "enc proj"),
          (46460, 1280, 3456, 1, "enc down"), (15360, 10240, 1280, 3, "dec prefill gate|up"), (15360, 1280, 5120, 1, "dec prefill down"),
          (8192, 8192, 8192, 0, "square 8k"), (4096, 4096, 4096, 0, "square 4k"), (256, 10240, 1280, 3, "decode gate|up"),
          (256, 81920, 1280, 0, "lm_head"), (256, 1792, 1280, 0, "decode qkv (unsplit)"), (256, 1280, 1280, 1, "decode o (unsplit)"),
          (256, 1280, 5120, 1, "decode down (unsplit)")]
import os
if os.environ.get("ONLY_DECODE"):
    shapes = [s for s in shapes if s[0] == 256]
if os.environ.get("ONLY_SHAPE"):                    # counter passes: one launch shape per run (tools/profile_bigtile_pmc.sh)
    shapes = [s for s in shapes if s[4] == os.environ["ONLY_SHAPE"]]
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
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    n = 10
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"{name:22s} M={M:6d} N={N:6d} K={K:5d}: {ms*1e3:9.1f} us  {2.0*M*N*K/ms/1e9:8.1f} TFLOP/s  W+X+C {(M*K+N*K+M*No)*2/ms/1e6:8.1f} GB/s")
