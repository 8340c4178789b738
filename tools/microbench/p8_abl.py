"""Where does a 256x256 8-phase tile spend its time outside the K loop? Times surya_op_gemm for the library named by SURYA_AMD_LIB
(ablation builds: -DSA_ABL=1 epilogue without its global stores, -DSA_ABL=2 no epilogue at all; results of those are garbage)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from surya_amd import _lib as L
lib = L.lib()
shapes = [(46460, 6912, 1280, 3, "enc gate|up"), (46460, 3840, 1280, 0, "enc qkv"), (46460, 1280, 1280, 1, "enc proj"),
          (46460, 1280, 3456, 1, "enc down"), (8192, 8192, 8192, 0, "square 8k"), (65536, 2048, 128, 0, "K=128 (2 K-tiles)")]
tag = os.environ.get("TAG", "")
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
    ts = []
    for rnd in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    ms = sorted(ts)[2]
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    rounds = (tiles + 255) // 256
    print(f"{tag:10s} {name:20s} M={M:6d} N={N:6d} K={K:5d}: {ms*1e3:9.1f} us  {2.0*M*N*K/ms/1e9:8.1f} TFLOP/s   {tiles} tiles = {rounds} rounds, {ms*1e3/rounds:6.2f} us per round", flush=True)
