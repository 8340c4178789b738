"""Does the row stride of the activation operand matter for the decode-regime GEMM (L2 channel hot-spotting when 640 workgroups read
the same 256 x 1280 X through rows 2560 bytes apart)? surya_op_gemm on the gate|up shape with X stored at several row strides.
    python tools/microbench/ldx_ab.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from surya_amd import _lib as L
lib = L.lib()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (M, N, K, epi, name) in ((256, 10240, 1280, 3, "gate|up"), (256, 81920, 1280, 0, "lm_head-like (bias epilogue)")):
    torch.manual_seed(0)
    nw = 8 if N < 50000 else 2
    ws = [(torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16() for _ in range(nw)]
    No = N // 2 if epi == 3 else N
    c = torch.zeros(M, No, device="cuda", dtype=torch.bfloat16)
    x0 = torch.randn(M, K, device="cuda").bfloat16()
    ref = None
    for ldx in (K, K + 64, K + 32, K + 128, K + 192, 2048):
        xb = torch.zeros(M, ldx, device="cuda", dtype=torch.bfloat16)
        xb[:, :K] = x0
        def run(w):
            rc = lib.surya_op_gemm(1, 0, epi, L.ptr(xb), C.c_long(ldx), L.ptr(w), C.c_long(K), L.ptr(c), C.c_long(No), L.ptr(None), L.ptr(None),
                                   C.c_long(No), M, N, K, st)
            assert rc == 0, rc
        run(ws[0]); torch.cuda.synchronize()
        if ref is None: ref = c.clone()
        same = bool((c.view(torch.int16) == ref.view(torch.int16)).all())
        for i in range(4): run(ws[i % nw])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for i in range(64): run(ws[i % nw])
        e1.record(); torch.cuda.synchronize()
        print(f"{name} M={M} ldx={ldx:5d} ({ldx * 2} B rows): {e0.elapsed_time(e1) / 64 * 1e3:7.2f} us/launch  identical={same}", flush=True)
