"""GPU: op-level kernels through the C ABI vs a plain PyTorch fp32 reference of the same op."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from surya_amd import _lib as L

pytestmark = pytest.mark.gpu


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def run_gemm(lib, x, w, bias, epi, res=None, out_f32=False):
    dt = L.DTYPE_F32 if x.dtype == torch.float32 else L.DTYPE_BF16
    M, K = x.shape
    N = w.shape[0]
    No = N // 2 if epi == L.EPI_SWIGLU else N
    odt = torch.float32 if (out_f32 or x.dtype == torch.float32) else torch.bfloat16
    c = torch.full((M, No), float("nan"), dtype=odt, device=x.device)
    rc = lib.surya_op_gemm(dt, int(out_f32), epi, L.ptr(x), C.c_long(x.stride(0)), L.ptr(w), C.c_long(w.stride(0)), L.ptr(c),
                           C.c_long(No), L.ptr(bias), L.ptr(res), C.c_long(No if res is not None else 0), M, N, K, _stream())
    assert rc == 0, rc
    torch.cuda.synchronize()
    return c


def ref_gemm(x, w, bias, epi, res=None):
    y = x.float() @ w.float().t()
    if bias is not None:
        y = y + bias.float()
    if epi == L.EPI_RESIDUAL:
        y = y + res.float()
    elif epi == L.EPI_GELU:
        y = F.gelu(y)
    elif epi == L.EPI_SWIGLU:
        y = F.silu(y[:, 0::2]) * y[:, 1::2]
    elif epi == L.EPI_HARDSWISH:
        y = F.hardswish(y)
    elif epi == L.EPI_RELU:
        y = F.relu(y)
    return y


SHAPES = [(128, 128, 64), (200, 136, 128), (1000, 1280, 640), (777, 3840, 1280), (3, 1792, 1280), (37, 64, 256),
          (256, 6912, 1280), (4100, 1280, 3456), (65, 1024, 128)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", SHAPES)
def test_gemm_bias(hip_lib, dtype, M, N, K):
    g = torch.Generator(device="cuda").manual_seed(M * 7 + N)
    x = torch.randn(M, K, device="cuda", generator=g).to(dtype)
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(dtype)
    b = torch.randn(N, device="cuda", generator=g).to(dtype)
    out = run_gemm(hip_lib, x, w, b, L.EPI_BIAS)
    ref = ref_gemm(x, w, b, L.EPI_BIAS)
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    err = (out.float() - ref).abs().max().item()
    assert err <= tol * max(1.0, ref.abs().max().item()), f"max err {err}"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("epi", [L.EPI_RESIDUAL, L.EPI_GELU, L.EPI_SWIGLU, L.EPI_HARDSWISH, L.EPI_RELU])
def test_gemm_epilogues(hip_lib, dtype, epi):
    M, N, K = 333, 512, 320
    g = torch.Generator(device="cuda").manual_seed(epi)
    x = torch.randn(M, K, device="cuda", generator=g).to(dtype)
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(dtype)
    b = torch.randn(N, device="cuda", generator=g).to(dtype)
    res = torch.randn(M, N, device="cuda", generator=g).to(dtype) if epi == L.EPI_RESIDUAL else None
    out = run_gemm(hip_lib, x, w, b, epi, res)
    ref = ref_gemm(x, w, b, epi, res)
    tol = 3e-5 if dtype == torch.float32 else 3e-2
    err = (out.float() - ref).abs().max().item()
    assert not torch.isnan(out.float()).any()
    assert err <= tol * max(1.0, ref.abs().max().item()), f"max err {err}"


@pytest.mark.parametrize("epi", [L.EPI_BIAS, L.EPI_RESIDUAL, L.EPI_SWIGLU])
def test_gemm_256x256_tiles_ragged(hip_lib, epi):
    """Shape that the launcher sends to the 256x256 tile (16 x 16 tiles = one full round of 256 workgroups beats two rounds of
    128x128 tiles), ragged in M and N so edge tiles clip rows and columns."""
    M, N, K = 3990, 4000, 320
    g = torch.Generator(device="cuda").manual_seed(100 + epi)
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
    b = torch.randn(N, device="cuda", generator=g).bfloat16()
    No = N // 2 if epi == L.EPI_SWIGLU else N
    res = torch.randn(M, No, device="cuda", generator=g).bfloat16() if epi == L.EPI_RESIDUAL else None
    out = run_gemm(hip_lib, x, w, b, epi, res)
    ref = ref_gemm(x, w, b, epi, res)
    err = (out.float() - ref).abs().max().item()
    assert not torch.isnan(out.float()).any()
    assert err <= 3e-2 * max(1.0, ref.abs().max().item()), f"max err {err}"


def test_gemm_transpose_detecting(hip_lib):
    """A = I against an asymmetric W: catches a swapped C layout (guide rule 16)."""
    K = 64
    x = torch.eye(K, device="cuda")[:48].contiguous()
    w = (torch.arange(80 * K, device="cuda", dtype=torch.float32).reshape(80, K) % 97) / 97.0
    out = run_gemm(hip_lib, x, w, None, L.EPI_BIAS)
    assert torch.equal(out, w[:, :48].t().contiguous())


def test_gemm_strided_views_and_f32_out(hip_lib):
    g = torch.Generator(device="cuda").manual_seed(1)
    big = torch.randn(300, 3 * 256, device="cuda", generator=g).bfloat16()
    x = big[:, 256:512]                                  # ldx = 768, base offset 256 elements
    w = (torch.randn(1024, 256, device="cuda", generator=g) / 16).bfloat16()
    out = run_gemm(hip_lib, x, w, None, L.EPI_BIAS, out_f32=True)
    ref = x.float() @ w.float().t()
    assert out.dtype == torch.float32
    assert (out - ref).abs().max().item() < 1e-3


def test_gemm_rejects_bad_k(hip_lib):
    x = torch.zeros(4, 48, device="cuda"); w = torch.zeros(8, 48, device="cuda"); c = torch.zeros(4, 8, device="cuda")
    rc = hip_lib.surya_op_gemm(0, 0, 0, L.ptr(x), C.c_long(48), L.ptr(w), C.c_long(48), L.ptr(c), C.c_long(8), None, None,
                               C.c_long(0), 4, 8, 48, _stream())
    assert rc == -2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,C_", [(1, 128), (257, 1280), (1000, 320), (5, 5120)])
def test_rmsnorm(hip_lib, dtype, rows, C_):
    g = torch.Generator(device="cuda").manual_seed(rows)
    x = (torch.randn(rows, C_, device="cuda", generator=g) * 3).to(dtype)
    w = (1 + 0.1 * torch.randn(C_, device="cuda", generator=g)).to(dtype)
    y = torch.empty_like(x)
    dt = L.DTYPE_F32 if dtype == torch.float32 else L.DTYPE_BF16
    rc = hip_lib.surya_op_rmsnorm(dt, L.ptr(x), C.c_long(C_), L.ptr(w), L.ptr(y), C.c_long(C_), rows, C_, C.c_float(1e-6), _stream())
    assert rc == 0
    torch.cuda.synchronize()
    xf = x.float()
    ref = w * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6)).to(dtype)   # Qwen2RMSNorm semantics
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert (y.float() - ref.float()).abs().max().item() <= tol * max(1.0, ref.float().abs().max().item())


@pytest.mark.parametrize("N,K,epi", [(1280, 1280, L.EPI_BIAS), (3840, 1280, L.EPI_BIAS), (6912, 1280, L.EPI_SWIGLU), (1280, 3456, L.EPI_RESIDUAL),
                                     (81920, 1280, L.EPI_BIAS)])
def test_gemm_rows_do_not_depend_on_batch_rows(hip_lib, N, K, epi):
    """Row r of X . W^T must be bit-identical whatever M is: the launcher picks other tile shapes / staging generations for other
    row counts (64x64 register-staged, 128x128 and 256x256 direct-to-LDS, decode-regime tiles), but all of them walk K in the same
    order with the same MFMA. This is what makes a line's bf16 results independent of batch composition."""
    g = torch.Generator(device="cuda").manual_seed(N + K)
    Ms = (40, 200, 700, 2200, 8200, 33000) if N < 50000 else (40, 200, 256)
    x = torch.randn(max(Ms), K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda", generator=g).to(torch.bfloat16)
    No = N // 2 if epi == L.EPI_SWIGLU else N
    res = torch.randn(max(Ms), No, device="cuda", generator=g).to(torch.bfloat16) if epi == L.EPI_RESIDUAL else None
    outs = [run_gemm(hip_lib, x[:M].contiguous(), w, b, epi, res[:M].contiguous() if res is not None else None) for M in Ms]
    for M, o in zip(Ms, outs):
        assert torch.equal(o[:40], outs[0][:40]), f"rows differ between M = {Ms[0]} and M = {M}"
