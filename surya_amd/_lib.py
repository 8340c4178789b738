"""ctypes binding of libsurya_amd.so (C ABI: include/surya_amd.h).

The product path has NO fallback: if the HIP library is missing or a call fails this raises -- a silent
PyTorch/CPU path would void every parity and performance claim.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SURYA_AMD_LIB") or os.path.join(HERE, "libsurya_amd.so")   # env: A/B builds of the kernels

SA_MAX_STEPS = 16
DTYPE_F32, DTYPE_BF16 = 0, 1
(RW_PATCH, RW_MERGER_LN, RW_FC1_W, RW_FC1_B, RW_FC2_W, RW_FC2_B, RW_IMG_H, RW_IMG_W, RW_DEC_NORM, RW_TOK_EMBED, RW_LM_W,
 RW_LM_B, RW_BBOX_W, RW_BBOX_B, RW_ENC_INVFREQ, RW_DEC_INVFREQ, RW_GLOBALS) = range(17)
(RE_NORM1, RE_QKV_W, RE_QKV_B, RE_PROJ_W, RE_PROJ_B, RE_NORM2, RE_GU_W, RE_GU_B, RE_DOWN_W, RE_DOWN_B, RE_COUNT) = range(11)
(RD_LN1, RD_QKV_W, RD_QKV_B, RD_O_W, RD_LN2, RD_GU_W, RD_DOWN_W, RD_COUNT) = range(8)

EPI_BIAS, EPI_RESIDUAL, EPI_GELU, EPI_SWIGLU, EPI_HARDSWISH, EPI_RELU = range(6)


class RecConfigC(C.Structure):
    _fields_ = [
        ("enc_depth", C.c_int32), ("enc_hidden", C.c_int32), ("enc_inter", C.c_int32), ("enc_inter_pad", C.c_int32),
        ("enc_heads", C.c_int32), ("patch_dim", C.c_int32), ("patch_dim_pad", C.c_int32), ("merge", C.c_int32),
        ("window_tokens", C.c_int32), ("enc_out_hidden", C.c_int32), ("fullatt_mask", C.c_uint32), ("enc_eps", C.c_float),
        ("vocab", C.c_int32), ("dec_hidden", C.c_int32), ("dec_inter", C.c_int32), ("dec_layers", C.c_int32),
        ("dec_heads", C.c_int32), ("dec_kv_heads", C.c_int32), ("dec_head_dim", C.c_int32), ("dec_eps", C.c_float),
        ("bbox_size", C.c_int32), ("embed_multiplier", C.c_int32), ("image_token_id", C.c_int32),
        ("pad_token_id", C.c_int32), ("eos_token_id", C.c_int32), ("max_slots", C.c_int32), ("max_kv_len", C.c_int32),
        ("max_patches", C.c_int32), ("max_prefill_tokens", C.c_int32), ("dtype", C.c_int32),
    ]


class DetConfigC(C.Structure):
    _fields_ = [("n_ops", C.c_int32), ("max_batch", C.c_int32), ("height", C.c_int32), ("width", C.c_int32),
                ("num_labels", C.c_int32), ("dtype", C.c_int32)]


class SuryaAmdError(RuntimeError):
    pass


SA_OK, SA_ERR_ARG, SA_ERR_SHAPE, SA_ERR_UNSUPPORTED, SA_ERR_STATE, SA_ERR_NOMEM = 0, -1, -2, -3, -4, -5      # include/surya_amd.h
_ERR = {-1: "SA_ERR_ARG", -2: "SA_ERR_SHAPE", -3: "SA_ERR_UNSUPPORTED", -4: "SA_ERR_STATE", -5: "SA_ERR_NOMEM"}
_lib = None


def lib() -> C.CDLL:
    """Load the HIP library (once). Raises if it has not been built: there is no fallback path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SuryaAmdError(
                f"{LIB_PATH} not found: build it with `python -m surya_amd.build` (hipcc --offload-arch=gfx950). "
                "surya_amd has no CPU/PyTorch fallback for the model path.")
        # torch must load ITS libamdhip64.so.7 first: the library shares device pointers and streams with torch, so both
        # have to bind to one HIP runtime instance (same SONAME -> the loader reuses the copy that is already mapped).
        import torch  # noqa: F401
        _lib = C.CDLL(LIB_PATH)
        _lib.surya_amd_version.restype = C.c_char_p
        _lib.surya_rec_workspace_bytes.restype = C.c_size_t
        _lib.surya_det_boxes_workspace_bytes.restype = C.c_size_t
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        raise SuryaAmdError(f"{what} failed: {_ERR.get(rc, 'hipError ' + str(rc))} ({rc})")


def ptr(t):
    """Device/host pointer of a torch tensor (or None) as c_void_p."""
    if t is None:
        return C.c_void_p(0)
    return C.c_void_p(t.data_ptr())


def np_ptr(a, ctype=C.c_int32):
    return a.ctypes.data_as(C.POINTER(ctype))
