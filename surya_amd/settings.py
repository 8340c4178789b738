"""Environment-driven settings with the reference's variable names (surya/settings.py:12-190).

pydantic_settings / dotenv are not in the image, so this is a small env reader; only the knobs the hot path
reads are kept (the reference's S3 / font / dataset settings are out of scope).
"""
from __future__ import annotations

import os
from typing import Optional


def _env(name, cast, default):
    v = os.environ.get(name)
    if v is None or v == "":
        return default
    if cast is bool:
        return v.lower() in ("1", "true", "yes", "on")
    return cast(v)


class Settings:
    def __init__(self):
        self.reload()

    def reload(self):
        self.TORCH_DEVICE: Optional[str] = _env("TORCH_DEVICE", str, None)
        self.DISABLE_TQDM: bool = _env("DISABLE_TQDM", bool, False)
        self.LOGLEVEL: str = _env("LOGLEVEL", str, "INFO")
        self.MODEL_CACHE_DIR: Optional[str] = _env("MODEL_CACHE_DIR", str, None)
        # detection (settings.py:54-73)
        self.DETECTOR_BATCH_SIZE: Optional[int] = _env("DETECTOR_BATCH_SIZE", int, None)
        self.DETECTOR_IMAGE_CHUNK_HEIGHT: int = _env("DETECTOR_IMAGE_CHUNK_HEIGHT", int, 1400)
        self.DETECTOR_TEXT_THRESHOLD: float = _env("DETECTOR_TEXT_THRESHOLD", float, 0.6)
        self.DETECTOR_BLANK_THRESHOLD: float = _env("DETECTOR_BLANK_THRESHOLD", float, 0.35)
        self.DETECTOR_POSTPROCESSING_CPU_WORKERS: int = _env("DETECTOR_POSTPROCESSING_CPU_WORKERS", int,
                                                            min(8, os.cpu_count() or 1))
        self.DETECTOR_MIN_PARALLEL_THRESH: int = _env("DETECTOR_MIN_PARALLEL_THRESH", int, 3)
        self.DETECTOR_POSTPROCESS_HOST: bool = _env("DETECTOR_POSTPROCESS_HOST", bool, False)   # this implementation only
        self.DETECTOR_RESIZE_HOST: bool = _env("DETECTOR_RESIZE_HOST", bool, False)   # this implementation only
        # keep the decode head in the reference's op order (linear_c, upsample, concat, linear_fuse) instead of the folded form
        # (detection/plan.py): the checker of the folded head in tests/test_gpu_det.py
        self.DETECTOR_HEAD_UNFOLDED: bool = _env("DETECTOR_HEAD_UNFOLDED", bool, False)   # this implementation only
        self.DETECTOR_BOX_Y_EXPAND_MARGIN: float = _env("DETECTOR_BOX_Y_EXPAND_MARGIN", float, 0.05)
        # recognition (settings.py:77-94)
        self.RECOGNITION_MAX_TOKENS: Optional[int] = _env("RECOGNITION_MAX_TOKENS", int, None)
        self.RECOGNITION_BATCH_SIZE: Optional[int] = _env("RECOGNITION_BATCH_SIZE", int, None)
        self.RECOGNITION_CHUNK_SIZE: Optional[int] = _env("RECOGNITION_CHUNK_SIZE", int, None)
        self.RECOGNITION_PAD_VALUE: int = _env("RECOGNITION_PAD_VALUE", int, 255)
        # this implementation only
        # crop / pad / resize / normalise / patchify of the line crops on the GPU (surya_rec_preprocess); 1 keeps the host
        # (numpy, thread pool) pre-processing of round 1 -- the checker of tests/test_gpu_prep.py, not a fallback
        self.RECOGNITION_PREPROCESS_HOST: bool = _env("RECOGNITION_PREPROCESS_HOST", bool, False)
        # pause Python's cyclic GC while a predictor call assembles its result objects (recognition/predictor.py gc_paused)
        self.SURYA_AMD_PAUSE_GC: bool = _env("SURYA_AMD_PAUSE_GC", bool, True)
        # decode steps on MXFP8 weights + activations (csrc/gemm_mx.h; BASELINE.json configs[4]). Off by default: the
        # reference computes in the checkpoint dtype, and fp8 changes which token wins a near-tie
        self.RECOGNITION_DECODE_FP8: bool = _env("RECOGNITION_DECODE_FP8", bool, False)
        # decode steps on an fp8 KV cache (csrc/decode_attn_kv8.h): pays at long horizons (texify); off by default for the same reason
        self.RECOGNITION_KV_FP8: bool = _env("RECOGNITION_KV_FP8", bool, False)
        # continuous batching: refill (prefill new lines) once more than this share of the slots is empty; 0.2 = the reference's
        # RecognitionPredictor.min_prefill_ratio (surya/recognition/__init__.py:71). Scheduling only: results do not depend on it.
        self.RECOGNITION_MIN_PREFILL_RATIO: float = _env("RECOGNITION_MIN_PREFILL_RATIO", float, 0.2)
        self.RECOGNITION_STEPS_PER_SYNC: int = _env("RECOGNITION_STEPS_PER_SYNC", int, 4)
        # layout / table recognition: decode steps per device-fed run (the host reads the records of this many boxes at a time; the
        # boxes after a page's end token are discarded, so results do not depend on it). 1..16.
        self.LAYOUT_STEPS_PER_SYNC: int = _env("LAYOUT_STEPS_PER_SYNC", int, 8)
        self.RECOGNITION_ENCODE_AHEAD: bool = bool(_env("RECOGNITION_ENCODE_AHEAD", int, 1))
        # detect -> recognise calls admit the first pages' lines while the detector still works on the later pages (predictor._call_streamed)
        self.RECOGNITION_STREAM_DETECTION: bool = bool(_env("RECOGNITION_STREAM_DETECTION", int, 1))
        # multi-GPU: shard ONE call's lines / pages over the ranks of the initialised process group (all ranks must pass the
        # same inputs). Off by default: the reference has no collectives, and a torchrun job where every rank OCRs its own
        # pages must not meet one.
        self.SURYA_AMD_SHARD: bool = _env("SURYA_AMD_SHARD", bool, False)
        # rank 0 repacks the weights, the other ranks receive them through one bucketed RCCL broadcast at construction
        self.SURYA_AMD_BROADCAST_WEIGHTS: bool = _env("SURYA_AMD_BROADCAST_WEIGHTS", bool, False)
        self.SURYA_AMD_REC_CONFIG: str = _env("SURYA_AMD_REC_CONFIG", str, "REC-FULL")
        self.SURYA_AMD_DET_CONFIG: str = _env("SURYA_AMD_DET_CONFIG", str, "DET-DEFAULT")

    @property
    def TORCH_DEVICE_MODEL(self) -> str:
        """settings.py:32-52 -- explicit TORCH_DEVICE wins, else cuda (= ROCm/HIP here) if visible, else cpu."""
        if self.TORCH_DEVICE is not None:
            return self.TORCH_DEVICE
        import torch
        return "cuda" if torch.cuda.is_available() else "cpu"


settings = Settings()
