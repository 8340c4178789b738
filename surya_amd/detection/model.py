"""HipDetModel: the detection forward pass behind libsurya_amd.so (no CPU fallback)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _lib as L
from ..config import DetConfig
from .plan import build_det_plan


class DetOpC(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("type", "in0", "in1", "out", "res", "cin", "cout", "k", "stride", "act", "hin", "win",
                                         "hout", "wout", "w_idx", "b_idx", "p0", "p1")]


class HipDetModel:
    def __init__(self, cfg: DetConfig, state_dict, *, height: int, width: int, dtype: torch.dtype = torch.bfloat16,
                 device="cuda:0", max_batch: int = 16, broadcast_weights: bool = False, process_group=None):
        if not torch.cuda.is_available():
            raise L.SuryaAmdError("HipDetModel needs a GPU (MI355X); there is no CPU fallback")
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("dtype must be float32 (reference mode) or bfloat16")
        self.lib = L.lib()
        self.lib.surya_det_create.argtypes = None
        self.cfg, self.dtype, self.device = cfg, dtype, torch.device(device)
        self.height, self.width, self.max_batch = height, width, max_batch
        torch.cuda.set_device(self.device)
        plan = build_det_plan(cfg, state_dict, height, width)
        # algorithmic FLOPs per page = the reference's own op order; the folded decode head executes fewer
        self.flops_per_image = plan.reference_flops_per_image
        self.executed_flops_per_image = plan.flops_per_image
        from .plan import OP_LITEMLA, OP_UPSUM_SRC
        self.launches_per_forward = sum(2 if o["type"] == OP_LITEMLA else (0 if o["type"] == OP_UPSUM_SRC else 1) for o in plan.ops)
        self.plan_ops = [dict(o) for o in plan.ops]
        self.weights = [w.to(device=self.device, dtype=dtype).contiguous() for w in plan.weights]
        if broadcast_weights:             # every rank planned the op list (shapes); the folded weights used are rank 0's
            from .. import dist as sdist
            if sdist.collectives_on(process_group):
                cdev = sdist.collective_device(self.device, process_group)
                tmp = [w.to(cdev) for w in self.weights]
                sdist.broadcast_tensors(tmp, src=0, group=process_group)
                for w, t in zip(self.weights, tmp):
                    w.copy_(t)
        ops = (DetOpC * len(plan.ops))(*[DetOpC(**{k: v for k, v in o.items() if k != "tag"}) for o in plan.ops])
        table = (C.c_void_p * len(self.weights))(*[w.data_ptr() for w in self.weights])
        bufs = (C.c_size_t * len(plan.buf_elems))(*plan.buf_elems)
        c = L.DetConfigC(n_ops=len(plan.ops), max_batch=max_batch, height=height, width=width, num_labels=cfg.num_labels,
                         dtype=L.DTYPE_F32 if dtype == torch.float32 else L.DTYPE_BF16)
        self.handle = C.c_void_p()
        L.check(self.lib.surya_det_create(C.byref(c), ops, table, len(self.weights), bufs, len(plan.buf_elems),
                                          C.byref(self.handle)), "surya_det_create")

    def __del__(self):
        h = getattr(self, "handle", None)
        if h:
            self.lib.surya_det_destroy(h)
            self.handle = None

    @property
    def config(self):
        return self.cfg

    def to(self, device_dtype=None):
        """See HipRecModel.to: same-placement requests are no-ops, anything else needs a new handle."""
        if device_dtype is None or device_dtype == self.dtype:
            return self
        if not isinstance(device_dtype, torch.dtype) and torch.device("cuda:0" if device_dtype == "cuda" else device_dtype) == self.device:
            return self
        raise NotImplementedError(f"HipDetModel lives on {self.device} as {self.dtype}; create a new predictor for {device_dtype}")

    def eval(self):
        return self

    def forward(self, pixel_values: torch.Tensor, want_lowres: bool = False):
        """pixel_values cuda fp32 [B,3,H,W] -> heatmaps fp32 [B, labels, H, W] (and [B, labels, H/4, W/4])."""
        assert pixel_values.is_cuda and pixel_values.dtype == torch.float32 and pixel_values.is_contiguous()
        B = pixel_values.shape[0]
        assert tuple(pixel_values.shape[1:]) == (self.cfg.num_channels, self.height, self.width) and B <= self.max_batch
        torch.cuda.set_device(self.device)
        heat = torch.empty((B, self.cfg.num_labels, self.height, self.width), dtype=torch.float32, device=self.device)
        low = torch.empty((B, self.cfg.num_labels, self.height // 4, self.width // 4), dtype=torch.float32,
                          device=self.device) if want_lowres else None
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        L.check(self.lib.surya_det_forward(self.handle, L.ptr(pixel_values), C.c_int(B), L.ptr(heat), L.ptr(low), stream),
                "surya_det_forward")
        return (heat, low) if want_lowres else heat

    def forward_timed(self, pixel_values: torch.Tensor):
        """Measurement support (surya_det_forward_timed): the same forward with a hipEvent in front of every op of the plan.
        Returns (heat maps, [(op dict, ms)]); ops folded into a fused form (sa::Tuning det_fuse) report 0 ms."""
        assert pixel_values.is_cuda and pixel_values.dtype == torch.float32 and pixel_values.is_contiguous()
        B = pixel_values.shape[0]
        torch.cuda.set_device(self.device)
        heat = torch.empty((B, self.cfg.num_labels, self.height, self.width), dtype=torch.float32, device=self.device)
        n = int(self.lib.surya_det_op_count(self.handle))
        ms = (C.c_float * n)()
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        L.check(self.lib.surya_det_forward_timed(self.handle, L.ptr(pixel_values), C.c_int(B), L.ptr(heat), L.ptr(None), stream, ms, C.c_int(n)),
                "surya_det_forward_timed")
        return heat, list(zip(self.plan_ops, [float(v) for v in ms]))

    def forward_u8(self, pages_u8: torch.Tensor, mean, std, want_lowres: bool = False):
        """pages cuda uint8 [B, H, W, 3 | 4] (RGB or RGBX, already at the processor size) -> heat maps; rescale + normalise run
        on the device."""
        assert pages_u8.is_cuda and pages_u8.dtype == torch.uint8 and pages_u8.is_contiguous()
        B, pix = pages_u8.shape[0], pages_u8.shape[3]
        assert tuple(pages_u8.shape[1:3]) == (self.height, self.width) and pix in (3, 4) and B <= self.max_batch
        torch.cuda.set_device(self.device)
        heat = torch.empty((B, self.cfg.num_labels, self.height, self.width), dtype=torch.float32, device=self.device)
        low = torch.empty((B, self.cfg.num_labels, self.height // 4, self.width // 4), dtype=torch.float32,
                          device=self.device) if want_lowres else None
        m = (C.c_float * 3)(*[float(np.float32(v)) for v in mean])
        sd = (C.c_float * 3)(*[float(np.float32(v)) for v in std])
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        L.check(self.lib.surya_det_forward_u8(self.handle, L.ptr(pages_u8), C.c_int(pix), m, sd, C.c_int(B), L.ptr(heat), L.ptr(low), stream),
                "surya_det_forward_u8")
        return (heat, low) if want_lowres else heat


class HipDetPost:
    """Heat map -> boxes on the device (surya_det_boxes, csrc/det_post.h): thresholds, 4-connected labelling, per-component
    dilation + minimum-area rectangle and confidences for all pages of a batch in a handful of launches; only the corner
    arrays come back to the host. No CPU fallback: this IS detect_boxes of the product path (surya/detection/heatmap.py:27-107)."""

    def __init__(self, device="cuda:0", max_boxes: int = 4096):
        if not torch.cuda.is_available():
            raise L.SuryaAmdError("HipDetPost needs a GPU (MI355X); there is no CPU fallback")
        self.lib = L.lib()
        self.device = torch.device(device)
        self.max_boxes = max_boxes
        self._ws = None

    def _workspace(self, B, H, W):
        need = int(self.lib.surya_det_boxes_workspace_bytes(C.c_int(B), C.c_int(H), C.c_int(W), C.c_int(self.max_boxes)))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws, need

    def launch(self, heat: torch.Tensor, text_threshold: float, low_text: float):
        """Enqueue the post-processing of a batch of maps and the D2H of its (small, fixed-size) outputs into pinned memory;
        returns a handle for `collect`. Nothing here waits for the GPU, so a caller can prepare and launch the NEXT batch before
        it looks at this one (DetectionPredictor._detect_device does)."""
        assert heat.is_cuda and heat.dtype == torch.float32 and heat.is_contiguous()
        if heat.dim() == 4:
            B, Lb, H, W = heat.shape
            stride = Lb * H * W
        else:
            B, H, W = heat.shape
            stride = H * W
        torch.cuda.set_device(self.device)
        ws, need = self._workspace(B, H, W)
        boxes = torch.empty((B, self.max_boxes, 4, 2), dtype=torch.float32, device=self.device)
        conf = torch.empty((B, self.max_boxes), dtype=torch.float32, device=self.device)
        count = torch.empty((B,), dtype=torch.int32, device=self.device)
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        L.check(self.lib.surya_det_boxes(L.ptr(heat), C.c_long(stride), C.c_int(B), C.c_int(H), C.c_int(W), C.c_float(text_threshold),
                                         C.c_float(low_text), C.c_int(self.max_boxes), L.ptr(boxes), L.ptr(conf), L.ptr(count), L.ptr(ws),
                                         C.c_size_t(need), stream), "surya_det_boxes")
        # 148 KB per page: cheaper to copy whole than to wait for the counts first
        hb = torch.empty(boxes.shape, dtype=torch.float32, pin_memory=True).copy_(boxes, non_blocking=True)
        hc = torch.empty(conf.shape, dtype=torch.float32, pin_memory=True).copy_(conf, non_blocking=True)
        hn = torch.empty(count.shape, dtype=torch.int32, pin_memory=True).copy_(count, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        return (ev, hb, hc, hn, heat, (text_threshold, low_text))     # `heat` rides along: its memory must outlive the kernels

    def collect(self, pending):
        ev, hb, hc, hn, heat, thr = pending
        ev.synchronize()
        n = hn.numpy()
        bh, ch = hb.numpy(), hc.numpy()
        out = [(bh[b, : n[b]].copy(), ch[b, : n[b]].copy()) if n[b] >= 0 else None for b in range(len(n))]
        over = [b for b in range(len(n)) if n[b] < 0]
        if over:
            # a noisy / very dense page: more kept components than the (fixed-size, copied-whole) output holds. The reference has
            # no cap, so run those pages again, alone, with room for 16 x as many boxes; only past that give up loudly.
            if self.max_boxes >= 1 << 16:
                self._raise_overflow()
            big = HipDetPost(self.device, max_boxes=self.max_boxes * 16)
            redo = big(heat[over].contiguous(), *thr)
            for b, r in zip(over, redo):
                out[b] = r
        return out

    def _raise_overflow(self):
        raise L.SuryaAmdError(f"surya_det_boxes: a page has more than max_boxes = {self.max_boxes} text components")

    def __call__(self, heat: torch.Tensor, text_threshold: float, low_text: float):
        """heat: cuda fp32, [B, H, W] contiguous, or a [B, labels, H, W] tensor whose plane 0 is the text map (then the page
        stride is labels * H * W). Returns a list of (boxes float32 [n, 4, 2], confidences float32 [n]) per page, in the
        component order cv2.connectedComponentsWithStats would label them."""
        return self.collect(self.launch(heat, text_threshold, low_text))


class DeviceResampler:
    """Pillow LANCZOS resizes of uint8 pages on the device (surya_resample_lanczos_u8, csrc/resample.h). Coefficient tables are
    built once per (source length, target length) pair by common/pil_resample.py and cached on the device."""

    def __init__(self, device="cuda:0"):
        if not torch.cuda.is_available():
            raise L.SuryaAmdError("DeviceResampler needs a GPU (MI355X)")
        self.lib = L.lib()
        self.device = torch.device(device)
        self._tables = {}

    def _axis(self, n_in: int, n_out: int):
        key = (n_in, n_out)
        t = self._tables.get(key)
        if t is None:
            from ..common.pil_resample import lanczos_coeffs
            b, kk, ks = lanczos_coeffs(n_in, n_out)
            t = self._tables[key] = (torch.from_numpy(b).to(self.device), torch.from_numpy(kk).to(self.device), ks)
        return t

    def resize(self, src: torch.Tensor, size, out: torch.Tensor = None) -> torch.Tensor:
        """src cuda uint8 [h, w, 3|4] -> Image.resize((W, H), LANCZOS) of it as uint8 [H, W, out.shape[2] or 4]."""
        assert src.is_cuda and src.dtype == torch.uint8 and src.is_contiguous() and src.dim() == 3
        h, w, sp = src.shape
        W, H = int(size[0]), int(size[1])
        if out is None:
            out = torch.empty((H, W, 4), dtype=torch.uint8, device=self.device)
        assert out.is_cuda and out.is_contiguous() and tuple(out.shape[:2]) == (H, W) and out.shape[2] in (3, 4)
        bx = kx = by = ky = None
        ksx = ksy = 0
        if W != w:
            bx, kx, ksx = self._axis(w, W)
        if H != h:
            by, ky, ksy = self._axis(h, H)
        tmp = torch.empty((h, W, 4), dtype=torch.uint8, device=self.device) if (W != w and H != h) else None
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        L.check(self.lib.surya_resample_lanczos_u8(L.ptr(src), C.c_int(w), C.c_int(h), C.c_int(sp), L.ptr(out), C.c_int(W), C.c_int(H),
                                                   C.c_int(out.shape[2]), L.ptr(bx), L.ptr(kx), C.c_int(ksx), L.ptr(by), L.ptr(ky),
                                                   C.c_int(ksy), L.ptr(tmp), stream), "surya_resample_lanczos_u8")
        return out
