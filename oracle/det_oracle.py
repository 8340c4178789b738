"""CPU oracle for the text-detection forward pass -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Plain-PyTorch fp32 restatement of EfficientViTForSemanticSegmentation.forward
(surya/detection/model/encoderdecoder.py:734-753: EfficientViT-L backbone :580-630, SegFormer-style decode head
:673-722, expit) plus the predictor-side x4 bilinear upsample (surya/detection/__init__.py:120-132), driven by a
state dict with the reference's parameter names. Pinned bit-for-bit against the real reference module through
oracle/ref_shim (oracle/make_golden.py, tests/test_oracle_vs_reference.py).
"""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.nn.functional as F

from surya_amd.config import DetConfig

SD = Dict[str, torch.Tensor]


def _cna(sd: SD, p: str, x, stride=1, groups=1, bn_eps=None, act=None):
    """ConvNormAct (encoderdecoder.py:52-86): conv (padding = get_padding) -> BatchNorm2d(eval) -> activation."""
    w = sd[p + ".conv.weight"]
    k = w.shape[-1]
    pad = ((stride - 1) + (k - 1)) // 2                                   # get_padding :48-50
    x = F.conv2d(x, w, sd.get(p + ".conv.bias"), stride=stride, padding=pad, groups=groups)
    if (p + ".norm.weight") in sd:
        x = F.batch_norm(x, sd[p + ".norm.running_mean"], sd[p + ".norm.running_var"], sd[p + ".norm.weight"],
                         sd[p + ".norm.bias"], False, 0.0, bn_eps)
    if act == "hswish":
        x = F.hardswish(x)
    elif act == "relu":
        x = F.relu(x)
    return x


def _fused_mbconv(sd, p, x, stride, eps):       # FusedMBConv :228-270 (spatial 3x3 + BN + HS, point 1x1 + BN)
    x = _cna(sd, p + ".spatial_conv", x, stride=stride, bn_eps=eps, act="hswish")
    return _cna(sd, p + ".point_conv", x, bn_eps=eps)


def _mbconv(sd, p, x, stride, eps):             # MBConv :174-225 in "fewer_norm" form: bias+HS, dw bias+HS, 1x1 + BN
    x = _cna(sd, p + ".inverted_conv", x, bn_eps=eps, act="hswish")
    x = _cna(sd, p + ".depth_conv", x, stride=stride, groups=x.shape[1], bn_eps=eps, act="hswish")
    return _cna(sd, p + ".point_conv", x, bn_eps=eps)


def _lite_mla(sd, p, x, dim, eps, attn_eps=1e-5):
    """LiteMLA :273-364: multi-scale ReLU linear attention, kv product in fp32."""
    B, _, H, W = x.shape
    qkv = _cna(sd, p + ".qkv", x)
    td3 = qkv.shape[1]
    agg = F.conv2d(qkv, sd[p + ".aggreg.0.0.weight"], None, padding=2, groups=td3)
    agg = F.conv2d(agg, sd[p + ".aggreg.0.1.weight"], None, groups=td3 // dim)
    ms = torch.cat([qkv, agg], dim=1).reshape(B, -1, 3 * dim, H * W).transpose(-1, -2)
    q, k, v = ms.chunk(3, dim=-1)
    q, k = F.relu(q), F.relu(k)
    v = F.pad(v, (0, 1), mode="constant", value=1.0)
    dt = v.dtype
    q, k, v = q.float(), k.float(), v.float()
    kv = k.transpose(-1, -2) @ v
    out = q @ kv
    out = (out[..., :-1] / (out[..., -1:] + attn_eps)).to(dt)
    out = out.transpose(-1, -2).reshape(B, -1, H, W)
    return _cna(sd, p + ".proj", out, bn_eps=eps)


def backbone(sd: SD, cfg: DetConfig, x: torch.Tensor) -> List[torch.Tensor]:
    """EfficientVitLarge.forward :623-630."""
    eps = cfg.layer_norm_eps
    x = _cna(sd, "vit.stem.in_conv", x, stride=cfg.strides[0], bn_eps=eps, act="hswish")
    for r in range(cfg.depths[0]):                                         # ConvBlock residual :130-171, :498-510
        p = f"vit.stem.res{r}.main"
        y = _cna(sd, p + ".conv1", x, bn_eps=eps, act="hswish")
        x = x + _cna(sd, p + ".conv2", y, bn_eps=eps)
    feats = []
    for si, depth in enumerate(cfg.depths[1:]):
        vit_stage, fewer = si >= 3, si >= 2
        p = f"vit.stages.{si}.blocks.0.main"
        x = (_mbconv if fewer else _fused_mbconv)(sd, p, x, cfg.strides[si + 1], eps)     # no shortcut :528-541
        for bi in range(1, depth + 1):
            if vit_stage:                                                   # EfficientVitBlock :367-403
                x = x + _lite_mla(sd, f"vit.stages.{si}.blocks.{bi}.context_module.main", x, cfg.head_dim, eps)
                x = x + _mbconv(sd, f"vit.stages.{si}.blocks.{bi}.local_module.main", x, 1, eps)
            else:
                p = f"vit.stages.{si}.blocks.{bi}.main"
                x = x + (_mbconv if fewer else _fused_mbconv)(sd, p, x, 1, eps)
        feats.append(x)
    return feats


def decode_head(sd: SD, cfg: DetConfig, feats: List[torch.Tensor]) -> torch.Tensor:
    """DecodeHead.forward :699-722 (BatchNorm2d default eps 1e-5)."""
    B = feats[-1].shape[0]
    size = feats[0].shape[2:]
    ups = []
    for i, f in enumerate(feats):
        h, w = f.shape[2], f.shape[3]
        y = F.linear(f.flatten(2).transpose(1, 2), sd[f"decode_head.linear_c.{i}.proj.weight"],
                     sd[f"decode_head.linear_c.{i}.proj.bias"])
        y = y.permute(0, 2, 1).reshape(B, -1, h, w)
        ups.append(F.interpolate(y, size=size, mode="bilinear", align_corners=False))
    x = F.conv2d(torch.cat(ups[::-1], dim=1), sd["decode_head.linear_fuse.weight"])
    x = F.batch_norm(x, sd["decode_head.batch_norm.running_mean"], sd["decode_head.batch_norm.running_var"],
                     sd["decode_head.batch_norm.weight"], sd["decode_head.batch_norm.bias"], False, 0.0, 1e-5)
    x = F.relu(x)
    return F.conv2d(x, sd["decode_head.classifier.weight"], sd["decode_head.classifier.bias"])


@torch.inference_mode()
def forward(sd: SD, cfg: DetConfig, pixel_values: torch.Tensor) -> torch.Tensor:
    """[B,3,H,W] normalised pixels -> sigmoid maps [B, num_labels, H/4, W/4] (:734-753)."""
    return torch.special.expit(decode_head(sd, cfg, backbone(sd, cfg, pixel_values)))


@torch.inference_mode()
def heatmaps(sd: SD, cfg: DetConfig, pixel_values: torch.Tensor) -> torch.Tensor:
    """Model output upsampled to the processor size in fp32, as batch_detection does (detection/__init__.py:120-132)."""
    logits = forward(sd, cfg, pixel_values)
    size = pixel_values.shape[2:]
    if tuple(logits.shape[2:]) != tuple(size):
        logits = F.interpolate(logits, size=size, mode="bilinear", align_corners=False)
    return logits.to(torch.float32)


def normalise_pages(pages) -> torch.Tensor:
    """uint8 HWC pages -> [B,3,H,W] fp32: rescale 1/255 + ImageNet normalise (SegformerImageProcessor._preprocess,
    surya/detection/processor.py:126-146). Resizing to the processor size is the caller's job (synthetic pages are
    generated at that size)."""
    import numpy as np
    mean = np.array([0.485, 0.456, 0.406], np.float32)
    std = np.array([0.229, 0.224, 0.225], np.float32)
    out = []
    for p in pages:
        a = ((p.astype(np.float64) * (1 / 255)).astype(np.float32) - mean) / std     # rescale in float64, then float32 (HF image_transforms.rescale)
        out.append(torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1))))
    return torch.stack(out)
