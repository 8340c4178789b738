"""Lower EfficientViTForSemanticSegmentation (surya/detection/model/encoderdecoder.py:484-753) to the op list of
libsurya_amd.so's detection interpreter (include/surya_amd.h, SA_DET_*).

One-time, at load: BatchNorm (eval) is folded into the preceding conv (eps 1e-6 in the backbone, :590-591; 1e-5 in
the decode head, :691), weights are re-laid for NHWC kernels:
  dense KxK conv   [Cout, Cin, K, K] -> [Cout, K, K, Cin_pad] flattened, K axis zero-padded to a multiple of 64
  depthwise        [C, 1, K, K]      -> [K*K, C]
  grouped 1x1      [C, dim, 1, 1]    -> [C, dim]
Every op writes its own activation buffer (288 GB of HBM: no buffer-reuse pass needed; ~0.4 GB per 1024^2 page).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import torch

from ..config import DetConfig

(OP_INPUT, OP_CONV, OP_DWCONV, OP_GROUPED1X1, OP_LITEMLA, OP_UPCAT, OP_CLASSIFY, OP_UPSAMPLE_OUT, OP_UPSUM_SRC, OP_UPSUM_CLASSIFY) = range(10)
ACT_NONE, ACT_HSWISH, ACT_RELU = 0, 1, 2


def pad64(n: int) -> int:
    return (n + 63) // 64 * 64


@dataclass
class DetPlan:
    ops: List[dict] = field(default_factory=list)
    weights: List[torch.Tensor] = field(default_factory=list)      # fp32 CPU, kernel layout
    buf_elems: List[int] = field(default_factory=list)             # per image
    flops_per_image: float = 0.0                                   # of the ops as listed (what the device executes)
    reference_flops_per_image: float = 0.0                         # of the reference's own op order (the algorithmic figure of SURVEY 8(d))

    def new_buf(self, elems: int) -> int:
        self.buf_elems.append(int(elems))
        return len(self.buf_elems) - 1

    def add_weight(self, t: torch.Tensor) -> int:
        self.weights.append(t.contiguous().float())
        return len(self.weights) - 1


def _fold(sd, p, eps):
    """conv weight [Cout, ...] + optional conv bias + optional BN -> (weight, bias or None)."""
    w = sd[p + ".conv.weight"].float()
    b = sd[p + ".conv.bias"].float() if (p + ".conv.bias") in sd else None
    if (p + ".norm.weight") in sd:
        scale = sd[p + ".norm.weight"].float() / torch.sqrt(sd[p + ".norm.running_var"].float() + eps)
        w = w * scale.view(-1, *([1] * (w.dim() - 1)))
        b0 = b if b is not None else torch.zeros_like(scale)
        b = (b0 - sd[p + ".norm.running_mean"].float()) * scale + sd[p + ".norm.bias"].float()
    return w, b


def build_det_plan(cfg: DetConfig, sd: Dict[str, torch.Tensor], height: int, width: int, folded_head: bool | None = None) -> DetPlan:
    """folded_head (default: on unless DETECTOR_HEAD_UNFOLDED=1): the decode head as sum_s up(A_s x_s) + c (see the head below);
    False keeps the reference's op order (linear_c, upsample, concat, linear_fuse) -- the checker of tests/test_gpu_det.py."""
    if folded_head is None:
        from ..settings import settings
        folded_head = not settings.DETECTOR_HEAD_UNFOLDED
    pl = DetPlan()
    eps = cfg.layer_norm_eps
    CP = 8                                                   # input channels padded 3 -> 8 (16-byte NHWC pixels)

    def conv(x, hw, cin, prefix, k, stride, act, res=-1, wb=None, cin_pad=None, tag="conv"):
        """Dense conv op on buffer x ([h, w, cin] per image). Returns (out buf, (ho, wo), cout)."""
        w, b = wb if wb is not None else _fold(sd, prefix, eps)
        cout = w.shape[0]
        cin_pad = cin_pad or cin
        wk = torch.zeros((cout, k, k, cin_pad))
        wk[..., : w.shape[1]] = w.reshape(cout, w.shape[1], k, k).permute(0, 2, 3, 1)
        kreal = k * k * cin_pad
        wflat = torch.zeros((cout, pad64(kreal)))
        wflat[:, :kreal] = wk.reshape(cout, kreal)
        pad = ((stride - 1) + (k - 1)) // 2                   # get_padding, encoderdecoder.py:48-50
        ho, wo = (hw[0] + 2 * pad - k) // stride + 1, (hw[1] + 2 * pad - k) // stride + 1
        out = pl.new_buf(ho * wo * cout)
        pl.ops.append(dict(type=OP_CONV, in0=x, in1=-1, out=out, res=res, cin=cin_pad, cout=cout, k=k, stride=stride, act=act,
                           hin=hw[0], win=hw[1], hout=ho, wout=wo, w_idx=pl.add_weight(wflat),
                           b_idx=pl.add_weight(b) if b is not None else -1, p0=pad, p1=wflat.shape[1], tag=tag))
        pl.flops_per_image += 2.0 * ho * wo * cout * k * k * w.shape[1]
        return out, (ho, wo), cout

    def dwconv(x, hw, c, w, b, k, stride, act, tag="mb_dw"):
        pad = ((stride - 1) + (k - 1)) // 2 if k == 3 else k // 2
        ho, wo = (hw[0] + 2 * pad - k) // stride + 1, (hw[1] + 2 * pad - k) // stride + 1
        out = pl.new_buf(ho * wo * c)
        pl.ops.append(dict(type=OP_DWCONV, in0=x, in1=-1, out=out, res=-1, cin=c, cout=c, k=k, stride=stride, act=act, hin=hw[0],
                           win=hw[1], hout=ho, wout=wo, w_idx=pl.add_weight(w.reshape(c, k * k).t()),
                           b_idx=pl.add_weight(b) if b is not None else -1, p0=pad, p1=0, tag=tag))
        pl.flops_per_image += 2.0 * ho * wo * c * k * k
        return out, (ho, wo)

    def fused_mbconv(x, hw, cin, p, stride, res):
        y, hw2, mid = conv(x, hw, cin, p + ".spatial_conv", 3, stride, ACT_HSWISH, tag="fmb_3x3")
        return conv(y, hw2, mid, p + ".point_conv", 1, 1, ACT_NONE, res=res, tag="fmb_proj")

    def mbconv(x, hw, cin, p, stride, res):
        wd, bd = _fold(sd, p + ".depth_conv", eps)
        y, hw1, mid = conv(x, hw, cin, p + ".inverted_conv", 1, 1, ACT_HSWISH, tag="mb_expand")
        y, hw2 = dwconv(y, hw1, mid, wd, bd, 3, stride, ACT_HSWISH, tag="mb_dw")
        return conv(y, hw2, mid, p + ".point_conv", 1, 1, ACT_NONE, res=res, tag="mb_proj")

    x = pl.new_buf(height * width * CP)
    pl.ops.append(dict(type=OP_INPUT, in0=-1, in1=-1, out=x, res=-1, cin=cfg.num_channels, cout=CP, k=0, stride=0, act=0,
                       hin=height, win=width, hout=height, wout=width, w_idx=-1, b_idx=-1, p0=0, p1=0, tag="input"))
    x, hw, c = conv(x, (height, width), cfg.num_channels, "vit.stem.in_conv", 3, cfg.strides[0], ACT_HSWISH, cin_pad=CP, tag="stem")
    for r in range(cfg.depths[0]):
        p = f"vit.stem.res{r}.main"
        y, _, _ = conv(x, hw, c, p + ".conv1", 3, 1, ACT_HSWISH, tag="stem")
        x, hw, c = conv(y, hw, c, p + ".conv2", 3, 1, ACT_NONE, res=x, tag="stem")
    feats: List[Tuple[int, Tuple[int, int], int]] = []
    for si, depth in enumerate(cfg.depths[1:]):
        vit_stage, fewer = si >= 3, si >= 2
        p = f"vit.stages.{si}.blocks.0.main"
        x, hw, c = (mbconv if fewer else fused_mbconv)(x, hw, c, p, cfg.strides[si + 1], -1)
        for bi in range(1, depth + 1):
            if vit_stage:
                cp = f"vit.stages.{si}.blocks.{bi}.context_module.main"
                dim = cfg.head_dim
                q, _, td3 = conv(x, hw, c, cp + ".qkv", 1, 1, ACT_NONE, tag="mla_qkv")
                a0, _ = dwconv(q, hw, td3, sd[cp + ".aggreg.0.0.weight"].float(), None, 5, 1, ACT_NONE, tag="mla_dw5")
                a1 = pl.new_buf(hw[0] * hw[1] * td3)
                pl.ops.append(dict(type=OP_GROUPED1X1, in0=a0, in1=-1, out=a1, res=-1, cin=td3, cout=td3, k=1, stride=1, act=0,
                                   hin=hw[0], win=hw[1], hout=hw[0], wout=hw[1],
                                   w_idx=pl.add_weight(sd[cp + ".aggreg.0.1.weight"].float().reshape(td3, dim)), b_idx=-1, p0=dim, p1=0,
                                   tag="mla_g1x1"))
                pl.flops_per_image += 2.0 * hw[0] * hw[1] * td3 * dim
                td2 = 2 * td3 // 3
                att = pl.new_buf(hw[0] * hw[1] * td2)
                pl.ops.append(dict(type=OP_LITEMLA, in0=q, in1=a1, out=att, res=-1, cin=td3, cout=td2, k=0, stride=0, act=0, hin=hw[0],
                                   win=hw[1], hout=hw[0], wout=hw[1], w_idx=-1, b_idx=-1, p0=dim, p1=0, tag="mla_attn"))
                pl.flops_per_image += 2.0 * 2 * hw[0] * hw[1] * (td2 // dim) * dim * (dim + 1)
                x, _, _ = conv(att, hw, td2, cp + ".proj", 1, 1, ACT_NONE, res=x, tag="mla_proj")
                x, hw, c = mbconv(x, hw, c, f"vit.stages.{si}.blocks.{bi}.local_module.main", 1, x)
            else:
                p = f"vit.stages.{si}.blocks.{bi}.main"
                x, hw, c = (mbconv if fewer else fused_mbconv)(x, hw, c, p, 1, x)
        feats.append((x, hw, c))
    # decode head (:699-722)
    nst = len(feats)
    h0, w0 = feats[0][1]
    dl = cfg.decoder_layer_hidden_size
    wf = sd["decode_head.linear_fuse.weight"].float()
    scale = sd["decode_head.batch_norm.weight"].float() / torch.sqrt(sd["decode_head.batch_norm.running_var"].float() + 1e-5)
    bf = sd["decode_head.batch_norm.bias"].float() - sd["decode_head.batch_norm.running_mean"].float() * scale
    # what the reference's order costs (the algorithmic FLOPs of the head, whichever form runs)
    head_ref_flops = sum(2.0 * fhw[0] * fhw[1] * dl * fc for _, fhw, fc in feats) + 2.0 * h0 * w0 * wf.shape[0] * dl * nst
    flops_before_head = pl.flops_per_image
    folded = folded_head and 1 <= nst <= 4 and all(fhw == (h0, w0) for _, fhw, _ in feats[:1])
    if folded:
        # Everything between the stage outputs and the ReLU is linear, and a 1x1 convolution commutes with a bilinear resize (its taps
        # sum to 1, so constants pass through): linear_fuse(cat_s(up(W_s x_s + b_s))) = sum_s up(A_s x_s) + c with
        #   A_s = (bn_scale * W_fuse[:, block of stage s]) W_s   [decoder_hidden, C_s],   c = bn_shift + sum_s (bn_scale * W_fuse_s) b_s.
        # cat(...[::-1]) (:715) puts stage i into channel block nst - 1 - i. One 1x1 conv per stage at the stage's OWN resolution, then
        # one pass over the full-resolution stage adds the resized others, applies ReLU + classifier + sigmoid (OP_UPSUM_CLASSIFY).
        # The [h0, w0, 4 x 128] concat and the K = 512 GEMM over it are gone; fp32 results agree with the reference order to
        # re-association (tests: <= 1e-4 on the [0, 1] maps against the oracle, which keeps the reference's order).
        wfs = (wf * scale.view(-1, 1, 1, 1)).reshape(wf.shape[0], nst, dl)
        c_all = bf.clone()
        zs = []
        for i, (fb, fhw, fc) in enumerate(feats):
            blk = wfs[:, nst - 1 - i, :]                                               # [ch, dl]
            a_s = blk @ sd[f"decode_head.linear_c.{i}.proj.weight"].float().reshape(dl, fc)
            c_all = c_all + blk @ sd[f"decode_head.linear_c.{i}.proj.bias"].float()
            zs.append((a_s, fb, fhw, fc))
        z_bufs = []
        for i, (a_s, fb, fhw, fc) in enumerate(zs):
            z, _, ch = conv(fb, fhw, fc, None, 1, 1, ACT_NONE, wb=(a_s.reshape(a_s.shape[0], fc, 1, 1), c_all if i == 0 else None), tag="head_z")
            z_bufs.append((z, fhw))
        for z, fhw in z_bufs[1:]:
            pl.ops.append(dict(type=OP_UPSUM_SRC, in0=z, in1=-1, out=-1, res=-1, cin=ch, cout=ch, k=0, stride=0, act=0, hin=fhw[0],
                               win=fhw[1], hout=h0, wout=w0, w_idx=-1, b_idx=-1, p0=0, p1=0, tag="head"))
        y = z_bufs[0][0]
        pl.flops_per_image += sum(8.0 * h0 * w0 * ch for _ in z_bufs[1:])                # 4 taps, multiply-add, per addend
    else:
        cat = pl.new_buf(h0 * w0 * dl * nst)
        for i, (fb, fhw, fc) in enumerate(feats):
            wl = sd[f"decode_head.linear_c.{i}.proj.weight"].float().reshape(dl, fc, 1, 1)
            y, _, _ = conv(fb, fhw, fc, None, 1, 1, ACT_NONE, wb=(wl, sd[f"decode_head.linear_c.{i}.proj.bias"].float()), tag="head_z")
            pl.ops.append(dict(type=OP_UPCAT, in0=y, in1=-1, out=cat, res=-1, cin=dl, cout=dl * nst, k=0, stride=0, act=0, hin=fhw[0],
                               win=fhw[1], hout=h0, wout=w0, w_idx=-1, b_idx=-1, p0=(nst - 1 - i) * dl, p1=0, tag="head"))   # cat(...[::-1]) :715
        y, _, ch = conv(cat, (h0, w0), dl * nst, None, 1, 1, ACT_RELU, wb=(wf * scale.view(-1, 1, 1, 1), bf), tag="head_z")
    L = cfg.num_labels
    pl.ops.append(dict(type=OP_UPSUM_CLASSIFY if folded else OP_CLASSIFY, in0=y, in1=-1, out=-1, res=-1, cin=ch, cout=L, k=1, stride=1, act=0, hin=h0, win=w0,
                       hout=h0, wout=w0, w_idx=pl.add_weight(sd["decode_head.classifier.weight"].float().reshape(L, ch)),
                       b_idx=pl.add_weight(sd["decode_head.classifier.bias"].float()), p0=0, p1=0, tag="head"))
    pl.flops_per_image += 2.0 * h0 * w0 * ch * L
    pl.reference_flops_per_image = flops_before_head + head_ref_flops + 2.0 * h0 * w0 * ch * L
    pl.ops.append(dict(type=OP_UPSAMPLE_OUT, in0=-1, in1=-1, out=-1, res=-1, cin=0, cout=L, k=0, stride=0, act=0, hin=h0, win=w0,
                       hout=height, wout=width, w_idx=-1, b_idx=-1, p0=0, p1=0, tag="head"))
    return pl
