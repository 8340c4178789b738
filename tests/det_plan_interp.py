"""Test infrastructure: a plain-PyTorch fp32 interpreter of the detector's op list (surya_amd/detection/plan.py -> include/surya_amd.h
SA_DET_*), op by op as csrc/det_model.hip executes it. It lets the CPU tier check what the plan LOWERS -- BatchNorm folding, the NHWC
weight layouts, the K padding, the folded decode head's merged weights -- against the oracle (= the reference's own op order) without a
GPU. Never imported by the product."""
import torch
import torch.nn.functional as F

from surya_amd.detection import plan as P


def _act(y, act):
    if act == P.ACT_HSWISH:
        return F.hardswish(y)
    if act == P.ACT_RELU:
        return F.relu(y)
    return y


def run_plan(pl: P.DetPlan, pixel_values: torch.Tensor):
    """pixel_values [B, 3, H, W] fp32 (normalised) -> (sigmoid planes [B, L, H/4, W/4], heat maps [B, L, H, W])."""
    bufs, addends, planes, heat = {}, [], None, None
    B = pixel_values.shape[0]
    nchw = lambda t: t.permute(0, 3, 1, 2)          # buffers are NHWC like the device's
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous()
    for op in pl.ops:
        t = op["type"]
        W_ = pl.weights[op["w_idx"]] if op["w_idx"] >= 0 else None
        b_ = pl.weights[op["b_idx"]] if op["b_idx"] >= 0 else None
        if t == P.OP_INPUT:
            x = nhwc(pixel_values.float())
            bufs[op["out"]] = F.pad(x, (0, op["cout"] - x.shape[-1]))
        elif t == P.OP_CONV:
            k, cin, cout = op["k"], op["cin"], op["cout"]
            assert W_.shape == (cout, op["p1"]) and op["p1"] % 64 == 0 and op["p1"] >= k * k * cin
            w = W_[:, : k * k * cin].reshape(cout, k, k, cin).permute(0, 3, 1, 2)        # [Cout][ky][kx][Cin] -> OIHW
            assert (W_[:, k * k * cin:] == 0).all()                                       # the K padding is zero
            y = F.conv2d(nchw(bufs[op["in0"]]), w, b_, stride=op["stride"], padding=op["p0"])
            y = _act(y, op["act"])
            if op["res"] >= 0:
                y = y + nchw(bufs[op["res"]])
            assert y.shape[2:] == (op["hout"], op["wout"])
            bufs[op["out"]] = nhwc(y)
        elif t == P.OP_DWCONV:
            k, c = op["k"], op["cin"]
            w = W_.t().reshape(c, 1, k, k)                                                # [K*K][C] -> depthwise OIHW
            y = _act(F.conv2d(nchw(bufs[op["in0"]]), w, b_, stride=op["stride"], padding=op["p0"], groups=c), op["act"])
            assert y.shape[2:] == (op["hout"], op["wout"])
            bufs[op["out"]] = nhwc(y)
        elif t == P.OP_GROUPED1X1:
            c, gd = op["cin"], op["p0"]
            bufs[op["out"]] = nhwc(F.conv2d(nchw(bufs[op["in0"]]), W_.reshape(c, gd, 1, 1), None, groups=c // gd))
        elif t == P.OP_LITEMLA:
            dim, heads = op["p0"], op["cout"] // op["p0"]
            qa, qb = bufs[op["in0"]].flatten(1, 2), bufs[op["in1"]].flatten(1, 2)         # [B, HW, heads_a * 3 * dim] each
            outs = []
            for h in range(heads):
                src, hh = (qa, h) if h < heads // 2 else (qb, h - heads // 2)
                q, k_, v = (src[..., hh * 3 * dim + j * dim: hh * 3 * dim + (j + 1) * dim] for j in range(3))
                q, k_ = F.relu(q), F.relu(k_)
                v1 = torch.cat([v, torch.ones_like(v[..., :1])], -1)
                o = q @ (k_.transpose(1, 2) @ v1)                                          # [B, HW, dim + 1]
                outs.append(o[..., :dim] / (o[..., dim:] + 1e-5))
            bufs[op["out"]] = torch.cat(outs, -1).reshape(B, op["hin"], op["win"], op["cout"])
        elif t == P.OP_UPCAT:
            if op["out"] not in bufs:
                bufs[op["out"]] = torch.zeros(B, op["hout"], op["wout"], op["cout"])
            up = F.interpolate(nchw(bufs[op["in0"]]), size=(op["hout"], op["wout"]), mode="bilinear", align_corners=False)
            bufs[op["out"]][..., op["p0"]: op["p0"] + op["cin"]] = nhwc(up)
        elif t == P.OP_UPSUM_SRC:
            addends.append(bufs[op["in0"]])
        elif t in (P.OP_CLASSIFY, P.OP_UPSUM_CLASSIFY):
            y = bufs[op["in0"]]
            if t == P.OP_UPSUM_CLASSIFY:
                for z in addends:
                    y = y + nhwc(F.interpolate(nchw(z), size=(op["hin"], op["win"]), mode="bilinear", align_corners=False))
                addends = []
                y = F.relu(y)
            planes = torch.special.expit(nchw(y @ W_.t() + b_))
        elif t == P.OP_UPSAMPLE_OUT:
            heat = F.interpolate(planes, size=(op["hout"], op["wout"]), mode="bilinear", align_corners=False)
        else:
            raise ValueError(f"unknown op type {t}")
    return planes, heat
