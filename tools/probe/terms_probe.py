#!/usr/bin/env python3
"""CPU experiment (no GPU): which of the split-precision engine's three MFMA terms (a_hi w_hi + a_hi w_lo + a_lo w_hi) could a layer
group drop?  Emulation as in storage_bits_probe.py (22-bit storage everywhere); for the chosen convolutions either the weights are
rounded to fp16 ("no hl": w_lo dropped) or the input activations are ("no lh": a_lo dropped).  Prints the map error against fp32.
    python tools/probe/terms_probe.py [--full]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F

from oracle import fcn_torch
from rcvpose_amd import synthetic_weights


def split22(t):
    hi = t.to(torch.float16).to(torch.float32)
    return hi + (t - hi).to(torch.float16).to(torch.float32)


def r11(t):
    return t.to(torch.float16).to(torch.float32)


MODE = {"sel": lambda conv, k, cin: False, "drop": None}


def conv16(sd, x, conv, bn, bias, stride=1, padding=0, relu=True, res=None):
    w, b = fcn_torch._fold(sd, conv, bn, bias, sd[conv + ".weight"].shape[0])  # (22-bit weights: _r16 is split22 here)
    # per-output-channel power-of-two scaling does not change which bits an fp16 rounding keeps (up to subnormals): plain rounding
    if MODE["sel"](conv, w.shape[-1], w.shape[1]):
        if MODE["drop"] == "hl":
            w = r11(w)
        elif MODE["drop"] == "lh":
            x = r11(x)
    y = F.conv2d(x, w, None, stride=stride, padding=padding) + b.view(1, -1, 1, 1)
    if res is not None:
        y = y + res
    if relu:
        y = F.relu(y)
    return split22(y)


def main():
    full = "--full" in sys.argv
    shapes = json.load(open(os.path.join(ROOT, "tests", "golden", "fcn_state_dict_keys.json")))
    torch.set_num_threads(8)
    fcn_torch._r16 = split22
    fcn_torch._conv16 = conv16
    groups = {
        "none": lambda c, k, cin: False,
        "layer-3 3x3": lambda c, k, cin: k == 3 and c.startswith("block3") and c.endswith("conv2"),
        "all bottleneck 3x3": lambda c, k, cin: k == 3 and c.endswith("conv2"),
        "decoder 3x3 (conv6, conv_up*)": lambda c, k, cin: k == 3 and (c.startswith("conv_up") or c == "conv6"),
        "every 3x3": lambda c, k, cin: k == 3,
        "every conv": lambda c, k, cin: True,
    }
    for seed in (0, 1):
        sd = {k: torch.from_numpy(v) for k, v in synthetic_weights.make_state_dict_numpy(shapes, seed=seed).items()}
        g = torch.Generator().manual_seed(100 + seed)
        H, W = (480, 640) if full else (64, 96)
        x = torch.randn((1, 3, H, W), generator=g)
        seg0, rad0 = fcn_torch.forward(sd, x)
        for gname, sel in groups.items():
            for drop in (("hl", "lh") if gname != "none" else (None,)):
                MODE["sel"], MODE["drop"] = sel, drop
                seg, rad = fcn_torch.forward_bf16(sd, x)
                print("seed %d %-32s drop %-4s radial %.2e seg %.2e" % (seed, gname, drop, float((rad - rad0).abs().max() / rad0.abs().max()),
                                                                       float((seg - seg0).abs().max() / seg0.abs().max())), flush=True)


if __name__ == "__main__":
    main()
