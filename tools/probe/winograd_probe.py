#!/usr/bin/env python3
"""CPU experiment (no GPU): what would Winograd F(2x2, 3x3) cost the split-precision engine in accuracy?

The 3x3 / stride-1 convolutions are 60 % of the FCN's MFMA work; F(2x2, 3x3) needs 16 multiplies per 2x2 outputs instead of 36.
Emulation as in storage_bits_probe.py (fp32 arithmetic, every tensor the engine would hold as a pair of halves rounded to 22 bits):
the transformed input tiles B^T d B and the transformed weights G g G^T are rounded to 22 bits (they would be MFMA operands), the
channel sums run in fp32, the output transform A^T m A in fp32, then bias / ReLU / the 22-bit store as usual.
    python tools/probe/winograd_probe.py"""
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


BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def winograd_conv3x3(x, w):
    """x [B,C,H,W] (22-bit valued), w [O,C,3,3] float64 folded weights -> [B,O,H,W], padding 1, stride 1"""
    B, C, H, W = x.shape
    He, We = H + (H & 1), W + (W & 1)
    xp = F.pad(x, (1, 1 + We - W, 1, 1 + He - H))
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)  # [B,C,nH,nW,4,4]
    V = split22(torch.einsum("ij,bcnmjk,lk->bcnmil", BT, d, BT))
    U = split22(torch.einsum("ij,ocjk,lk->ocil", G, w, G).to(torch.float32))  # [O,C,4,4]
    M = torch.einsum("ocil,bcnmil->bonmil", U, V)
    Y = torch.einsum("ij,bonmjk,lk->bonmil", AT, M, AT)  # [B,O,nH,nW,2,2]
    Y = Y.permute(0, 1, 2, 4, 3, 5).reshape(B, -1, He, We)
    return Y[:, :, :H, :W]


USE = {"on": False, "sel": lambda conv: True}


def conv16(sd, x, conv, bn, bias, stride=1, padding=0, relu=True, res=None):
    cout = sd[conv + ".weight"].shape[0]
    k = sd[conv + ".weight"].shape[-1]
    if USE["on"] and k == 3 and stride == 1 and USE["sel"](conv):
        w = sd[conv + ".weight"].to(torch.float64)
        b = sd[conv + ".bias"].to(torch.float64) if bias else torch.zeros(cout, dtype=torch.float64)
        if bn is not None:
            scale = sd[bn + ".weight"].to(torch.float64) / torch.sqrt(sd[bn + ".running_var"].to(torch.float64) + 1e-5)
            w = w * scale.view(-1, 1, 1, 1)
            b = (b - sd[bn + ".running_mean"].to(torch.float64)) * scale + sd[bn + ".bias"].to(torch.float64)
        y = winograd_conv3x3(x, w) + b.to(torch.float32).view(1, -1, 1, 1)
    else:
        w, b = fcn_torch._fold(sd, conv, bn, bias, cout)
        y = F.conv2d(x, w, None, stride=stride, padding=padding) + b.view(1, -1, 1, 1)
    if res is not None:
        y = y + res
    if relu:
        y = F.relu(y)
    return split22(y)


def main():
    shapes = json.load(open(os.path.join(ROOT, "tests", "golden", "fcn_state_dict_keys.json")))
    torch.set_num_threads(8)
    fcn_torch._r16 = split22
    fcn_torch._conv16 = conv16
    groups = {"direct (now)": None, "Winograd in every 3x3 / stride 1": lambda c: True,
              "Winograd in the decoder (conv6, conv_up*)": lambda c: c.startswith("conv_up") or c == "conv6",
              "Winograd in the bottlenecks' 3x3": lambda c: c.endswith("conv2")}
    for seed in (0, 1, 2):
        sd = {k: torch.from_numpy(v) for k, v in synthetic_weights.make_state_dict_numpy(shapes, seed=seed).items()}
        x = torch.randn((1, 3, 64, 96), generator=torch.Generator().manual_seed(100 + seed))
        seg0, rad0 = fcn_torch.forward(sd, x)
        for name, sel in groups.items():
            USE["on"], USE["sel"] = sel is not None, sel or (lambda c: True)
            seg, rad = fcn_torch.forward_bf16(sd, x)
            print("seed %d %-44s radial %.2e seg %.2e" % (seed, name, float((rad - rad0).abs().max() / rad0.abs().max()),
                                                         float((seg - seg0).abs().max() / seg0.abs().max())), flush=True)


if __name__ == "__main__":
    main()
