#!/usr/bin/env python3
"""CPU experiment (no GPU): what would a narrower activation format cost the split-precision engine in accuracy?

oracle/fcn_torch.py's storage emulation (fp32 convolutions on stored inputs, every stored tensor rounded to the storage format)
with the rounding replaced by
    22 bits: hi = fp16(v), lo = fp16(v - hi)                      (the f16x3 engine's format: 4 bytes per value)
    19 bits: hi = fp16(v), lo = (v - hi) rounded to 8 bits of hi's ulp   (3 bytes per value: VERDICT r03 item 1b)
    16 bits: the top 24 bits of the fp32 value                    (3 bytes, trivially packed)
on the golden input size (64 x 96) and on a 480 x 640 crop-free input if --full.  Prints max |map - map_fp32| / max |map_fp32|.
    python tools/probe/storage_bits_probe.py [--full]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from oracle import fcn_torch
from rcvpose_amd import synthetic_weights


def split22(t):
    hi = t.to(torch.float16).to(torch.float32)
    lo = (t - hi).to(torch.float16).to(torch.float32)
    return hi + lo


def split19(t):
    hi = t.to(torch.float16).to(torch.float32)
    # ulp of hi as an fp16 number: 2^(e - 10), e = floor(log2 |hi|) (subnormals: 2^-24)
    e = torch.floor(torch.log2(hi.abs().clamp_min(2.0 ** -14)))
    ulp = torch.exp2(e - 10)
    step = ulp / 256.0
    lo = torch.round((t - hi) / step).clamp(-128, 127) * step
    return hi + lo


def trunc24(t):
    i = t.contiguous().view(torch.int32)
    r = ((i + 0x80) & ~0xFF)  # round to nearest on the dropped byte
    return r.view(torch.float32)


def run(fmt, sd, x):
    fcn_torch._r16 = fmt
    try:
        return fcn_torch.forward_bf16(sd, x)
    finally:
        pass


def main():
    full = "--full" in sys.argv
    shapes = json.load(open(os.path.join(ROOT, "tests", "golden", "fcn_state_dict_keys.json")))
    torch.set_num_threads(8)
    for seed in (0, 1, 2):
        sd = {k: torch.from_numpy(v) for k, v in synthetic_weights.make_state_dict_numpy(shapes, seed=seed).items()}
        g = torch.Generator().manual_seed(100 + seed)
        H, W = (480, 640) if full else (64, 96)
        x = torch.randn((1, 3, H, W), generator=g)
        seg0, rad0 = fcn_torch.forward(sd, x)
        orig = fcn_torch._r16
        res = {}
        for name, fmt in (("22-bit (f16x3 now)", split22), ("19-bit (fp16 hi + 8-bit lo)", split19), ("16-bit mantissa (top 3 bytes of fp32)", trunc24)):
            seg, rad = run(fmt, sd, x)
            res[name] = (float((rad - rad0).abs().max() / rad0.abs().max()), float((seg - seg0).abs().max() / seg0.abs().max()))
        fcn_torch._r16 = orig
        print("weights seed %d, %dx%d:" % (seed, H, W), {k: "radial %.2e seg %.2e" % v for k, v in res.items()})


if __name__ == "__main__" and "--stages" not in sys.argv:
    main()


def stage_limited():
    """19-bit storage only where a 3-byte stream could be built: the block outputs inside layers 1 and 2 (consumed by the chain kernels'
    VALU epilogues); everything else stays at 22 bits.  python tools/probe/storage_bits_probe.py --stages"""
    shapes = json.load(open(os.path.join(ROOT, "tests", "golden", "fcn_state_dict_keys.json")))
    torch.set_num_threads(8)
    orig_conv = fcn_torch._conv16

    def conv16(sd, x, conv, bn, bias, stride=1, padding=0, relu=True, res=None):
        y = orig_conv(sd, x, conv, bn, bias, stride=stride, padding=padding, relu=relu, res=res)
        if conv.endswith("conv3") and (conv.startswith("block1") or conv.startswith("block2")):
            y = split19(y)
        return y

    for seed in (0, 1, 2):
        sd = {k: torch.from_numpy(v) for k, v in synthetic_weights.make_state_dict_numpy(shapes, seed=seed).items()}
        x = torch.randn((1, 3, 64, 96), generator=torch.Generator().manual_seed(100 + seed))
        seg0, rad0 = fcn_torch.forward(sd, x)
        fcn_torch._r16 = split22
        fcn_torch._conv16 = conv16
        seg, rad = fcn_torch.forward_bf16(sd, x)
        fcn_torch._conv16 = orig_conv
        print("weights seed %d: block outputs of layers 1-2 at 19 bits, the rest at 22: radial %.2e seg %.2e" % (
            seed, float((rad - rad0).abs().max() / rad0.abs().max()), float((seg - seg0).abs().max() / seg0.abs().max())))


if __name__ == "__main__" and "--stages" in sys.argv:
    stage_limited()
