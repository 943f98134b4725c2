#!/usr/bin/env python3
"""CPU experiment (no GPU): could the two cross terms of the split-precision product, a_hi w_lo + a_lo w_hi, run on the block-scaled
8-bit / 6-bit matrix instruction (v_mfma_scale_f32_16x16x128_f8f6f4: twice / four times the fp16 rate), leaving only a_hi w_hi in fp16?

Emulation as in terms_probe.py: every tensor the engine holds is rounded to 22 bits (pair of halves); a selected convolution is
computed as   conv(a_hi, w_hi)  [exact in fp32, as the fp16 MFMA with fp32 accumulation is]
            + conv(q(a_hi), q(w_lo)) + conv(q(a_lo), q(w_hi))
with q one of
  sig4 / sig3 : round to a 4- / 3-bit significand, unlimited exponent range (what e4m3 / e5m2 give with perfect scaling)
  mx-e4m3     : OCP MX: blocks of 32 along the channel axis share a power-of-two scale (largest element -> top binade), elements e4m3
  mx-e2m3     : the same with fp6 e2m3 elements (4x the fp16 rate)
Prints the map error against the fp32 forward next to the plain 22-bit engine's.
    python tools/probe/fp8_cross_probe.py [--full] [--only 3x3]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F

from oracle import fcn_torch
from rcvpose_amd import synthetic_weights


def r11(t):
    return t.to(torch.float16).to(torch.float32)


def split22(t):
    hi = r11(t)
    return hi + r11(t - hi)


def sig(t, bits):
    """round to `bits` significand bits (implicit one included), any exponent"""
    m, e = torch.frexp(t)                      # t = m 2^e, 0.5 <= |m| < 1
    return torch.ldexp(torch.round(m * (1 << bits)) / (1 << bits), e)


def mx(t, axis, mbits, emin, emax_elem, blk=32, nosat=False):
    """OCP MX block format along `axis` (blocks of 32): shared scale 2^(floor(log2 max) - emax_elem); elements with `mbits` mantissa
    bits, normal exponents emin..emax_elem, subnormals below, saturation above"""
    t = t.movedim(axis, -1)
    shp = t.shape
    C = shp[-1]
    pad = (-C) % blk
    tt = F.pad(t, (0, pad)).reshape(*shp[:-1], -1, blk)
    amax = tt.abs().amax(dim=-1, keepdim=True)
    _, e = torch.frexp(amax)                   # amax = m 2^e, m in [0.5, 1) -> floor(log2 amax) = e - 1
    top = (2 - 2.0 ** -mbits) * 2.0 ** emax_elem
    if mbits == 3 and emax_elem == 8:          # e4m3fn: the all-ones mantissa of the top binade is NaN: max 448
        top = 448.0
    scale = torch.ldexp(torch.ones_like(amax), (e - 1) - emax_elem)
    if nosat:                                  # one binade up when the largest element would round beyond the format's top
        half_ulp = 2.0 ** (emax_elem - mbits - 1)
        scale = torch.where(amax / scale >= top + half_ulp, scale * 2, scale)
    scale = torch.where(amax > 0, scale, torch.ones_like(scale))
    v = tt / scale
    _, ev = torch.frexp(v)
    ev = torch.clamp(ev - 1, min=emin)         # exponent of the element's binade, subnormals share emin
    q = torch.ldexp(torch.ones_like(v), ev - mbits)
    vq = torch.round(v / q) * q
    vq = torch.clamp(vq, -top, top)
    out = (vq * scale).reshape(*shp[:-1], -1)[..., :C]
    return out.movedim(-1, axis)


Q = {
    "sig4": lambda t, axis: sig(t, 4),
    "sig3": lambda t, axis: sig(t, 3),
    "sig5": lambda t, axis: sig(t, 5),
    "sig6": lambda t, axis: sig(t, 6),
    "mx-e4m3": lambda t, axis: mx(t, axis, 3, -6, 8),
    "mx-e2m3": lambda t, axis: mx(t, axis, 3, 0, 2),
    "mx-e3m2": lambda t, axis: mx(t, axis, 2, -2, 4),
    "mx16-e2m3": lambda t, axis: mx(t, axis, 3, 0, 2, blk=16),
    "mx16ns-e2m3": lambda t, axis: mx(t, axis, 3, 0, 2, blk=16, nosat=True),
    "mx32ns-e2m3": lambda t, axis: mx(t, axis, 3, 0, 2, blk=32, nosat=True),
}
MODE = {"sel": lambda conv, k: False, "q": None}


def conv16(sd, x, conv, bn, bias, stride=1, padding=0, relu=True, res=None):
    w, b = fcn_torch._fold(sd, conv, bn, bias, sd[conv + ".weight"].shape[0])
    if MODE["q"] is not None and MODE["sel"](conv, w.shape[-1]):
        q = Q[MODE["q"]]
        a_hi = r11(x)
        a_lo = x - a_hi
        w_hi = r11(w)
        w_lo = w - w_hi
        y = F.conv2d(a_hi, w_hi, None, stride=stride, padding=padding)
        y = y + F.conv2d(q(a_hi, 1), q(w_lo, 1), None, stride=stride, padding=padding)
        y = y + F.conv2d(q(a_lo, 1), q(w_hi, 1), None, stride=stride, padding=padding)
        y = y + b.view(1, -1, 1, 1)
    else:
        y = F.conv2d(x, w, None, stride=stride, padding=padding) + b.view(1, -1, 1, 1)
    if res is not None:
        y = y + res
    if relu:
        y = F.relu(y)
    return split22(y)


def main():
    full = "--full" in sys.argv
    only33 = "--only" in sys.argv
    shapes = json.load(open(os.path.join(ROOT, "tests", "golden", "fcn_state_dict_keys.json")))
    torch.set_num_threads(8)
    fcn_torch._r16 = split22
    fcn_torch._conv16 = conv16
    groups = {
        "layer-3 3x3": lambda c, k: k == 3 and c.startswith("block3") and c.endswith("conv2"),
        "every 3x3 stride 1/2": lambda c, k: k == 3,
        "every conv": lambda c, k: True,
    }
    if only33:
        groups.pop("every conv")
    if "--each" in sys.argv:
        groups = {n: (lambda c, k, n=n: c == n) for n in ("conv6", "conv_up5.0", "conv_up4.0", "conv_up3.0", "conv_up2.0", "conv_up1.0", "conv7.0")}
    elif "--enc" in sys.argv:
        import re
        groups = {
            "block[234].N.conv2": lambda c, k: re.fullmatch(r"block[234]\.\d+\.conv2", c) is not None,
            "block[1234].N.conv2": lambda c, k: re.fullmatch(r"block[1234]\.\d+\.conv2", c) is not None,
            "block[234].N.conv2 + conv6": lambda c, k: re.fullmatch(r"block[234]\.\d+\.conv2", c) is not None or c == "conv6",
        }
    elif "--groups" in sys.argv:
        l3 = groups["layer-3 3x3"]
        dec = lambda c, k: k == 3 and (c.startswith("conv_up") or c == "conv6")
        groups = {
            "decoder 3x3 (conv6, conv_up1-5)": dec,
            "conv6 + conv_up5 + conv_up4": lambda c, k: k == 3 and c in ("conv6", "conv_up5.0", "conv_up4.0"),
            "layer-4 3x3": lambda c, k: k == 3 and c.startswith("block4") and c.endswith("conv2"),
            "layers 1-2 3x3": lambda c, k: k == 3 and (c.startswith("block1") or c.startswith("block2")) and c.endswith("conv2"),
            "layer 3 + decoder": lambda c, k: l3(c, k) or dec(c, k),
            "layer 3 + layer 4 + decoder": lambda c, k: l3(c, k) or dec(c, k) or (k == 3 and c.startswith("block4") and c.endswith("conv2")),
        }
    qs = [a for a in sys.argv[1:] if a in Q] or ["sig4", "mx-e4m3", "mx-e2m3", "sig3", "sig6"]
    for seed in (0, 1):
        sd = {k: torch.from_numpy(v) for k, v in synthetic_weights.make_state_dict_numpy(shapes, seed=seed).items()}
        g = torch.Generator().manual_seed(100 + seed)
        H, W = (480, 640) if full else (64, 96)
        x = torch.randn((1, 3, H, W), generator=g)
        seg0, rad0 = fcn_torch.forward(sd, x)
        MODE["q"] = None
        seg, rad = fcn_torch.forward_bf16(sd, x)
        print("seed %d %-24s %-8s radial %.2e seg %.2e" % (seed, "(plain 22-bit engine)", "-", float((rad - rad0).abs().max() / rad0.abs().max()),
                                                          float((seg - seg0).abs().max() / seg0.abs().max())), flush=True)
        for gname, sel in groups.items():
            for qn in qs:
                MODE["sel"], MODE["q"] = sel, qn
                seg, rad = fcn_torch.forward_bf16(sd, x)
                print("seed %d %-24s %-8s radial %.2e seg %.2e" % (seed, gname, qn, float((rad - rad0).abs().max() / rad0.abs().max()),
                                                                  float((seg - seg0).abs().max() / seg0.abs().max())), flush=True)


if __name__ == "__main__":
    main()
