"""Torch-CPU fp32 restatement of the reference FCN forward -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product module (rcvpose_amd/models/fcnresnet.py) never does.

It restates, as a function of a plain state dict, the graph of the reference
`DenseFCNResNet152.forward` (reference models/fcnresnet.py:121-191) and `Bottleneck.forward`
(:21-45) with torch.nn.functional ops (floating-point kernel => a torch fp32 reference is the
oracle, eval-mode BatchNorm).  It is pinned to the reference by tests/golden/fcn1_64x96.npz,
fcn2_480x640_samples.npz and fcn3_backbone_64x96.npz, produced by the reference's own module
loaded with the same deterministic weights (tests/golden/make_golden.py, tests/fcn_weights.py).
"""
import numpy as np
import torch
import torch.nn.functional as F

# (prefix of the first bottleneck, prefix of the nn.Sequential of the remaining ones, count of
# the remaining ones, stride of the first) -- reference models/fcnresnet.py:59-83
_STAGES = (
    ("block1up", "block1", 2, 1),
    ("block2up", "block2", 7, 2),
    ("block3up", "block3", 35, 2),
    ("block4up", "block4", 2, 2),
)
_IMAGENET_MEAN = np.array([0.485, 0.456, 0.406])  # AccumulatorSpace.py:145
_IMAGENET_STD = np.array([0.229, 0.224, 0.225])   # AccumulatorSpace.py:146


def _bn(sd, prefix, x):
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"], sd[prefix + ".weight"],
                        sd[prefix + ".bias"], training=False, eps=1e-5)


def _bottleneck(sd, p, x, stride, project):
    """reference models/fcnresnet.py:21-45; the 3x3 carries the stride and a bias (:12)."""
    out = F.relu(_bn(sd, p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"])))
    out = F.relu(_bn(sd, p + ".bn2", F.conv2d(out, sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], stride=stride,
                                               padding=1)))
    out = _bn(sd, p + ".bn3", F.conv2d(out, sd[p + ".conv3.weight"]))
    res = x
    if project:  # `upsample_` is only applied when upsample=True (:35-36); elsewhere it is dead weight
        res = _bn(sd, p + ".upsample_.1", F.conv2d(x, sd[p + ".upsample_.0.weight"], stride=stride))
    return F.relu(out + res)


def _cbr(sd, p, x):
    """nn.Sequential(Conv2d 3x3 pad 1 (bias), BatchNorm2d, ReLU) -- reference :90-113"""
    return F.relu(_bn(sd, p + ".1", F.conv2d(x, sd[p + ".0.weight"], sd[p + ".0.bias"], padding=1)))


def _up2(x):
    return F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)


def forward(state_dict, x, return_stages=False):
    """x: float32 [B,3,H,W] (H, W multiples of 32) -> (seg [B,1,H,W], radial [B,1,H,W])."""
    sd = {k[7:] if k.startswith("module.") else k: v for k, v in state_dict.items()}
    st = {}
    with torch.no_grad():
        # stem; the ReLU is in place, so the skip tensor `x` used by conv_up1 is post-ReLU
        x = F.relu(_bn(sd, "bn1", F.conv2d(x, sd["conv1.weight"], stride=2, padding=3)))
        st["bn1"] = x
        h = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
        st["maxpool"] = h
        skips = []
        for first, rest, count, stride in _STAGES:
            h = _bottleneck(sd, first, h, stride, True)
            for i in range(count):
                h = _bottleneck(sd, "%s.%d" % (rest, i), h, 1, False)
            skips.append(h)
            st[rest] = h
        x2s, x4s, x8s, x16s = skips
        x32s = F.relu(_bn(sd, "bn6", F.conv2d(x16s, sd["conv6.weight"], sd["conv6.bias"], padding=1)))
        st["bn6"] = x32s
        up = x32s
        for name, skip, first_is_up in (("5", x16s, True), ("4", x8s, True), ("3", x4s, True), ("2", x2s, True),
                                        ("1", x, True)):
            up = _cbr(sd, "conv_up" + name, torch.cat((up, skip), 1))
            st["conv_up" + name] = up
            up = _up2(up)
            st["up" + name] = up
        up = _cbr(sd, "conv7", up)
        st["conv7"] = up
        out = F.conv2d(up, sd["conv8.weight"], sd["conv8.bias"])
        st["conv8"] = out
    seg, radial = out[:, :1], out[:, 1:]
    if return_stages:
        return seg, radial, st
    return seg, radial


# ---------------------------------------------------------------------------------------------------------------------
# bf16-emulating variant (VERDICT r01 #1a): the same graph with every tensor rounded where the bf16 engine rounds it.
#
# The engine (rcvpose_amd/csrc/fcn.hip, dtype bf16) stores weights and activations as bfloat16 and accumulates in fp32:
#   * BatchNorm is folded into the preceding convolution in float64 on the host, the folded weight is rounded to bf16
#     (round-to-nearest-even), the folded bias stays fp32;
#   * the network input is rounded to bf16 by the stem; every convolution computes sum(bf16 x bf16) in fp32 (the products
#     are exact in fp32; only the summation order is the implementation's), adds the fp32 bias, adds the residual
#     (a stored bf16 tensor) in fp32, applies ReLU and rounds the result to bf16;
#   * max-pool is exact on bf16; the bilinear x2 upsampling blends four bf16 values in fp32 and rounds to bf16;
#   * the head (conv8) is an fp32 FMA chain over the bf16 conv7 output and bf16 weights; its output stays fp32.
# This restatement rounds at exactly those places with torch's own bf16 conversion (also round-to-nearest-even) and runs
# the convolutions in fp32 on the rounded operands.  What remains between it and the engine is fp32 summation order
# (~1e-6..1e-5 relative before a rounding), which now and then tips a stored value to the neighbouring bf16 number.  ONE
# layer fed with identical inputs therefore agrees with the engine except for a ~1e-3 fraction of elements that are one
# bf16 step apart -- that is how the engine is tied to this oracle (tests/test_gpu_fcn.py::
# test_bf16_engine_layer_by_layer_against_bf16_emulation, every op of the program on the engine's own stored inputs).
# Through the whole network the flips multiply (a flipped input moves hundreds of outputs of the next layer by a
# fraction of a step, which flips some of them ...) until, a few blocks in, about half of all stored values differ by a
# step: two correct bf16 implementations with different summation orders are two independent realisations of the
# rounding noise, and differ from each other by as much as each differs from fp32 (measured: tools/bf16_emu_probe.py).
# ---------------------------------------------------------------------------------------------------------------------
_STORAGE = torch.bfloat16  # the 16-bit storage type being emulated (forward_fp16 switches it to torch.float16)


def _r16(t):
    """Round an fp32 tensor to the nearest value of the emulated 16-bit storage type (ties to even), returned as fp32."""
    return t.to(_STORAGE).to(torch.float32)


def _fold(sd, conv, bn, bias, cout):
    """(conv [+ bias]) followed by eval-mode BatchNorm as one convolution, folded in float64:
    w' = w * gamma / sqrt(var + eps),  b' = (b - mean) * gamma / sqrt(var + eps) + beta."""
    w = sd[conv + ".weight"].to(torch.float64)
    b = sd[conv + ".bias"].to(torch.float64) if bias else torch.zeros(cout, dtype=torch.float64)
    if bn is not None:
        scale = sd[bn + ".weight"].to(torch.float64) / torch.sqrt(sd[bn + ".running_var"].to(torch.float64) + 1e-5)
        w = w * scale.view(-1, 1, 1, 1)
        b = (b - sd[bn + ".running_mean"].to(torch.float64)) * scale + sd[bn + ".bias"].to(torch.float64)
    return _r16(w.to(torch.float32)), b.to(torch.float32)


def conv_bf16(sd, x, conv, bn, bias, stride=1, padding=0, relu=True, res=None):
    """One convolution of the bf16 engine: x (bf16-valued fp32, NCHW) -> stored output (bf16-valued fp32)."""
    return _conv16(sd, x, conv, bn, bias, stride=stride, padding=padding, relu=relu, res=res)


def upsample2_bf16(x):
    return _r16(_up2(x))


def head_bf16(sd, x):
    """conv8 on the stored conv7 activation: fp32 output [B,2,H,W]."""
    return F.conv2d(x, _r16(sd["conv8.weight"]), None) + sd["conv8.bias"].view(1, -1, 1, 1)


def _conv16(sd, x, conv, bn, bias, stride=1, padding=0, relu=True, res=None):
    w, b = _fold(sd, conv, bn, bias, sd[conv + ".weight"].shape[0])
    y = F.conv2d(x, w, None, stride=stride, padding=padding) + b.view(1, -1, 1, 1)
    if res is not None:
        y = y + res
    if relu:
        y = F.relu(y)
    return _r16(y)


def _bottleneck16(sd, p, x, stride, project):
    out = _conv16(sd, x, p + ".conv1", p + ".bn1", False)
    out = _conv16(sd, out, p + ".conv2", p + ".bn2", True, stride=stride, padding=1)
    res = x
    if project:
        res = _conv16(sd, x, p + ".upsample_.0", p + ".upsample_.1", False, stride=stride, relu=False)
    return _conv16(sd, out, p + ".conv3", p + ".bn3", False, res=res)


def forward_bf16(state_dict, x, return_stages=False):
    """The DenseFCNResNet152 forward as the bf16 engine computes it (see the block comment above).
    x: float32 [B,3,H,W] -> (seg, radial) float32 [B,1,H,W]; stages are the stored (bf16-valued) tensors as fp32."""
    sd = {k[7:] if k.startswith("module.") else k: v for k, v in state_dict.items()}
    st = {}
    with torch.no_grad():
        x = _conv16(sd, _r16(x), "conv1", "bn1", False, stride=2, padding=3)
        st["bn1"] = x
        h = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
        st["maxpool"] = h
        skips = []
        for first, rest, count, stride in _STAGES:
            h = _bottleneck16(sd, first, h, stride, True)
            for i in range(count):
                h = _bottleneck16(sd, "%s.%d" % (rest, i), h, 1, False)
            skips.append(h)
            st[rest] = h
        x2s, x4s, x8s, x16s = skips
        up = _conv16(sd, x16s, "conv6", "bn6", True, padding=1)
        st["bn6"] = up
        for name, skip in (("5", x16s), ("4", x8s), ("3", x4s), ("2", x2s), ("1", x)):
            up = _conv16(sd, torch.cat((up, skip), 1), "conv_up%s.0" % name, "conv_up%s.1" % name, True, padding=1)
            st["conv_up" + name] = up
            up = _r16(_up2(up))
            st["up" + name] = up
        up = _conv16(sd, up, "conv7.0", "conv7.1", True, padding=1)
        st["conv7"] = up
        out = F.conv2d(up, _r16(sd["conv8.weight"]), None) + sd["conv8.bias"].view(1, -1, 1, 1)
        st["conv8"] = out
    seg, radial = out[:, :1], out[:, 1:]
    if return_stages:
        return seg, radial, st
    return seg, radial


def forward_fp16(state_dict, x, return_stages=False):
    """forward_bf16 with IEEE half as the storage type (the engine's "fp16" mode)."""
    global _STORAGE
    _STORAGE = torch.float16
    try:
        return forward_bf16(state_dict, x, return_stages=return_stages)
    finally:
        _STORAGE = torch.bfloat16


def forward_res(state_dict, x, return_stages=False):
    """`ResFCNResNet152.forward` (reference models/fcnresnet.py:263-318): the same encoder, additive skips
    `conv_upK(up + conv_upK_1(skip))`, a final x2 upsampling before conv7.  Pinned by tests/golden/resfcn1_64x96.npz."""
    sd = {k[7:] if k.startswith("module.") else k: v for k, v in state_dict.items()}
    st = {}
    with torch.no_grad():
        x = F.relu(_bn(sd, "bn1", F.conv2d(x, sd["conv1.weight"], stride=2, padding=3)))  # in-place ReLU: `x` is post-ReLU (:266-267)
        h = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
        skips = []
        for first, rest, count, stride in _STAGES:
            h = _bottleneck(sd, first, h, stride, True)
            for i in range(count):
                h = _bottleneck(sd, "%s.%d" % (rest, i), h, 1, False)
            skips.append(h)
            st[rest] = h
        x2s, x4s, x8s, x16s = skips
        up = x16s
        for name, skip in (("4", x8s), ("3", x4s), ("2", x2s), ("1", x)):
            up = _up2(up)
            lat = F.conv2d(skip, sd["conv_up%s_1.weight" % name], sd["conv_up%s_1.bias" % name])
            up = _cbr(sd, "conv_up" + name, up + lat)
            st["conv_up" + name] = up
        up = _up2(up)  # `self.up1` a second time (:308)
        up = _cbr(sd, "conv7", up)
        st["conv7"] = up
        out = F.conv2d(up, sd["conv8.weight"], sd["conv8.bias"])
        st["conv8"] = out
    seg, radial = out[:, :1], out[:, 1:]
    if return_stages:
        return seg, radial, st
    return seg, radial


def preprocess(rgb_u8):
    """FCResBackbone's image preparation, AccumulatorSpace.py:140-151: uint8 HWC -> float64
    /255, -mean, /std -> CHW float32 [1,3,H,W]."""
    img = np.array(rgb_u8, dtype=np.float64)
    img /= 255.0
    img -= _IMAGENET_MEAN
    img /= _IMAGENET_STD
    img = img.transpose(2, 0, 1)
    return torch.from_numpy(img).float().unsqueeze(0)


def backbone(state_dict, rgb_u8):
    """FCResBackbone (AccumulatorSpace.py:122-162) on an in-memory RGB image -> (sem[H,W], radial[H,W])."""
    seg, radial = forward(state_dict, preprocess(rgb_u8))
    return np.asarray(seg[0, 0]), np.asarray(radial[0, 0])
