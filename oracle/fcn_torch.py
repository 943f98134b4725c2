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
