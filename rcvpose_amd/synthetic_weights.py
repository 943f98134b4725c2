"""Deterministic DenseFCNResNet152 weights for fixtures and tests.

Every tensor is drawn from its own numpy PCG64 stream keyed by (seed, crc32(state-dict key)),
so the same state dict can be rebuilt bit-for-bit wherever numpy is available: in the golden
generator (which loads it into the *reference* model, reference models/fcnresnet.py:48-119),
in the CPU tests (the torch-CPU restatement) and on the GPU box (rcvpose_amd.models.fcnresnet).

The distributions are chosen so that activations stay O(1) through the 50 residual blocks
("Random-weight numerics"): He-scaled convolutions, non-trivial BN
running statistics, and a damped gamma on the last BN of each bottleneck.
"""
import zlib

import numpy as np


def _rng(seed, name):
    return np.random.default_rng([int(seed), zlib.crc32(name.encode())])


def make_tensor(name, shape, seed=0):
    """Return the numpy array for state-dict entry `name` with `shape` (float32 / int64)."""
    shape = tuple(int(s) for s in shape)
    g = _rng(seed, name)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return np.zeros(shape, dtype=np.int64)
    if leaf == "running_mean":
        return g.normal(0.0, 0.1, size=shape).astype(np.float32)
    if leaf == "running_var":
        return g.uniform(0.5, 1.5, size=shape).astype(np.float32)
    if len(shape) == 4:  # convolution weight [Cout, Cin, kh, kw]
        fan_in = shape[1] * shape[2] * shape[3]
        std = (2.0 / fan_in) ** 0.5
        if name == "conv8.weight":
            std = (1.0 / fan_in) ** 0.5
        return g.normal(0.0, std, size=shape).astype(np.float32)
    if len(shape) == 1:
        parent = name.rsplit(".", 1)[0]
        is_bn = parent.split(".")[-1].startswith("bn") or _is_sequential_bn(name)
        if leaf == "weight" and is_bn:
            if parent.endswith("bn3"):
                return g.uniform(0.10, 0.30, size=shape).astype(np.float32)
            return g.uniform(0.5, 1.5, size=shape).astype(np.float32)
        if leaf == "bias" and is_bn:
            return g.normal(0.0, 0.1, size=shape).astype(np.float32)
        if leaf == "bias":  # convolution bias
            return g.normal(0.0, 0.05, size=shape).astype(np.float32)
    raise ValueError("unhandled state-dict entry %s %s" % (name, shape))


def _is_sequential_bn(name):
    # nn.Sequential(conv, bn[, relu]) members are numbered: '<seq>.1.*' is the BatchNorm
    parts = name.split(".")
    return len(parts) >= 2 and parts[-2] == "1"


def make_state_dict_numpy(shapes, seed=0):
    """shapes: mapping key -> shape (e.g. {k: v.shape for k, v in model.state_dict().items()})."""
    return {k: make_tensor(k, s, seed) for k, s in shapes.items()}


def make_state_dict(model, seed=0):
    """Torch state dict for `model` (any module exposing the reference key set)."""
    import torch

    sd = model.state_dict()
    out = {}
    for k, v in sd.items():
        out[k] = torch.from_numpy(make_tensor(k, v.shape, seed))
    return out


def make_input(seed, shape):
    """Normalised-image-like input N(0,1), float32, NCHW."""
    g = np.random.default_rng([int(seed), 0xF00D])
    return g.normal(0.0, 1.0, size=shape).astype(np.float32)
