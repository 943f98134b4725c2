"""Loads rcvpose_amd/synthetic_weights.py by file path (so that tests/golden/make_golden.py, which keeps the
repo root off sys.path while the reference is imported, can use the same generator)."""
import importlib.util
import os

_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rcvpose_amd", "synthetic_weights.py")
_spec = importlib.util.spec_from_file_location("_rcv_synthetic_weights", _path)
_mod = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_mod)

make_tensor = _mod.make_tensor
make_state_dict_numpy = _mod.make_state_dict_numpy
make_state_dict = _mod.make_state_dict
make_input = _mod.make_input


def make_hostile_state_dict(sd, seed=0):
    """A function-preserving re-parametrisation of a state dict (torch tensors) with hostile BatchNorm statistics, for the
    range / precision tests of the half-precision engines: inside every bottleneck, for bn1 and bn2 (whose only consumers are
    conv2 and conv3 of the same block),
      * running_var becomes log-uniform in [1e-6, 1e2] per channel, with the convolution in front of the BatchNorm (weights,
        bias) and running_mean rescaled so that the normalised value is unchanged;
      * gamma and beta are multiplied by g_c, log-uniform in [1e-2, 1e2] per channel, and input channel c of the consuming
        convolution is divided by g_c (ReLU commutes with a positive scale).
    In exact arithmetic the network computes the same maps; its folded weights span ten orders of magnitude per layer and its
    `mid` activations four per tensor -- what a released checkpoint with small running variances would look like."""
    import numpy as np
    import torch

    out = {k: v.clone() for k, v in sd.items()}
    rng = np.random.default_rng([int(seed), 0xBAD])
    eps = 1e-5
    blocks = sorted({k[: -len(".conv1.weight")] for k in sd if k.endswith(".conv1.weight") and k.startswith("block")})
    for p in blocks:
        for bn, conv, consumer in ((".bn1", ".conv1", ".conv2"), (".bn2", ".conv2", ".conv3")):
            c = out[p + bn + ".weight"].numel()
            var_new = torch.from_numpy(np.exp(rng.uniform(np.log(1e-6), np.log(1e2), size=c)))
            a = torch.sqrt((var_new + eps) / (out[p + bn + ".running_var"].double() + eps))
            out[p + conv + ".weight"] = (out[p + conv + ".weight"].double() * a.view(-1, 1, 1, 1)).float()
            if p + conv + ".bias" in out:
                out[p + conv + ".bias"] = (out[p + conv + ".bias"].double() * a).float()
            out[p + bn + ".running_mean"] = (out[p + bn + ".running_mean"].double() * a).float()
            out[p + bn + ".running_var"] = var_new.float()
            g = torch.from_numpy(np.exp(rng.uniform(np.log(1e-2), np.log(1e2), size=c)))
            out[p + bn + ".weight"] = (out[p + bn + ".weight"].double() * g).float()
            out[p + bn + ".bias"] = (out[p + bn + ".bias"].double() * g).float()
            out[p + consumer + ".weight"] = (out[p + consumer + ".weight"].double() / g.view(1, -1, 1, 1)).float()
    return out
