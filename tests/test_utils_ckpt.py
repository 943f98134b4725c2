"""utils.load_checkpoint mirror (reference utils.py:39-54) and the .dpt / .ply readers -- CPU only."""
import os

import numpy as np
import torch

from rcvpose_amd import utils
from rcvpose_amd.AccumulatorSpace import read_depth, read_ply_points
from rcvpose_amd.models.fcnresnet import DenseFCNResNet152


def test_load_checkpoint_roundtrip_and_module_prefix(tmp_path, capsys):
    src = DenseFCNResNet152(3, 2)
    with torch.no_grad():
        src.conv8.bias.fill_(0.25)
    path = str(tmp_path / "ape_pt1.pth.tar")
    # the dict train.py:119-128 writes, here from a DataParallel-wrapped model ("module." prefix)
    torch.save({"epoch": 7, "iteration": 1, "arch": "DenseFCNResNet152", "optim_state_dict": {},
                "model_state_dict": {"module." + k: v for k, v in src.state_dict().items()}, "best_acc_mean": 1.0,
                "loss": [0.5]}, path)
    dst = DenseFCNResNet152(3, 2)
    model, epoch, optim, loss = utils.load_checkpoint(dst, None, path)
    assert epoch == 7 and loss == [0.5]
    assert torch.equal(model.conv8.bias, src.conv8.bias)
    model, epoch, _, loss = utils.load_checkpoint(dst, None, str(tmp_path / "missing.pth.tar"))
    assert epoch == 0 and loss == [] and "no checkpoint found" in capsys.readouterr().out
    assert utils.str2bool("yes") and not utils.str2bool("0") and utils.get_config()[1]["max_iteration"] == 700000


def test_dpt_and_ply_readers(tmp_path):
    d = (np.arange(12, dtype=np.uint16).reshape(3, 4) * 37)
    p = str(tmp_path / "depth3.dpt")
    with open(p, "wb") as f:
        np.array(d.shape, dtype=np.uint32).tofile(f)
        d.tofile(f)
    assert np.array_equal(read_depth(p), d)
    pts = np.random.default_rng(1).normal(size=(50, 3)).astype(np.float32)
    pa = str(tmp_path / "a.ply")
    with open(pa, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex 50\nproperty float x\nproperty float y\nproperty float z\nend_header\n")
        for q in pts:
            f.write("%r %r %r\n" % tuple(float(v) for v in q))
    assert np.allclose(read_ply_points(pa), pts, atol=1e-7)
    pb = str(tmp_path / "b.ply")
    with open(pb, "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\nelement vertex 50\nproperty float x\nproperty float y\nproperty float z\n"
                b"property uchar red\nelement face 0\nproperty list uchar int vertex_indices\nend_header\n")
        rec = np.zeros(50, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("red", "u1")])
        rec["x"], rec["y"], rec["z"] = pts[:, 0], pts[:, 1], pts[:, 2]
        rec.tofile(f)
    assert np.array_equal(read_ply_points(pb), pts.astype(np.float64))
