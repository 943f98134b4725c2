"""oracle/fcn_torch.py (torch-CPU restatement) against outputs of the reference's own module."""
import json
import os

import numpy as np
import pytest
import torch

import fcn_weights
from oracle import fcn_torch


@pytest.fixture(scope="module")
def state_dict(golden_dir):
    shapes = json.load(open(os.path.join(golden_dir, "fcn_state_dict_keys.json")))
    assert len(shapes) == 1307
    sd = fcn_weights.make_state_dict_numpy(shapes, seed=0)
    return {k: torch.from_numpy(v) for k, v in sd.items()}


def test_fcn1_reduced_input(state_dict, golden_dir):
    g = np.load(os.path.join(golden_dir, "fcn1_64x96.npz"))
    x = torch.from_numpy(fcn_weights.make_input(int(g["input_seed"]), (1, 3, 64, 96)))
    seg, rad, st = fcn_torch.forward(state_dict, x, return_stages=True)
    scale = np.abs(g["radial"]).max()
    assert np.abs(rad.numpy() - g["radial"]).max() <= 1e-6 * scale
    assert np.abs(seg.numpy() - g["seg"]).max() <= 1e-6 * np.abs(g["seg"]).max()
    for k in g.files:
        if k.startswith("stage_"):
            ref = g[k]
            if k[6:] in ("bn1", "bn6"):  # hooked before the in-place ReLU (reference :57,:179)
                ref = np.maximum(ref, 0)
            got = st[k[6:]].numpy()
            assert got.shape == ref.shape, k
            assert np.abs(got - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max()), k


def test_fcn2_full_size_samples(state_dict, golden_dir):
    g = np.load(os.path.join(golden_dir, "fcn2_480x640_samples.npz"))
    x = torch.from_numpy(fcn_weights.make_input(int(g["input_seed"]), (1, 3, 480, 640)))
    seg, rad, st = fcn_torch.forward(state_dict, x, return_stages=True)
    pix = g["pix"]
    assert np.abs(rad.numpy().reshape(-1)[pix] - g["radial"]).max() <= 1e-6 * float(g["radial_absmax"])
    assert np.abs(seg.numpy().reshape(-1)[pix] - g["seg"]).max() <= 1e-6 * float(g["seg_absmax"])
    for k in g.files:
        if k.startswith("idx_"):
            name = k[4:]
            got = st[name].numpy().reshape(-1)[g[k]]
            ref = g["val_" + name]
            if name in ("bn1", "bn6"):
                ref = np.maximum(ref, 0)
            assert np.abs(got - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()), name


def test_fcn3_backbone_preprocessing(state_dict, golden_dir):
    g = np.load(os.path.join(golden_dir, "fcn3_backbone_64x96.npz"))
    sem, radial = fcn_torch.backbone(state_dict, g["rgb"])
    assert sem.dtype == np.float32 and sem.shape == (64, 96)
    assert np.abs(sem - g["sem"]).max() <= 1e-6 * np.abs(g["sem"]).max()
    assert np.abs(radial - g["radial"]).max() <= 1e-6 * np.abs(g["radial"]).max()


def test_resfcn_oracle_matches_reference_golden(golden_dir):
    GOLDEN = golden_dir
    """ResFCNResNet152 (SURVEY 8 f-4): the torch restatement against the reference module's own output and stages."""
    g = np.load(os.path.join(GOLDEN, "resfcn1_64x96.npz"))
    shapes = json.load(open(os.path.join(GOLDEN, "resfcn_state_dict_keys.json")))
    sd = {k: torch.from_numpy(v) for k, v in fcn_weights.make_state_dict_numpy(shapes, seed=int(g["weight_seed"])).items()}
    x = torch.from_numpy(fcn_weights.make_input(int(g["input_seed"]), (1, 3, 64, 96)))
    seg, rad, st = fcn_torch.forward_res(sd, x, return_stages=True)
    assert np.abs(seg.numpy() - g["seg"]).max() <= 1e-6 * np.abs(g["seg"]).max()
    assert np.abs(rad.numpy() - g["radial"]).max() <= 1e-6 * np.abs(g["radial"]).max()
    for k in g.files:
        if k.startswith("stage_"):
            assert np.abs(st[k[6:]].numpy() - g[k]).max() <= 1e-6 * np.abs(g[k]).max(), k


def test_resfcn_model_has_the_reference_state_dict_keys(golden_dir):
    GOLDEN = golden_dir
    from rcvpose_amd.models.fcnresnet import ResFCNResNet152

    ref = json.load(open(os.path.join(GOLDEN, "resfcn_state_dict_keys.json")))
    mine = {k: list(v.shape) for k, v in ResFCNResNet152(3, 2).state_dict().items()}
    assert list(mine.keys()) == list(ref.keys()) and mine == ref


def test_bf16_emulation_is_the_fp32_graph_with_rounded_storage():
    """oracle/fcn_torch.forward_bf16 (what the bf16 engine is compared with, tests/test_gpu_fcn.py) against the pinned fp32
    restatement: same graph, so the outputs agree to bf16-storage accuracy; every stored stage holds bf16 numbers."""
    import json
    import os

    import torch

    import fcn_weights
    from oracle import fcn_torch

    golden = os.path.join(os.path.dirname(__file__), "golden")
    shapes = json.load(open(os.path.join(golden, "fcn_state_dict_keys.json")))
    sd = {k: torch.from_numpy(v) for k, v in fcn_weights.make_state_dict_numpy(shapes, seed=0).items()}
    x = torch.from_numpy(fcn_weights.make_input(3, (1, 3, 64, 96)))
    s32, r32, st32 = fcn_torch.forward(sd, x, return_stages=True)
    s16, r16, st16 = fcn_torch.forward_bf16(sd, x, return_stages=True)
    assert set(st16) == set(st32)
    for k, v in st16.items():
        assert v.shape == st32[k].shape
        if k != "conv8":
            assert torch.equal(v, v.to(torch.bfloat16).to(torch.float32)), k  # stored activations are bf16 numbers
        assert float((v - st32[k]).abs().max() / st32[k].abs().max()) < 5e-2, k
    assert 1e-3 < float((r16 - r32).abs().max() / r32.abs().max()) < 5e-2
