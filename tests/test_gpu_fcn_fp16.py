"""The engine's third storage type: IEEE half ("fp16": the bf16 kernels compiled with _Float16 and the f16 MFMA, fp32
accumulation).  Same speed as bf16, 11 instead of 8 significand bits."""
import json
import os

import numpy as np
import pytest
import torch

import fcn_weights

pytestmark = pytest.mark.gpu

REL_FP16 = 8e-3   # measured ~2-4e-3 on the radial maps (bf16: 1.5-5.5e-2)


@pytest.fixture(scope="module")
def shapes(golden_dir):
    return json.load(open(os.path.join(golden_dir, "fcn_state_dict_keys.json")))


def _sd(shapes, seed=0):
    return {k: torch.from_numpy(v) for k, v in fcn_weights.make_state_dict_numpy(shapes, seed=seed).items()}


def _rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


def test_fp16_vs_golden(shapes, golden_dir):
    from rcvpose_amd.fcn_engine import FcnEngine

    sd = _sd(shapes)
    g = np.load(os.path.join(golden_dir, "fcn1_64x96.npz"))
    x = torch.from_numpy(fcn_weights.make_input(int(g["input_seed"]), (1, 3, 64, 96))).cuda()
    seg, rad = FcnEngine([sd], 64, 96, dtype="fp16").forward(x)
    er, es = _rel(rad[0].cpu().numpy(), g["radial"]), _rel(seg[0].cpu().numpy(), g["seg"])
    print("fp16 relative error 64x96: radial %.3e seg %.3e" % (er, es))
    assert er <= REL_FP16 and es <= REL_FP16
    g = np.load(os.path.join(golden_dir, "fcn2_480x640_samples.npz"))
    x = torch.from_numpy(fcn_weights.make_input(int(g["input_seed"]), (1, 3, 480, 640))).cuda()
    seg, rad = FcnEngine([sd, _sd(shapes, 1), _sd(shapes, 2)], 480, 640, dtype="fp16").forward(x)
    err = np.abs(rad[0].cpu().numpy().reshape(-1)[g["pix"]] - g["radial"]).max() / float(g["radial_absmax"])
    print("fp16 relative error 480x640 (sampled): radial %.3e" % err)
    assert err <= REL_FP16


def test_fp16_kernel_variants_identical(shapes):
    """Every kernel family of the 16-bit engine under fp16: the fused / LDS-DMA / halo / latency kernels (engine options on)
    against the plain register-staged kernels (all options off) -- bit-identical, as for bf16."""
    from rcvpose_amd.fcn_engine import FcnEngine

    sds = [_sd(shapes, s) for s in range(3)]
    for (H, W, B) in ((480, 640, 8), (480, 640, 1), (96, 160, 3)):
        eng = FcnEngine(sds, H, W, dtype="fp16")
        x = torch.from_numpy(fcn_weights.make_input(31, (B, 3, H, W))).cuda()
        seg_a, rad_a = [t.clone() for t in eng.forward(x)]
        for key in (1, 3, 4, 5, 6, 7, 9):
            eng.set_option(key, 0)
        seg_b, rad_b = eng.forward(x)
        assert torch.equal(seg_a, seg_b) and torch.equal(rad_a, rad_b), (H, W, B)


def test_fp16_engine_layer_by_layer_against_emulation(shapes):
    """As test_bf16_engine_layer_by_layer_against_bf16_emulation, for fp16 storage: each op on the engine's own stored inputs
    against the fp16-emulating oracle -- a small fraction of elements one fp16 step apart."""
    from oracle import fcn_torch
    from rcvpose_amd import _hip, fcn_engine
    from rcvpose_amd.fcn_engine import FcnEngine

    H, W = 96, 160
    sd = _sd(shapes)
    x = torch.from_numpy(fcn_weights.make_input(11, (1, 3, H, W)))
    prog = fcn_engine.build_program(H, W)
    eng = FcnEngine([sd], H, W, dtype="fp16", reuse_slots=False)
    eng.set_option(3, 0)
    eng.set_option(4, 0)
    eng.forward(x.cuda())
    stored = lambda t: eng.debug_slot(1, 0, t, prog.shapes[t]).permute(0, 3, 1, 2).contiguous().cpu()
    fcn_torch._STORAGE = torch.float16
    try:
        worst = 0.0
        for op, spec in zip(prog.ops, prog.wspecs):
            if op["kind"] != _hip.OP_CONV:
                continue
            src = stored(op["src0"])
            if op["src1"] >= 0:
                src = torch.cat((src, stored(op["src1"])), 1)
            ref = fcn_torch.conv_bf16(sd, src, spec["conv"], spec["bn"], spec["bias"], stride=op["stride"], padding=op["pad"],
                                      relu=bool(op["relu"]), res=stored(op["res"]) if op["res"] >= 0 else None)
            got = stored(op["dst"])
            rel = float((got - ref).abs().max() / ref.abs().max())
            frac = float((got != ref).float().mean())
            worst = max(worst, frac)
            assert rel <= 2.0 ** -10 and frac <= 0.02, (spec["conv"], rel, frac)   # one fp16 step of the top binade
    finally:
        fcn_torch._STORAGE = torch.bfloat16
    print("fp16 engine vs emulation, layer by layer: worst mismatch fraction %.2e" % worst)
