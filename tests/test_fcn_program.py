"""Host-side logic of the FCN engine (no GPU): state-dict surface, layer program, BN folding, slots."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import fcn_weights
from rcvpose_amd import _hip, fcn_engine
from rcvpose_amd.models.fcnresnet import DenseFCNResNet152


def test_state_dict_surface_matches_reference(golden_dir):
    ref = json.load(open(os.path.join(golden_dir, "fcn_state_dict_keys.json")))
    sd = DenseFCNResNet152(3, 2).state_dict()
    assert list(sd.keys()) == list(ref.keys())           # 1307 keys, same order
    assert all(list(sd[k].shape) == ref[k] for k in ref)
    assert sum(p.numel() for p in DenseFCNResNet152(3, 2).parameters()) == 165009122


def test_program_work_matches_survey():
    b = fcn_engine.build_program(480, 640)
    ops, slots = fcn_engine.assign_slots(b)
    convs = [o for o in ops if o["kind"] == _hip.OP_CONV]
    assert len(convs) == 50 * 3 + 4 + 1 + 5 + 1          # bottleneck convs + projections + conv6 + conv_up + conv7
    assert fcn_engine.conv_flops(ops) == 265568256000     # 132.784 GMAC
    for o in convs:                                        # shapes the MFMA kernel supports in both dtypes
        assert o["c0"] % 64 == 0 and o["c1"] % 64 == 0 and o["cout"] % 32 == 0
    with pytest.raises(ValueError):
        fcn_engine.build_program(480, 650)


def test_slot_reuse_never_aliases_live_tensors():
    b = fcn_engine.build_program(64, 96)
    ops, slots = fcn_engine.assign_slots(b)
    raw = b.ops
    last_use = {}
    for i, o in enumerate(raw):
        for k in ("src0", "src1", "res", "dst"):
            if o[k] >= 0:
                last_use[o[k]] = i
    owner = {}
    for i, (o, r) in enumerate(zip(ops, raw)):
        for k in ("src0", "src1", "res"):
            if r[k] >= 0:
                assert owner[o[k]] == r[k], "op %d reads a slot that was overwritten" % i
        if r["dst"] >= 0:
            s = o["dst"]
            assert s not in (o["src0"], o["src1"], o["res"])
            if s in owner:
                assert last_use[owner[s]] < i
            owner[s] = r["dst"]
    assert len(slots) < len(b.tensors) / 4


def test_bn_folding_equals_conv_then_bn():
    g = torch.Generator().manual_seed(0)
    cin, cout = 8, 6
    sd = {
        "c.weight": torch.randn(cout, cin, 3, 3, generator=g),
        "c.bias": torch.randn(cout, generator=g),
        "b.weight": torch.rand(cout, generator=g) + 0.5,
        "b.bias": torch.randn(cout, generator=g),
        "b.running_mean": torch.randn(cout, generator=g),
        "b.running_var": torch.rand(cout, generator=g) + 0.5,
    }
    w, b = fcn_engine.fold_conv_bn(sd, dict(conv="c", bn="b", bias=True), cout)
    x = torch.randn(2, cin, 7, 9, generator=g, dtype=torch.float64)
    ref = F.batch_norm(F.conv2d(x, sd["c.weight"].double(), sd["c.bias"].double(), padding=1), sd["b.running_mean"].double(),
                       sd["b.running_var"].double(), sd["b.weight"].double(), sd["b.bias"].double(), False, 0.0, 1e-5)
    got = F.conv2d(x, w.permute(0, 3, 1, 2), b, padding=1)
    assert torch.allclose(got, ref, atol=1e-12, rtol=1e-12)


def test_weight_packing_layout(golden_dir):
    shapes = json.load(open(os.path.join(golden_dir, "fcn_state_dict_keys.json")))
    sd = {k: torch.from_numpy(v) for k, v in fcn_weights.make_state_dict_numpy(shapes, seed=0).items()}
    b = fcn_engine.build_program(64, 96)
    ops, _ = fcn_engine.assign_slots(b)
    w, bias, ops = fcn_engine.pack_weights([sd, {"module." + k: v for k, v in sd.items()}], b, ops, _hip.RCV_F32)
    # live weights only: the unused `upsample_` branches are not packed (117 866 210 live params)
    live = sum(o["cout"] * o["ksize"] ** 2 * (o["c0"] + (o["c1"] if o["src1"] >= 0 else 0)) for o, s in zip(ops, b.wspecs) if s)
    assert live + sum(o["cout"] for o, s in zip(ops, b.wspecs) if s and s["bias"]) <= 117866210
    assert w.numel() >= 2 * live
    for o, s in zip(ops, b.wspecs):
        if s:
            assert o["w_off"] % 64 == 0  # 16-byte vector loads need aligned rows
    # group 1 (the "module."-prefixed copy) packs to the same numbers as group 0
    o = next(o for o, s in zip(ops, b.wspecs) if s and s["conv"] == "block2up.conv2")
    n = o["cout"] * 9 * o["c0"]
    assert torch.equal(w[o["w_off"]:o["w_off"] + n], w[o["w_off"] + n:o["w_off"] + 2 * n])
    wf, bf = fcn_engine.fold_conv_bn(sd, dict(conv="block2up.conv2", bn="block2up.bn2", bias=True), o["cout"])
    assert torch.equal(w[o["w_off"]:o["w_off"] + n], wf.reshape(-1).float())


def test_slots_are_not_recycled_inside_a_fusion_window():
    """The executor may run up to FUSE_SPAN consecutive ops as one kernel whose tiles read an input while other tiles
    already write the outputs of the following ops: no output may share a slot with an input of the ops in its window."""
    from rcvpose_amd import fcn_engine

    for builder in (fcn_engine.build_program(480, 640), fcn_engine.build_program_res(96, 128)):
        ops, slots = fcn_engine.assign_slots(builder)
        for i, o in enumerate(ops):
            for j in range(max(0, i - fcn_engine.FUSE_SPAN + 1), i + 1):
                for key in ("src0", "src1", "res"):
                    if ops[j][key] >= 0:
                        assert ops[j][key] != o["dst"], (i, j, key)
        assert len(slots) <= 24


def test_split_weight_packing_roundtrip():
    """RCV_F16X3 layout: [cout][kh][kw][cin/32][hi 32 | lo 32] halves with a power-of-two scale per output channel; hi + lo,
    unscaled, reproduces the float64 weight to 2^-21 of the channel's largest magnitude; tiny weights keep normal lo halves."""
    import torch

    from rcvpose_amd import _hip, fcn_engine

    g = torch.Generator().manual_seed(0)
    for wsd in (0.03, 1e-4):
        w = torch.randn(8, 3, 3, 64, generator=g, dtype=torch.float64) * wsd
        w[3] = 0.0  # an all-zero output channel: scale 1
        flat, scale = fcn_engine.pack_split(w, _hip.OP_CONV)
        assert flat.dtype == torch.float16 and flat.numel() == 2 * w.numel() and scale.dtype == torch.float32
        v = flat.to(torch.float64).reshape(8, 3, 3, 2, 2, 32)
        back = (v[:, :, :, :, 0] + v[:, :, :, :, 1]).reshape(8, 3, 3, 64) * scale.to(torch.float64).view(-1, 1, 1, 1)
        mx = w.reshape(8, -1).abs().amax(dim=1).clamp_min(1e-30).view(-1, 1, 1, 1)
        assert float(((back - w).abs() / mx).max()) <= 2.0 ** -21
        k = torch.log2(scale.to(torch.float64))
        assert torch.equal(k, k.round()) and float(scale[3]) == 1.0
        hi_max = v[:, :, :, :, 0].abs().reshape(8, -1).amax(dim=1)
        assert all(8192 <= float(m) < 16384 for i, m in enumerate(hi_max) if i != 3)
    # stem / head: all hi halves, then all lo halves
    w = torch.randn(64, 7, 7, 3, generator=g, dtype=torch.float64) * 0.05
    flat, scale = fcn_engine.pack_split(w, _hip.OP_STEM)
    n = w.numel()
    back = (flat[:n].to(torch.float64) + flat[n:].to(torch.float64)).reshape(w.shape) * scale.to(torch.float64).view(-1, 1, 1, 1)
    assert float((back - w).abs().max() / w.abs().max()) <= 2.0 ** -21


def test_split_engine_blob_offsets():
    """pack_weights(RCV_F16X3): two halves per weight, [bias | scale] per network, offsets 128-byte aligned."""
    import json
    import os

    import torch

    import fcn_weights
    from rcvpose_amd import _hip, fcn_engine

    shapes = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "fcn_state_dict_keys.json")))
    sd = {k: torch.from_numpy(v) for k, v in fcn_weights.make_state_dict_numpy(shapes, seed=0).items()}
    b = fcn_engine.build_program(64, 96)
    ops, _ = fcn_engine.assign_slots(b)
    w, bias, ops = fcn_engine.pack_weights([sd, sd], b, ops, _hip.RCV_F16X3)
    assert w.dtype == torch.float16 and bias.dtype == torch.float32
    for op, spec in zip(ops, b.wspecs):
        if spec is None:
            continue
        assert op["w_off"] % 64 == 0 and op["b_off"] % 4 == 0
        k = op["ksize"] ** 2 * (op["c0"] + (op["c1"] if op["src1"] >= 0 else 0))
        per_net = 2 * op["cout"] * k
        assert torch.equal(w[op["w_off"]:op["w_off"] + per_net], w[op["w_off"] + per_net:op["w_off"] + 2 * per_net])
        sc = bias[op["b_off"] + op["cout"]:op["b_off"] + 2 * op["cout"]]
        assert torch.equal(torch.log2(sc), torch.log2(sc).round())


def test_hostile_state_dict_is_the_same_function():
    """tests/fcn_weights.make_hostile_state_dict (the weights of the GPU range / precision tests) re-parametrises the network
    without changing what it computes: checked with the torch oracle in float64; its statistics span the advertised ranges."""
    import json
    import os

    import torch

    import fcn_weights
    from oracle import fcn_torch

    shapes = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fcn_state_dict_keys.json")))
    sd = {k: torch.from_numpy(v) for k, v in fcn_weights.make_state_dict_numpy(shapes, seed=0).items()}
    hs = fcn_weights.make_hostile_state_dict(sd, seed=1)
    x = torch.from_numpy(fcn_weights.make_input(3, (1, 3, 32, 64))).double()
    d = lambda s: {k: (v.double() if v.dtype == torch.float32 else v) for k, v in s.items()}
    (s0, r0), (s1, r1) = fcn_torch.forward(d(sd), x), fcn_torch.forward(d(hs), x)
    assert float((r1 - r0).abs().max() / r0.abs().max()) < 2e-6 and float((s1 - s0).abs().max() / s0.abs().max()) < 2e-6
    var = torch.cat([v.flatten() for k, v in hs.items() if k.endswith("bn1.running_var") or k.endswith("bn2.running_var")])
    gam = torch.cat([v.flatten().abs() for k, v in hs.items() if k.startswith("block") and (k.endswith("bn1.weight") or k.endswith("bn2.weight"))])
    assert float(var.min()) < 2e-6 and float(var.max()) > 50 and float(gam.max() / gam.min()) > 1e4


def test_range_status_in_header_and_strerror():
    from rcvpose_amd import _hip

    L = _hip.load_library()
    assert _hip.RCV_E_RANGE == -9 and b"range" in L.rcv_strerror(_hip.RCV_E_RANGE)
    assert issubclass(_hip.HalfRangeError, FloatingPointError) and issubclass(_hip.HalfRangeError, _hip.HipLibraryError)
