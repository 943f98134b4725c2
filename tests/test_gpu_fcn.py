"""HIP FCN engine (through the C ABI) against the golden vectors of the reference module and the
torch-CPU oracle.  fp32 mode carries the 1e-4 relative gate of BASELINE.json; bf16 is the
performance mode with its own measured tolerance."""
import json
import os
import tempfile

import numpy as np
import pytest
import torch

import fcn_weights

pytestmark = pytest.mark.gpu

REL_FP32 = 1e-4   # north star: radial maps within 1e-4 relative (fp32)
REL_BF16 = 6e-2   # bf16 storage through 152 layers; measured ~1-2e-2, see DESIGN.md


@pytest.fixture(scope="module")
def shapes(golden_dir):
    return json.load(open(os.path.join(golden_dir, "fcn_state_dict_keys.json")))


def _sd(shapes, seed=0):
    return {k: torch.from_numpy(v) for k, v in fcn_weights.make_state_dict_numpy(shapes, seed=seed).items()}


@pytest.fixture(scope="module")
def state_dict(shapes):
    return _sd(shapes, 0)


def _rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


def test_fp32_stages_and_output_vs_golden(state_dict, golden_dir):
    from rcvpose_amd.fcn_engine import FcnEngine

    g = np.load(os.path.join(golden_dir, "fcn1_64x96.npz"))
    x = torch.from_numpy(fcn_weights.make_input(int(g["input_seed"]), (1, 3, 64, 96))).cuda()
    eng = FcnEngine([state_dict], 64, 96, dtype="fp32", reuse_slots=False)
    seg, rad = eng.forward(x)
    for k in g.files:
        if not k.startswith("stage_") or k[6:] == "conv8":
            continue
        ref = g[k]
        if k[6:] in ("bn1", "bn6"):
            ref = np.maximum(ref, 0)
        got = eng.debug_stage(1, 0, k[6:]).cpu().numpy()
        assert got.shape == ref.shape, k
        assert _rel(got, ref) <= REL_FP32, (k, _rel(got, ref))
    assert _rel(rad[0].cpu().numpy(), g["radial"]) <= REL_FP32
    assert _rel(seg[0].cpu().numpy(), g["seg"]) <= REL_FP32


def test_fp32_full_size_vs_golden_samples(state_dict, golden_dir):
    from rcvpose_amd.models.fcnresnet import DenseFCNResNet152

    g = np.load(os.path.join(golden_dir, "fcn2_480x640_samples.npz"))
    model = DenseFCNResNet152(3, 2)
    model.load_state_dict(state_dict, strict=True)
    model.eval().to("cuda")
    x = torch.from_numpy(fcn_weights.make_input(int(g["input_seed"]), (1, 3, 480, 640))).cuda()
    seg, rad = model(x)
    assert seg.shape == (1, 1, 480, 640) and rad.shape == (1, 1, 480, 640) and rad.dtype == torch.float32
    pix = g["pix"]
    err_r = np.abs(rad.cpu().numpy().reshape(-1)[pix] - g["radial"]).max() / float(g["radial_absmax"])
    err_s = np.abs(seg.cpu().numpy().reshape(-1)[pix] - g["seg"]).max() / float(g["seg_absmax"])
    assert err_r <= REL_FP32 and err_s <= REL_FP32, (err_r, err_s)


def test_bf16_vs_golden(state_dict, golden_dir):
    from rcvpose_amd.fcn_engine import FcnEngine

    g = np.load(os.path.join(golden_dir, "fcn1_64x96.npz"))
    x = torch.from_numpy(fcn_weights.make_input(int(g["input_seed"]), (1, 3, 64, 96))).cuda()
    eng = FcnEngine([state_dict], 64, 96, dtype="bf16")
    seg, rad = eng.forward(x)
    er, es = _rel(rad[0].cpu().numpy(), g["radial"]), _rel(seg[0].cpu().numpy(), g["seg"])
    print("bf16 relative error: radial %.3e seg %.3e" % (er, es))
    assert er <= REL_BF16 and es <= REL_BF16


def test_bf16_full_size_both_conv_kernels(shapes, state_dict, golden_dir):
    """480x640, three networks: the LDS-DMA convolution kernel (used where the grid is large enough) and the
    register-staged one give the same bf16 result, and both sit within the bf16 tolerance of the reference."""
    from rcvpose_amd.fcn_engine import FcnEngine

    g = np.load(os.path.join(golden_dir, "fcn2_480x640_samples.npz"))
    x = torch.from_numpy(fcn_weights.make_input(int(g["input_seed"]), (1, 3, 480, 640))).cuda()
    eng = FcnEngine([state_dict, _sd(shapes, 1), _sd(shapes, 2)], 480, 640, dtype="bf16")
    seg_a, rad_a = eng.forward(x)
    seg_a, rad_a = seg_a.clone(), rad_a.clone()
    eng.set_option(1, 0)  # register-staged kernel everywhere
    seg_b, rad_b = eng.forward(x)
    assert torch.equal(rad_a, rad_b) and torch.equal(seg_a, seg_b)  # same K order, same fp32 accumulation chain
    pix = g["pix"]
    err = np.abs(rad_a[0].cpu().numpy().reshape(-1)[pix] - g["radial"]).max() / float(g["radial_absmax"])
    print("bf16 full-size radial relative error %.3e" % err)
    assert err <= REL_BF16


def test_mfma_stem_matches_direct_stem(shapes, state_dict):
    """bf16 engine: the MFMA stem (input rounded to bf16, fp32 accumulation) against the direct stem kernel (fp32 input) at
    the stem's own output, and the effect on the network output."""
    from rcvpose_amd.fcn_engine import FcnEngine

    x = torch.from_numpy(fcn_weights.make_input(5, (2, 3, 64, 96))).cuda()
    eng = FcnEngine([state_dict, _sd(shapes, 1)], 64, 96, dtype="bf16", reuse_slots=False)
    seg_a, rad_a = [t.clone() for t in eng.forward(x)]
    stem_a = [eng.debug_stage(2, g, "bn1").clone() for g in range(2)]
    eng.set_option(2, 0)
    seg_b, rad_b = [t.clone() for t in eng.forward(x)]
    stem_b = [eng.debug_stage(2, g, "bn1").clone() for g in range(2)]
    for a, b in zip(stem_a, stem_b):
        assert a.shape == b.shape and a.shape[1] == 64
        err = (a - b).abs().max().item() / b.abs().max().item()
        print("stem relative difference %.3e" % err)
        assert err <= 2e-2  # bf16 rounding of the input (2^-9 relative per tap) and of the output
        assert (a - b).abs().mean().item() / b.abs().mean().item() <= 4e-3
    assert _rel(rad_a.cpu().numpy(), rad_b.cpu().numpy()) <= REL_BF16


def test_fused_tail_identical_to_three_kernels(shapes, state_dict):
    """upsample x2 + conv7 + head in one kernel (option 3, default on) gives bit-identical outputs to the three separate
    kernels (same arithmetic in the same order), full size and a size with partial tiles."""
    from rcvpose_amd.fcn_engine import FcnEngine

    for (H, W, B) in ((480, 640, 2), (64, 96, 3), (96, 160, 1)):
        x = torch.from_numpy(fcn_weights.make_input(11, (B, 3, H, W))).cuda()
        eng = FcnEngine([state_dict, _sd(shapes, 1)], H, W, dtype="bf16")
        seg_a, rad_a = [t.clone() for t in eng.forward(x)]
        eng.set_option(3, 0)
        seg_b, rad_b = [t.clone() for t in eng.forward(x)]
        assert torch.equal(seg_a, seg_b) and torch.equal(rad_a, rad_b), (H, W)


def test_fused_bottleneck_half_identical_to_two_kernels(shapes, state_dict):
    """3x3 (-> 256, ReLU) + 1x1 expand (-> 1024, + residual, ReLU) of the layer-3 blocks as one kernel (option 4, default on)
    gives bit-identical network outputs to the two separate convolution kernels; batch sizes with full and ragged tiles."""
    from rcvpose_amd.fcn_engine import FcnEngine

    # the fused kernel is used from 192 tiles of 128 pixels on: 7 images x 30 x 40 pixels x 3 networks = 198 (last tile ragged)
    # ... and more tiles than CUs from 9 images on (second-round tiles read inputs while first-round tiles write outputs)
    for (H, W, B) in ((480, 640, 7), (480, 640, 8), (352, 480, 14), (480, 640, 16)):
        x = torch.from_numpy(fcn_weights.make_input(13, (B, 3, H, W))).cuda()
        eng = FcnEngine([state_dict, _sd(shapes, 1), _sd(shapes, 2)], H, W, dtype="bf16")
        seg_a, rad_a = [t.clone() for t in eng.forward(x)]
        eng.set_option(6, 0)  # the 256-wide chained block with all waves in lockstep instead of the staggered schedule
        seg_d, rad_d = [t.clone() for t in eng.forward(x)]
        assert torch.equal(seg_a, seg_d) and torch.equal(rad_a, rad_d), (H, W, B)
        eng.set_option(5, 0)  # without chaining the next block's 1x1 reduce into the kernel
        seg_c, rad_c = [t.clone() for t in eng.forward(x)]
        assert torch.equal(seg_a, seg_c) and torch.equal(rad_a, rad_c), (H, W, B)
        eng.set_option(4, 0)
        seg_b, rad_b = [t.clone() for t in eng.forward(x)]
        assert torch.equal(seg_a, seg_b) and torch.equal(rad_a, rad_b), (H, W, B)


def test_halo_patch_convolution_identical_to_dma_kernel(shapes, state_dict):
    """k_conv_halo (option 7, default on: the 3x3 / stride-1 decoder layers read their A operand from a halo'd pixel patch in LDS)
    gives bit-identical network outputs to k_conv_dma; with option 7 = 2 also at image sizes whose pixel rectangles hang over
    the image edges (ragged rectangles, masked stores) and for both channel-tile widths (128 and 64 output channels)."""
    from rcvpose_amd.fcn_engine import FcnEngine

    for (H, W, B, mode) in ((480, 640, 3, 1), (480, 640, 9, 1), (416, 544, 5, 2), (352, 480, 6, 2), (64, 96, 40, 2)):
        x = torch.from_numpy(fcn_weights.make_input(17, (B, 3, H, W))).cuda()
        eng = FcnEngine([state_dict, _sd(shapes, 1), _sd(shapes, 2)], H, W, dtype="bf16")
        eng.set_option(7, 0)
        seg_a, rad_a = [t.clone() for t in eng.forward(x)]
        eng.set_option(7, mode)
        seg_b, rad_b = [t.clone() for t in eng.forward(x)]
        assert torch.equal(seg_a, seg_b) and torch.equal(rad_a, rad_b), (H, W, B, mode)


def test_batch_whose_tensors_exceed_2gib(shapes, state_dict):
    """At 112 frames the 240 x 320 x 128 activations of one network pass 2 GiB: the LDS-DMA kernels (32-bit buffer offsets) must
    step aside for those layers, and every image still gets the result it gets in a small batch."""
    from rcvpose_amd.fcn_engine import FcnEngine

    eng = FcnEngine([state_dict], 480, 640, dtype="bf16")
    small = torch.from_numpy(fcn_weights.make_input(23, (2, 3, 480, 640))).cuda()
    seg_s, rad_s = [t.clone() for t in eng.forward(small)]
    big = small.repeat(56, 1, 1, 1)  # images 0, 1, 0, 1, ...
    seg_b, rad_b = eng.forward(big)
    for i in (0, 1, 55, 110, 111):
        assert torch.equal(seg_b[0, i], seg_s[0, i % 2]) and torch.equal(rad_b[0, i], rad_s[0, i % 2]), i


def test_forward_as_hip_graph(shapes, state_dict):
    """FcnEngine.capture: the forward's launches replayed as one HIP graph give the eager result, also after the static input
    has been refilled."""
    from rcvpose_amd.fcn_engine import FcnEngine

    eng = FcnEngine([state_dict, _sd(shapes, 1)], 64, 96, dtype="bf16")
    x = torch.from_numpy(fcn_weights.make_input(21, (2, 3, 64, 96))).cuda()
    seg_a, rad_a = [t.clone() for t in eng.forward(x)]
    graph, seg_g, rad_g = eng.capture(x)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(seg_a, seg_g) and torch.equal(rad_a, rad_g)
    x2 = torch.from_numpy(fcn_weights.make_input(22, (2, 3, 64, 96))).cuda()
    seg_b, rad_b = [t.clone() for t in eng.forward(x2)]
    x.copy_(x2)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(seg_b, seg_g) and torch.equal(rad_b, rad_g)
    assert not torch.equal(seg_a, seg_b)


def test_groups_and_batch(shapes, state_dict):
    """3 networks x 2 images in one launch sequence == the six single runs (blockIdx.z / batch indexing)."""
    from rcvpose_amd.fcn_engine import FcnEngine

    sds = [state_dict, _sd(shapes, 1), _sd(shapes, 2)]
    x = torch.from_numpy(np.concatenate([fcn_weights.make_input(5, (1, 3, 64, 96)), fcn_weights.make_input(6, (1, 3, 64, 96))])).cuda()
    eng3 = FcnEngine(sds, 64, 96, dtype="fp32")
    seg3, rad3 = eng3.forward(x)
    assert seg3.shape == (3, 2, 1, 64, 96)
    for gi, sd in enumerate(sds):
        eng1 = FcnEngine([sd], 64, 96, dtype="fp32")
        for bi in range(2):
            s1, r1 = eng1.forward(x[bi:bi + 1])
            assert torch.equal(s1[0, 0], seg3[gi, bi]) and torch.equal(r1[0, 0], rad3[gi, bi]), (gi, bi)


def test_module_boundary(state_dict):
    from rcvpose_amd.models.fcnresnet import DenseFCNResNet152

    model = DenseFCNResNet152(3, 2)
    model.load_state_dict(state_dict, strict=True)
    assert len(model.state_dict()) == 1307
    model.eval()
    x = torch.from_numpy(fcn_weights.make_input(9, (1, 3, 64, 96)))
    with pytest.raises(RuntimeError):
        model(x)  # CPU input: no fallback
    model.to("cuda")
    seg, rad = model(x.cuda())
    # DataParallel-style checkpoints ("module." prefix, reference README.md:13,52) give the same engine weights
    from rcvpose_amd.fcn_engine import FcnEngine

    assert model.compute_dtype == "f16x3"  # the default arithmetic: the fast mode within the 1e-4 tolerance
    eng = FcnEngine([{"module." + k: v for k, v in state_dict.items()}], 64, 96, dtype=model.compute_dtype)
    s2, r2 = eng.forward(x.cuda())
    assert torch.equal(s2[0], seg) and torch.equal(r2[0], rad)
    model.set_compute_dtype("fp32")
    s3, r3 = model(x.cuda())
    assert float((r3 - rad).abs().max() / r3.abs().max()) <= REL_FP32 and float((s3 - seg).abs().max() / s3.abs().max()) <= REL_FP32
    model.train()
    with pytest.raises(NotImplementedError):
        model(x.cuda())


def test_preprocess_bit_exact_and_backbone(state_dict, golden_dir):
    from oracle import fcn_torch
    from PIL import Image

    from rcvpose_amd import AccumulatorSpace as A
    from rcvpose_amd.fcn_engine import preprocess_rgb
    from rcvpose_amd.models.fcnresnet import DenseFCNResNet152

    g = np.load(os.path.join(golden_dir, "fcn3_backbone_64x96.npz"))
    x = preprocess_rgb(torch.from_numpy(g["rgb"]).cuda())
    assert torch.equal(x.cpu(), fcn_torch.preprocess(g["rgb"]))  # float64 arithmetic of the reference, bit for bit
    model = DenseFCNResNet152(3, 2)
    model.load_state_dict(state_dict, strict=True)
    model.eval()
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "frame.png")
        Image.fromarray(g["rgb"]).save(p)
        sem, radial = A.FCResBackbone(model, p, [])
    assert sem.shape == (64, 96) and sem.dtype == np.float32
    assert _rel(radial, g["radial"]) <= REL_FP32 and _rel(sem, g["sem"]) <= REL_FP32


def test_resfcn_model_vs_golden(golden_dir):
    """ResFCNResNet152 (additive skips) through the model class: fp32 within 1e-4 of the reference's
    output and stages, bf16 within the bf16 tolerance."""
    import json

    from rcvpose_amd.fcn_engine import FcnEngine
    from rcvpose_amd.models.fcnresnet import ResFCNResNet152

    g = np.load(os.path.join(golden_dir, "resfcn1_64x96.npz"))
    shapes = json.load(open(os.path.join(golden_dir, "resfcn_state_dict_keys.json")))
    sd = {k: torch.from_numpy(v) for k, v in fcn_weights.make_state_dict_numpy(shapes, seed=int(g["weight_seed"])).items()}
    x = torch.from_numpy(fcn_weights.make_input(int(g["input_seed"]), (1, 3, 64, 96))).cuda()
    model = ResFCNResNet152(3, 2)
    model.load_state_dict(sd, strict=True)
    model.eval().to("cuda")
    seg, rad = model(x)
    assert seg.shape == (1, 1, 64, 96)
    assert _rel(rad.cpu().numpy(), g["radial"]) <= REL_FP32 and _rel(seg.cpu().numpy(), g["seg"]) <= REL_FP32
    eng = FcnEngine([sd], 64, 96, dtype="fp32", reuse_slots=False, arch="res")
    eng.forward(x)
    for k in g.files:
        if k.startswith("stage_") and k[6:] in eng.stages:
            got = eng.debug_stage(1, 0, k[6:]).cpu().numpy()
            assert _rel(got, g[k]) <= REL_FP32, k
    model.set_compute_dtype("bf16")
    seg_b, rad_b = model(x)
    er, es = _rel(rad_b.cpu().numpy(), g["radial"]), _rel(seg_b.cpu().numpy(), g["seg"])
    print("resfcn bf16 relative error: radial %.3e seg %.3e" % (er, es))
    assert er <= REL_BF16 and es <= REL_BF16


def test_batch_size_may_grow_between_calls(shapes, state_dict):
    """One engine, first a small batch (register-staged kernels only), then a large one (LDS-DMA and fused kernels with
    > 64 KB of dynamic LDS): the large-LDS kernels are registered at engine creation, not by the first forward."""
    from rcvpose_amd.fcn_engine import FcnEngine

    eng = FcnEngine([state_dict, _sd(shapes, 1), _sd(shapes, 2)], 480, 640, dtype="bf16")
    x = torch.from_numpy(fcn_weights.make_input(17, (9, 3, 480, 640))).cuda()
    seg1, rad1 = [t.clone() for t in eng.forward(x[:1])]
    seg9, rad9 = [t.clone() for t in eng.forward(x)]
    assert torch.equal(seg9[:, :1], seg1) and torch.equal(rad9[:, :1], rad1)


# The bf16 engine against its emulation (oracle/fcn_torch: the same graph, rounded to bf16 exactly where the engine rounds,
# fp32 accumulation).  What is left between the two is the fp32 summation order (MFMA blocks vs torch's loops), ~1e-6..1e-5
# relative, which now and then tips a stored value to the neighbouring bf16 number.  Fed with IDENTICAL inputs, one layer
# therefore agrees except for a small fraction of elements that are one bf16 step apart.  (Through the whole network those
# flips multiply until the two are independent realisations of the rounding noise -- see oracle/fcn_torch.py -- so the
# whole-network comparison cannot be tighter than bf16-vs-fp32 itself; the per-layer comparison is the tight one.)
EMU_STEP = 2.0 ** -7   # one bf16 step of a value in the top binade, relative to the tensor's max |value|
EMU_MISMATCH = 0.02    # fraction of a layer's outputs that may differ at all (measured <= ~3e-3)
EMU_HEAD = 2e-5        # the fp32 head output on identical conv7 input: summation order only


def test_bf16_engine_layer_by_layer_against_bf16_emulation(state_dict):
    """480x640, every op of the layer program (stem, max-pool, 161 convolutions, 5 upsamplings, head): the bf16 engine's
    stored output against the bf16-emulating torch oracle applied to the engine's own stored inputs of that op
    (VERDICT r01 #1a: the tight link of the timed precision to the oracle).  The fused kernels of the default path are
    bit-identical to the per-op kernels used here (test_fused_* / test_halo_* / test_bf16_full_size_both_conv_kernels)."""
    from oracle import fcn_torch
    from rcvpose_amd import _hip, fcn_engine
    from rcvpose_amd.fcn_engine import FcnEngine

    H, W = 480, 640
    sd = state_dict
    x = torch.from_numpy(fcn_weights.make_input(11, (1, 3, H, W)))
    prog = fcn_engine.build_program(H, W)
    eng = FcnEngine([sd], H, W, dtype="bf16", reuse_slots=False)   # one slot per logical tensor
    eng.set_option(3, 0)   # no fused tail: up1 / conv7 are materialised
    eng.set_option(4, 0)   # no fused bottleneck halves: every intermediate tensor is materialised
    seg, rad = eng.forward(x.cuda())

    def stored(t):
        return eng.debug_slot(1, 0, t, prog.shapes[t]).permute(0, 3, 1, 2).contiguous().cpu()   # NCHW fp32 (bf16 values)

    worst_frac, worst_rel, n_conv = 0.0, 0.0, 0
    for op, spec in zip(prog.ops, prog.wspecs):
        kind = op["kind"]
        if kind == _hip.OP_STEM:
            ref = fcn_torch.conv_bf16(sd, x.to(torch.bfloat16).to(torch.float32), spec["conv"], spec["bn"], False, stride=2, padding=3)
        elif kind == _hip.OP_MAXPOOL:
            ref = torch.nn.functional.max_pool2d(stored(op["src0"]), kernel_size=3, stride=2, padding=1)
        elif kind == _hip.OP_UPSAMPLE2:
            ref = fcn_torch.upsample2_bf16(stored(op["src0"]))
        elif kind == _hip.OP_CONV:
            src = stored(op["src0"])
            if op["src1"] >= 0:
                src = torch.cat((src, stored(op["src1"])), 1)
            ref = fcn_torch.conv_bf16(sd, src, spec["conv"], spec["bn"], spec["bias"], stride=op["stride"], padding=op["pad"],
                                      relu=bool(op["relu"]), res=stored(op["res"]) if op["res"] >= 0 else None)
            n_conv += 1
        else:  # head: fp32 planes
            ref = fcn_torch.head_bf16(sd, stored(op["src0"]))
            got = torch.cat((seg[0].cpu(), rad[0].cpu()), 1)
            assert float((got - ref).abs().max() / ref.abs().max()) <= EMU_HEAD
            continue
        got = stored(op["dst"])
        assert got.shape == ref.shape
        rel = float((got - ref).abs().max() / ref.abs().max())
        frac = float((got != ref).float().mean())
        if kind == _hip.OP_MAXPOOL:
            assert rel == 0.0   # exact
        worst_frac, worst_rel = max(worst_frac, frac), max(worst_rel, rel)
        assert rel <= EMU_STEP and frac <= EMU_MISMATCH, (spec and spec["conv"], kind, rel, frac)
    assert n_conv == 161
    print("bf16 engine vs emulation, layer by layer: worst mismatch fraction %.2e, worst max-rel %.2e" % (worst_frac, worst_rel))


def test_bf16_engine_whole_network_against_bf16_emulation(state_dict):
    """The whole forward against the emulation, for the record: both are bf16 realisations of the same graph, so they sit
    within the bf16 tolerance of each other (and of the fp32 golden), not closer (see the note above)."""
    from oracle import fcn_torch
    from rcvpose_amd.fcn_engine import FcnEngine

    x = torch.from_numpy(fcn_weights.make_input(11, (1, 3, 480, 640)))
    s_ref, r_ref = fcn_torch.forward_bf16(state_dict, x)
    s32, r32 = fcn_torch.forward(state_dict, x)
    seg, rad = FcnEngine([state_dict], 480, 640, dtype="bf16").forward(x.cuda())
    e_emu, e_32 = _rel(rad[0].cpu().numpy(), r_ref.numpy()), _rel(rad[0].cpu().numpy(), r32.numpy())
    o_32 = _rel(r_ref.numpy(), r32.numpy())
    print("radial: engine vs emulation %.3e, engine vs fp32 %.3e, emulation vs fp32 %.3e" % (e_emu, e_32, o_32))
    assert e_emu <= REL_BF16 and e_32 <= REL_BF16
    assert e_32 <= 2.0 * o_32   # the engine is no worse a bf16 implementation than the straightforward one


def test_radial_plane_as_float16(shapes, state_dict):
    """Engine option 8 (BASELINE.json stress configuration: radii maps stored as fp16): the radial plane equals the float32
    plane rounded to nearest-even float16, from both head kernels (fused tail, separate head) and both engines."""
    from rcvpose_amd.fcn_engine import FcnEngine

    x = torch.from_numpy(fcn_weights.make_input(13, (2, 3, 64, 96))).cuda()
    for dtype in ("bf16", "fp32"):
        for tail in ((1, 0) if dtype == "bf16" else (1,)):
            eng = FcnEngine([state_dict, _sd(shapes, 1)], 64, 96, dtype=dtype)
            eng.set_option(3, tail)
            seg32, rad32 = [t.clone() for t in eng.forward(x)]
            eng.set_option(8, 1)
            seg16, rad16 = eng.forward(x)
            assert rad16.dtype == torch.float16 and torch.equal(seg16, seg32)
            assert torch.equal(rad16, rad32.to(torch.float16)), (dtype, tail)
            with pytest.raises(ValueError):
                eng.forward(x, out=(seg32, rad32))  # float32 radial buffer under option 8


def test_latency_kernel_identical_to_register_staged_kernel(shapes, state_dict):
    """k_conv_lat (engine option 9: small tiles, eight LDS-DMA stages in flight, used when a launch has few output tiles)
    against k_conv on every layer it takes over: bit-identical seg / radial maps and stages, at batch 1 / one network
    (BASELINE.json configs[1]), batch 2 / three networks, and a small image whose tiles are partial."""
    from rcvpose_amd.fcn_engine import FcnEngine

    for (H, W, B, G) in ((480, 640, 1, 1), (480, 640, 2, 3), (64, 96, 3, 2), (96, 160, 1, 3)):
        sds = [state_dict, _sd(shapes, 1), _sd(shapes, 2)][:G]
        eng = FcnEngine(sds, H, W, dtype="bf16", reuse_slots=False)
        x = torch.from_numpy(fcn_weights.make_input(23, (B, 3, H, W))).cuda()
        seg_a, rad_a = [t.clone() for t in eng.forward(x)]
        stages_a = {n: eng.debug_stage(B, G - 1, n).clone() for n in ("block1", "block3", "block4", "bn6", "conv_up5", "conv_up3")}
        eng.set_option(9, 0)
        seg_b, rad_b = eng.forward(x)
        assert torch.equal(seg_a, seg_b) and torch.equal(rad_a, rad_b), (H, W, B, G)
        for n, t in stages_a.items():
            assert torch.equal(t, eng.debug_stage(B, G - 1, n)), (n, H, W, B, G)


def test_fast_kernels_against_the_plain_kernel_over_sizes_and_batches(shapes, state_dict):
    """The LDS-DMA, halo-patch, fused-block, fused-tail and latency kernels (all on by default) against the register-staged
    k_conv alone (options 1, 3, 4, 7, 9 off) on image sizes whose tiles are ragged in every layer, batches from one frame
    to dozens and one to three networks: bit-identical seg / radial maps."""
    from rcvpose_amd.fcn_engine import FcnEngine

    sds = [state_dict, _sd(shapes, 1), _sd(shapes, 2)]
    for (H, W, B, G) in ((32, 32, 1, 1), (32, 64, 7, 2), (96, 32, 3, 3), (64, 96, 33, 1), (160, 96, 5, 2), (224, 288, 2, 3),
                         (480, 640, 1, 3), (480, 640, 2, 1), (256, 640, 11, 2)):
        x = torch.from_numpy(fcn_weights.make_input(31 + B, (B, 3, H, W))).cuda()
        eng = FcnEngine(sds[:G], H, W, dtype="bf16")
        seg_a, rad_a = [t.clone() for t in eng.forward(x)]
        for key in (1, 3, 4, 7, 9):
            eng.set_option(key, 0)
        seg_b, rad_b = eng.forward(x)
        assert torch.equal(seg_a, seg_b) and torch.equal(rad_a, rad_b), (H, W, B, G)
        assert bool(torch.isfinite(seg_a).all()) and bool(torch.isfinite(rad_a).all())
