"""The engine's fast parity mode RCV_F16X3 ("f16x3"): activations and weights as IEEE-half hi / lo pairs, three MFMAs per
product (hi*hi + hi*lo + lo*hi), fp32 accumulation.  It carries north_star's 1e-4 gate on the radial maps like the fp32
engine (reference models/fcnresnet.py:121-191 is an fp32 forward) at a third of the fp16 matrix rate instead of a sixteenth."""
import json
import os

import numpy as np
import pytest
import torch

import fcn_weights

pytestmark = pytest.mark.gpu

REL = 1e-4        # north star: radial maps within 1e-4 relative of the fp32 reference
REL_TIGHT = 3e-5  # regression guard: measured ~5e-6 (as the fp32 engine)


@pytest.fixture(scope="module")
def shapes(golden_dir):
    return json.load(open(os.path.join(golden_dir, "fcn_state_dict_keys.json")))


def _sd(shapes, seed=0):
    return {k: torch.from_numpy(v) for k, v in fcn_weights.make_state_dict_numpy(shapes, seed=seed).items()}


def _rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


def test_x3_stages_and_output_vs_golden(shapes, golden_dir):
    """Every named stage of the reference module's golden forward (64x96) and the output planes, <= 1e-4."""
    from rcvpose_amd.fcn_engine import FcnEngine

    g = np.load(os.path.join(golden_dir, "fcn1_64x96.npz"))
    x = torch.from_numpy(fcn_weights.make_input(int(g["input_seed"]), (1, 3, 64, 96))).cuda()
    eng = FcnEngine([_sd(shapes)], 64, 96, dtype="f16x3", reuse_slots=False)
    for key in (3, 4):  # the fused kernels never write the tensors between their layers: every stage through its own kernel
        eng.set_option(key, 0)
    seg, rad = eng.forward(x)
    worst = 0.0
    for k in g.files:
        if not k.startswith("stage_") or k[6:] == "conv8":
            continue
        ref = g[k]
        if k[6:] in ("bn1", "bn6"):
            ref = np.maximum(ref, 0)
        got = eng.debug_stage(1, 0, k[6:]).cpu().numpy()
        assert got.shape == ref.shape, k
        worst = max(worst, _rel(got, ref))
        assert _rel(got, ref) <= REL, (k, _rel(got, ref))
    er, es = _rel(rad[0].cpu().numpy(), g["radial"]), _rel(seg[0].cpu().numpy(), g["seg"])
    print("f16x3 64x96: worst stage %.2e radial %.2e seg %.2e" % (worst, er, es))
    assert er <= REL_TIGHT and es <= REL_TIGHT


def test_x3_full_size_vs_golden_samples(shapes, golden_dir):
    """480x640, three networks in one engine: 4096 sampled pixels of the reference's golden forward."""
    from rcvpose_amd.fcn_engine import FcnEngine

    g = np.load(os.path.join(golden_dir, "fcn2_480x640_samples.npz"))
    x = torch.from_numpy(fcn_weights.make_input(int(g["input_seed"]), (1, 3, 480, 640))).cuda()
    seg, rad = FcnEngine([_sd(shapes), _sd(shapes, 1), _sd(shapes, 2)], 480, 640, dtype="f16x3").forward(x)
    pix = g["pix"]
    err_r = np.abs(rad[0].cpu().numpy().reshape(-1)[pix] - g["radial"]).max() / float(g["radial_absmax"])
    err_s = np.abs(seg[0].cpu().numpy().reshape(-1)[pix] - g["seg"]).max() / float(g["seg_absmax"])
    print("f16x3 480x640 (sampled): radial %.2e seg %.2e" % (err_r, err_s))
    assert err_r <= REL_TIGHT and err_s <= REL_TIGHT


def test_x3_against_fp32_engine_whole_maps(shapes):
    """Whole 480x640 maps of three networks and four frames against the fp32 engine (v_mfma_f32_16x16x4_f32)."""
    from rcvpose_amd.fcn_engine import FcnEngine

    sds = [_sd(shapes, s) for s in range(3)]
    x = torch.from_numpy(fcn_weights.make_input(17, (4, 3, 480, 640))).cuda()
    seg_a, rad_a = FcnEngine(sds, 480, 640, dtype="f16x3").forward(x)
    seg_b, rad_b = FcnEngine(sds, 480, 640, dtype="fp32").forward(x)
    for g in range(3):
        er = float((rad_a[g] - rad_b[g]).abs().max() / rad_b[g].abs().max())
        es = float((seg_a[g] - seg_b[g]).abs().max() / seg_b[g].abs().max())
        print("f16x3 vs fp32 engine, network %d: radial %.2e seg %.2e" % (g, er, es))
        assert er <= REL_TIGHT and es <= REL_TIGHT


def test_x3_kernel_variants_identical(shapes):
    """The fast kernels of the split engine (engine options on) against the plain register-staged kernels (all off): the
    same K order and MFMA sequence per output element, so bit-identical maps."""
    from rcvpose_amd.fcn_engine import FcnEngine

    sds = [_sd(shapes, s) for s in range(3)]
    # 7 frames: the smallest batch at which every fused kernel is active at 480x640 -- k_chain_x3<256> needs 192 tiles of 128
    # pixels (7 x 1200 px x 3 networks = 197, the last tile of each network partial: 80 of 128 rows)
    for (H, W, B) in ((480, 640, 7), (480, 640, 1), (96, 160, 3)):
        eng = FcnEngine(sds, H, W, dtype="f16x3")
        x = torch.from_numpy(fcn_weights.make_input(31, (B, 3, H, W))).cuda()
        seg_d, rad_d = [t.clone() for t in eng.forward(x)]   # the default path: the tail parity-composed at 480x640 (option 13)
        eng.set_option(13, 0)                                 # ... which is another summation: not of the bit-identical family
        seg_a, rad_a = [t.clone() for t in eng.forward(x)]
        assert (H < 480) == bool(torch.equal(rad_d, rad_a))   # (small images keep k_tail_x3 by the rule of option 13)
        assert float((rad_d - rad_a).abs().max() / rad_a.abs().max()) <= 5e-6 and float((seg_d - seg_a).abs().max() / seg_a.abs().max()) <= 5e-6
        for form in (2, 0):  # the chain kernel as k_chain3_x3 (one wave per SIMD, 256-wide block) and in its round-3 form (k_chain_x3)
            eng.set_option(10, form)
            seg_c, rad_c = [t.clone() for t in eng.forward(x)]
            assert torch.equal(seg_a, seg_c) and torch.equal(rad_a, rad_c), (H, W, B, form)
        for key in (1, 3, 4, 5, 6, 7, 9):
            eng.set_option(key, 0)
        seg_b, rad_b = eng.forward(x)
        assert torch.equal(seg_a, seg_b) and torch.equal(rad_a, rad_b), (H, W, B)
        assert eng.range_status() == 0


def test_x3_composed_tail_in_image_slices(shapes, monkeypatch):
    """The composed tail addresses a group's tensor with 32-bit DMA offsets; a batch beyond that range (> ~108 frames at 480x640) runs
    in slices of images instead of falling back to another kernel, so a frame's bits do not depend on the batch.  The slice size is
    forced down to two images here (RCV_TAILC_MAX_IMAGES): five frames = slices of 2 + 2 + 1, identical maps."""
    from rcvpose_amd.fcn_engine import FcnEngine

    sds = [_sd(shapes, s) for s in range(3)]
    H, W, B = 96, 160, 5
    eng = FcnEngine(sds, H, W, dtype="f16x3")
    eng.set_option(13, 2)
    x = torch.from_numpy(fcn_weights.make_input(43, (B, 3, H, W))).cuda()
    seg0, rad0 = [t.clone() for t in eng.forward(x)]
    monkeypatch.setenv("RCV_TAILC_MAX_IMAGES", "2")
    seg1, rad1 = [t.clone() for t in eng.forward(x)]
    monkeypatch.delenv("RCV_TAILC_MAX_IMAGES")
    assert torch.equal(seg0, seg1) and torch.equal(rad0, rad1)
    eng.set_option(13, 0)
    seg2, rad2 = eng.forward(x)
    assert not torch.equal(rad2, rad0)   # (the composed form was really the one that ran)


def test_x3_trunk_experiments_of_round_6(shapes):
    """Two measured experiments that stay in the tree switched off.  Engine option 15 (the 128-wide chain kernel as four-wave
    workgroups, three per CU): the same arithmetic, bit-identical maps.  Engine option 14 (the trunk between two chain kernels of layer 2
    in a 3-byte form: 19 instead of 22 significant bits): another function -- maps within 2e-5 of the default path and within 1e-4 of
    the fp32 engine."""
    from rcvpose_amd.fcn_engine import FcnEngine

    sds = [_sd(shapes, s) for s in range(3)]
    H, W, B = 480, 640, 7
    eng = FcnEngine(sds, H, W, dtype="f16x3")
    x = torch.from_numpy(fcn_weights.make_input(53, (B, 3, H, W))).cuda()
    seg0, rad0 = [t.clone() for t in eng.forward(x)]
    eng.set_option(15, 1)
    seg1, rad1 = [t.clone() for t in eng.forward(x)]
    assert torch.equal(seg0, seg1) and torch.equal(rad0, rad1)
    eng.set_option(15, 0)
    eng.set_option(14, 1)
    seg2, rad2 = [t.clone() for t in eng.forward(x)]
    d = float((rad2 - rad0).abs().max() / rad0.abs().max())
    print("3-byte trunk in layer 2: radial maps %.2e from the default path" % d)
    assert 0.0 < d <= 2e-5 and eng.range_status() == 0
    seg_f, rad_f = FcnEngine(sds, H, W, dtype="fp32").forward(x[:2])
    for g in range(3):
        assert float((rad2[g, :2] - rad_f[g]).abs().max() / rad_f[g].abs().max()) <= REL_TIGHT
    eng.set_option(14, 0)
    seg3, rad3 = eng.forward(x)
    assert torch.equal(rad3, rad0) and torch.equal(seg3, seg0)


def test_x3_mfma_stem_against_direct_stem(shapes):
    """The MFMA stem (image split into half hi / lo pairs, three MFMAs per product) against the direct fp32 stem kernel, at the
    stem's own output and at the network output: both are fp32-grade, not bit-identical (different summation)."""
    from rcvpose_amd.fcn_engine import FcnEngine

    x = torch.from_numpy(fcn_weights.make_input(5, (2, 3, 64, 96))).cuda()
    outs = []
    for mfma in (1, 0):
        eng = FcnEngine([_sd(shapes), _sd(shapes, 1)], 64, 96, dtype="f16x3", reuse_slots=False)
        eng.set_option(2, mfma)
        seg, rad = eng.forward(x)
        outs.append((eng.debug_stage(2, 1, "bn1").cpu().numpy(), rad.cpu().numpy()))
    e_stem, e_out = _rel(outs[0][0], outs[1][0]), _rel(outs[0][1], outs[1][1])
    print("f16x3 MFMA stem vs direct stem: stem output %.2e, radial maps %.2e" % (e_stem, e_out))
    assert e_stem <= 2e-6 and e_out <= REL_TIGHT


def test_x3_groups_and_batch(shapes):
    """A frame's maps do not depend on the batch or the group slot it travels in."""
    from rcvpose_amd.fcn_engine import FcnEngine

    sds = [_sd(shapes, s) for s in range(3)]
    x = torch.from_numpy(fcn_weights.make_input(23, (3, 3, 64, 96))).cuda()
    seg3, rad3 = FcnEngine(sds, 64, 96, dtype="f16x3").forward(x)
    for g in range(3):
        seg1, rad1 = FcnEngine([sds[g]], 64, 96, dtype="f16x3").forward(x[1:2])
        assert torch.equal(rad1[0, 0], rad3[g, 1]) and torch.equal(seg1[0, 0], seg3[g, 1])


def test_half_precision_range_is_guarded(shapes):
    """Half-precision storage has a range: f16x3 saturates (hi and lo clamp at 65504) instead of producing inf - inf = NaN, and
    the module boundary (`check_finite`) refuses non-finite maps of the plain fp16 engine instead of handing them to the seam."""
    from rcvpose_amd.fcn_engine import FcnEngine

    sd = _sd(shapes)
    x = torch.from_numpy(fcn_weights.make_input(3, (1, 3, 64, 96))).cuda() * 3.0e5   # activations far beyond 65504
    seg, rad = FcnEngine([sd], 64, 96, dtype="f16x3").forward(x)
    assert bool(torch.isfinite(seg).all()) and bool(torch.isfinite(rad).all())
    # (a ReLU turns NaN into 0, so an overflow inside the network does not always reach the maps; what does reach them is refused)
    bad = dict(sd)
    bad["conv8.bias"] = torch.tensor([float("inf"), 0.0])
    eng = FcnEngine([bad], 64, 96, dtype="fp16")
    eng.check_finite = True
    with pytest.raises(FloatingPointError):
        eng.forward(x / 3.0e5)
    eng = FcnEngine([sd], 64, 96, dtype="fp16")
    eng.check_finite = True
    seg, rad = eng.forward(x / 3.0e5)   # an ordinary image passes
    assert bool(torch.isfinite(rad).all())


def test_range_guard_reports_clamped_activations(shapes):
    """An activation beyond the half range is clamped by the f16x3 store (finite and wrong): the engine's sticky flag word
    records it, rcv_fcn_range_status returns RCV_E_RANGE, the next rcv_fcn_forward refuses to run until the word is cleared,
    and the module boundary raises.  The reference's forward is fp32 (models/fcnresnet.py:121-191) and would not clip."""
    from rcvpose_amd import _hip
    from rcvpose_amd.fcn_engine import FcnEngine
    from rcvpose_amd.models.fcnresnet import DenseFCNResNet152

    sd = _sd(shapes)
    x = torch.from_numpy(fcn_weights.make_input(3, (1, 3, 64, 96))).cuda()
    eng = FcnEngine([sd], 64, 96, dtype="f16x3")
    seg0, rad0 = eng.forward(x)
    assert eng.range_status(clear=False) == 0                       # an ordinary image: nothing left the range
    eng.forward(x * 3.0e5)                                          # activations far beyond 65504
    torch.cuda.synchronize()
    with pytest.raises(_hip.HalfRangeError):                        # sticky: the forward after a flagged one is refused
        eng.forward(x)
    assert eng.range_status(clear=False) & 1
    with pytest.raises(FloatingPointError):
        eng.range_status(clear=True, raise_on_flag=True)
    seg1, rad1 = eng.forward(x)                                     # cleared: runs again, same maps as before
    assert eng.range_status() == 0 and torch.equal(rad1, rad0) and torch.equal(seg1, seg0)
    # the plain fp16 engine overflows to infinity instead: bit 1
    e16 = FcnEngine([sd], 64, 96, dtype="fp16")
    e16.forward(x * 3.0e5)
    assert e16.range_status() & 2
    # fp32 / bf16 engines have fp32's exponent range: never flagged
    e32 = FcnEngine([sd], 64, 96, dtype="fp32")
    e32.forward(x * 3.0e5)
    assert e32.range_status() == 0
    # module boundary (the per-call surface of the reference): raises instead of returning clipped maps
    model = DenseFCNResNet152().eval()
    model.load_state_dict(sd, strict=True)
    model.cuda()
    with pytest.raises(FloatingPointError):
        model(x * 3.0e5)
    seg_m, rad_m = model(x)
    assert torch.equal(rad_m, rad0[0])


def _max_activation(sd, x):
    """Largest magnitude over the named stages of the fp32 engine's forward (a lower bound of the largest stored value)."""
    from rcvpose_amd.fcn_engine import FcnEngine

    eng = FcnEngine([sd], x.shape[2], x.shape[3], dtype="fp32", reuse_slots=False)
    seg, rad = eng.forward(x)
    return max(float(eng.debug_stage(1, 0, name).abs().max()) for name in eng.stages), seg, rad


def test_x3_accuracy_with_large_activations(shapes):
    """f16x3 stores activations unscaled: accuracy must hold up to the edge of the half range.  The input is scaled until
    the first store clamps; one step below that (largest activation within a factor 1.5 of 65504) and at ~1e3 the maps are
    compared with the fp32 engine: <= 1e-4, no flag."""
    from rcvpose_amd.fcn_engine import FcnEngine

    sd = _sd(shapes)
    x = torch.from_numpy(fcn_weights.make_input(3, (1, 3, 64, 96))).cuda()
    eng = FcnEngine([sd], 64, 96, dtype="f16x3")
    s, last_ok = 1.0, None
    while s < 1e9:
        eng.forward(x * s)
        if eng.range_status(clear=True):
            break
        last_ok = s
        s *= 1.5
    assert last_ok is not None and s < 1e9, "the guard never fired"
    m1, _, _ = _max_activation(sd, x)
    for scale in (last_ok, max(1.0, 2.0e3 / m1)):
        mx, seg_r, rad_r = _max_activation(sd, x * scale)
        seg, rad = eng.forward(x * scale)
        assert eng.range_status() == 0
        er = float((rad[0] - rad_r[0]).abs().max() / rad_r[0].abs().max())
        es = float((seg[0] - seg_r[0]).abs().max() / seg_r[0].abs().max())
        print("f16x3 at input scale %.3g: largest named-stage activation %.3g, radial %.2e seg %.2e vs fp32 engine" % (scale, mx, er, es))
        assert 1.0e3 <= mx <= 65504.0 * 1.01
        assert er <= REL and es <= REL
    assert _max_activation(sd, x * last_ok)[0] >= 65504.0 / 1.5 / 1.6   # (named stages only: a lower bound)


def test_x3_hostile_batchnorm_statistics_vs_oracle(shapes):
    """A state dict with hostile BatchNorm statistics (running_var 1e-6 ... 1e2, gamma spread 1e-2 ... 1e2 per channel inside the
    bottlenecks; tests/fcn_weights.make_hostile_state_dict -- the same function in exact arithmetic) through f16x3 against
    the torch oracle evaluated in float64: <= 1e-4 at the output, no range flag.  Folded weights then span ~10 orders of
    magnitude within a layer (the per-channel power-of-two scale absorbs that) and the `mid` activations four per tensor."""
    from oracle import fcn_torch
    from rcvpose_amd.fcn_engine import FcnEngine

    sd = fcn_weights.make_hostile_state_dict(_sd(shapes), seed=1)
    x = torch.from_numpy(fcn_weights.make_input(3, (1, 3, 64, 96)))
    sd64 = {k: (v.double() if v.dtype == torch.float32 else v) for k, v in sd.items()}
    seg_r, rad_r = fcn_torch.forward(sd64, x.double())
    out = {}
    for dtype in ("f16x3", "fp32"):
        eng = FcnEngine([sd], 64, 96, dtype=dtype)
        seg, rad = eng.forward(x.cuda())
        assert eng.range_status() == 0
        out[dtype] = (float((rad[0].cpu().double() - rad_r).abs().max() / rad_r.abs().max()),
                      float((seg[0].cpu().double() - seg_r).abs().max() / seg_r.abs().max()))
    print("hostile BN statistics, vs float64 oracle: f16x3 radial %.2e seg %.2e; fp32 engine radial %.2e seg %.2e" % (out["f16x3"] + out["fp32"]))
    assert max(out["f16x3"]) <= REL and max(out["fp32"]) <= REL


def test_x3_engine_refuses_channel_counts_it_cannot_store():
    """Split storage keeps channels in groups of [32 hi | 32 lo]: rcv_fcn_create returns RCV_E_UNSUPPORTED for a program whose
    tensors are not multiples of 32 channels instead of mis-indexing them (a C-ABI caller does not go through pack_split)."""
    import ctypes

    from rcvpose_amd import _hip

    L = _hip.lib()
    op = _hip.Op(kind=_hip.OP_CONV, src0=0, src1=-1, res=-1, dst=1, c0=48, c1=0, cout=64, hin=8, win=8, hout=8, wout=8, ksize=1,
                 stride=1, pad=0, relu=1, w_off=0, b_off=0)
    arr = (_hip.Op * 1)(op)
    slots = (ctypes.c_int64 * 2)(8 * 8 * 48, 8 * 8 * 64)
    w = torch.zeros(64 * 48 * 2, dtype=torch.float16, device="cuda")
    b = torch.zeros(128, dtype=torch.float32, device="cuda")
    h = ctypes.c_void_p()
    assert L.rcv_fcn_create(arr, 1, 2, slots, 1, _hip.RCV_F16X3, _hip.ptr(w), _hip.ptr(b), ctypes.byref(h)) == _hip.RCV_E_UNSUPPORTED
    arr[0].c0, arr[0].cout = 64, 48
    assert L.rcv_fcn_create(arr, 1, 2, slots, 1, _hip.RCV_F16X3, _hip.ptr(w), _hip.ptr(b), ctypes.byref(h)) == _hip.RCV_E_UNSUPPORTED
    arr[0].cout = 64
    assert L.rcv_fcn_create(arr, 1, 2, slots, 1, _hip.RCV_F16X3, _hip.ptr(w), _hip.ptr(b), ctypes.byref(h)) == _hip.RCV_OK
    L.rcv_fcn_destroy(h)


def test_latency_option_split_k_within_tolerance_and_repeatable(shapes, golden_dir):
    """Engine option 12 (opt-in, one frame in flight -- the reference's call pattern, AccumulatorSpace.py:596): k_conv_lat deals the
    K steps of launches with few tiles over several workgroups and k_splitk_finish adds the partial sums in a fixed order.  Not
    the bits of the unsplit kernels (another summation order), so: the golden forward within tolerance, run-to-run identical
    maps, and close to the default path."""
    from rcvpose_amd.fcn_engine import FcnEngine

    g = np.load(os.path.join(golden_dir, "fcn2_480x640_samples.npz"))
    x = torch.from_numpy(fcn_weights.make_input(int(g["input_seed"]), (1, 3, 480, 640))).cuda()
    pix = g["pix"]
    for dtype, tol in (("f16x3", REL_TIGHT), ("bf16", 6e-2)):
        eng = FcnEngine([_sd(shapes)], 480, 640, dtype=dtype)
        seg0, rad0 = [t.clone() for t in eng.forward(x)]
        eng.set_option(12, 1)
        seg1, rad1 = [t.clone() for t in eng.forward(x)]
        seg2, rad2 = eng.forward(x)
        assert torch.equal(seg1, seg2) and torch.equal(rad1, rad2)          # deterministic
        assert not torch.equal(rad0, rad1)                                  # ... and really another summation order
        err_r = np.abs(rad1[0].cpu().numpy().reshape(-1)[pix] - g["radial"]).max() / float(g["radial_absmax"])
        err_s = np.abs(seg1[0].cpu().numpy().reshape(-1)[pix] - g["seg"]).max() / float(g["seg_absmax"])
        d = float((rad1 - rad0).abs().max() / rad0.abs().max())
        print("%s, split-K latency option at batch 1: radial %.2e seg %.2e vs golden; %.2e from the default path" % (dtype, err_r, err_s, d))
        assert err_r <= tol and err_s <= tol
        if dtype == "f16x3":
            assert d <= 1e-5 and eng.range_status() == 0
        eng.set_option(12, 0)
        seg3, rad3 = eng.forward(x)
        assert torch.equal(rad3, rad0) and torch.equal(seg3, seg0)


def test_x3_composed_tail_against_the_three_kernel_tail(shapes):
    """Engine option 13: upsampling + conv7 + head as one parity-composed 3x3 convolution on the low-resolution tensor (+ the
    border tiles recomputed by k_tail_x3 in its ring_only mode) against k_tail_x3 alone (option 13 = 0) and against the three
    separate kernels (option 3 = 0): the same function with another summation order -- whole maps within 5e-6; the outermost ring of
    pixels, where the upsampling's clamp meets the convolution's zero padding, carries k_tail_x3's bits; the float16 radial plane
    (option 8) likewise."""
    from rcvpose_amd.fcn_engine import FcnEngine

    sds = [_sd(shapes, s) for s in range(3)]
    # (at 480 x 640 with two frames the persistent workgroups of the composed kernel walk five or six items each; the small sizes run one item
    # per workgroup)
    for (H, W, B) in ((480, 640, 2), (96, 160, 5), (160, 288, 3)):
        eng = FcnEngine(sds, H, W, dtype="f16x3")
        assert any(o["wq_off"] for o in eng.ops if o["cout"] == 32)
        x = torch.from_numpy(fcn_weights.make_input(41, (B, 3, H, W))).cuda()
        eng.set_option(13, 2)   # (2: also on images with fewer than 64 rectangles, where the default rule keeps k_tail_x3)
        seg_a, rad_a = [t.clone() for t in eng.forward(x)]
        eng.set_option(13, 0)
        seg_b, rad_b = [t.clone() for t in eng.forward(x)]
        eng.set_option(3, 0)
        seg_c, rad_c = [t.clone() for t in eng.forward(x)]
        assert torch.equal(seg_b, seg_c) and torch.equal(rad_b, rad_c)      # (the bit-identical family)
        for g in range(3):
            er = float((rad_a[g] - rad_b[g]).abs().max() / rad_b[g].abs().max())
            es = float((seg_a[g] - seg_b[g]).abs().max() / seg_b[g].abs().max())
            ring = torch.ones_like(rad_b[g], dtype=torch.bool)
            ring[..., 1:-1, 1:-1] = False
            err_ring = float((rad_a[g] - rad_b[g])[ring].abs().max() / rad_b[g].abs().max())
            print("%dx%d network %d: composed tail vs k_tail_x3: radial %.2e (ring %.2e) seg %.2e" % (H, W, g, er, err_ring, es))
            assert er <= 5e-6 and es <= 5e-6 and err_ring == 0.0
        assert not torch.equal(rad_a, rad_b)
        eng.set_option(3, 1)
        eng.set_option(13, 2)
        eng.set_option(8, 1)
        seg_h, rad_h = eng.forward(x)
        assert rad_h.dtype == torch.float16 and torch.equal(rad_h, rad_a.to(torch.float16)) and torch.equal(seg_h, seg_a)
        assert eng.range_status() == 0
