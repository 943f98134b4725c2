"""Engine option 11 of the split-precision FCN engine: the cross terms hi*lo + lo*hi of the 3x3 convolutions of layer 3 run as one
block-scaled fp6 (e2m3) matrix instruction (include/rcvpose_hip.h).  The reference's forward is fp32
(models/fcnresnet.py:121-191); north_star's gate on the maps is 1e-4.  Tested here: the in-place pack kernel against the oracle's
restatement of the format, one convolution against the oracle's emulation of the arithmetic (both kernels that can run it, bit
for bit equal to each other), and the whole network against the reference's golden forward."""
import ctypes
import json
import os

import numpy as np
import pytest
import torch

import fcn_weights
from oracle import q6_ref

pytestmark = pytest.mark.gpu

REL = 1e-4         # north star
REL_GUARD = 9e-5   # regression guard for the q6 engine (measured 2-8e-5 by network; the three-MFMA engine: 0.4-3.4e-5)


@pytest.fixture(scope="module")
def shapes(golden_dir):
    return json.load(open(os.path.join(golden_dir, "fcn_state_dict_keys.json")))


def _sd(shapes, seed=0):
    return {k: torch.from_numpy(v) for k, v in fcn_weights.make_state_dict_numpy(shapes, seed=seed).items()}


def _rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


def _rows(v):
    """float64 [..., C] -> split-precision rows uint8 [..., C/32, 128]: [32 hi halves | 32 lo halves]."""
    hi, lo = q6_ref.split_halves(v)
    shp = v.shape[:-1] + (v.shape[-1] // 32, 32)
    return np.concatenate([hi.reshape(shp).view(np.uint8), lo.reshape(shp).view(np.uint8)], axis=-1)


def test_q6_pack_kernel_against_the_oracle_format():
    from rcvpose_amd import _hip

    rng = np.random.default_rng(11)
    n = 64 * 37 + 5  # not a multiple of the 64 rows a wave takes
    v = np.maximum(rng.standard_normal((n, 32)), -0.2) * np.exp2(rng.integers(-12, 15, size=(n, 1)))
    v[3] = 0
    v[4, :16] = 0
    v[5] = 60000.0
    v[6] = 3.0e-6
    rows = _rows(v).reshape(n, 128)
    dev = torch.from_numpy(rows.copy()).cuda()
    _hip.check(_hip.lib().rcv_fcn_debug_q6_pack(_hip.ptr(dev), n, _hip.current_stream()), "rcv_fcn_debug_q6_pack")
    got = dev.cpu().numpy()
    np.testing.assert_array_equal(got[:, :64], rows[:, :64])
    hi = rows[:, :64].copy().view(np.float16).astype(np.float64)
    lo = rows[:, 64:].copy().view(np.float16).astype(np.float64)
    for k, src in enumerate((hi[:, :16], hi[:, 16:], lo[:, :16], lo[:, 16:])):
        chunk = got[:, 64 + 16 * k:80 + 16 * k]
        c, s = q6_ref.unpack_chunk(chunk, 12)
        ce, se = q6_ref.encode(src)
        np.testing.assert_array_equal(s, se)
        np.testing.assert_array_equal(q6_ref.decode(c, s), q6_ref.decode(ce, se))   # (values: a zero may carry either sign)
        assert not chunk[:, 13:].any()


def _one_conv_engine(w, bias, H, W, q6):
    """A one-op program (3x3 / stride 1, C -> Cout) on the split engine: returns (engine pieces, run(x rows) -> float32 [B,H,W,Cout])."""
    from rcvpose_amd import _hip, fcn_engine

    cout, _, _, cin = w.shape
    flat, scale = fcn_engine.pack_split(torch.from_numpy(w), _hip.OP_CONV)
    parts = [flat]
    if q6:
        parts.append(fcn_engine.pack_split_q6(torch.from_numpy(w)))
    wdev = torch.cat(parts).cuda()
    bdev = torch.cat([torch.from_numpy(bias).to(torch.float32), scale]).cuda()
    op = _hip.Op()
    for f, val in dict(kind=_hip.OP_CONV, src0=0, src1=-1, res=-1, dst=1, c0=cin, c1=0, cout=cout, hin=H, win=W, hout=H, wout=W, ksize=3,
                       stride=1, pad=1, relu=1, w_off=0, b_off=0, wq_off=flat.numel() if q6 else 0).items():
        setattr(op, f, val)
    slots = (ctypes.c_int64 * 2)(H * W * cin, H * W * cout)
    h = ctypes.c_void_p()
    L = _hip.lib()
    _hip.check(L.rcv_fcn_create((_hip.Op * 1)(op), 1, 2, slots, 1, _hip.RCV_F16X3, _hip.ptr(wdev), _hip.ptr(bdev), ctypes.byref(h)), "rcv_fcn_create")

    def run(x_rows, B, halo=1):
        _hip.check(L.rcv_fcn_set_option(h, 7, halo))
        nbytes = int(L.rcv_fcn_workspace_bytes(h, B))
        ws = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
        ws[:x_rows.size] = torch.from_numpy(x_rows.reshape(-1)).cuda()
        dummy = torch.zeros((B, 3, H, W), device="cuda")
        seg = torch.zeros((1, B, 1, H, W), device="cuda")
        rad = torch.zeros_like(seg)
        _hip.check(L.rcv_fcn_forward(h, _hip.ptr(dummy), B, _hip.ptr(seg), _hip.ptr(rad), _hip.ptr(ws), nbytes, _hip.current_stream()), "rcv_fcn_forward")
        out = torch.empty((B, H, W, cout), dtype=torch.float32, device="cuda")
        _hip.check(L.rcv_fcn_debug_slot(h, _hip.ptr(ws), B, 0, 1, _hip.ptr(out), _hip.current_stream()), "rcv_fcn_debug_slot")
        torch.cuda.synchronize()
        return out.cpu().numpy()

    return (wdev, bdev, h), scale.numpy().astype(np.float64), flat, run


def test_q6_convolution_against_the_oracle_emulation():
    """256 -> 256 at 30x40 (the layer-3 shape): 24 frames run through k_conv_halo<128,5,true>, one frame through the plain kernel;
    both equal the oracle's emulation of hi*hi + q(a_hi) q(w_lo) + q(a_lo) q(w_hi) to summation order and each other bit for bit."""
    rng = np.random.default_rng(5)
    C, H, W, B = 256, 30, 40, 24
    w = rng.standard_normal((C, 3, 3, C)) * (0.5 / np.sqrt(9 * C)) * np.exp2(rng.integers(-3, 4, size=(C, 1, 1, 1)))
    bias = rng.standard_normal(C) * 0.1
    x = np.maximum(rng.standard_normal((B, H, W, C)), 0) * np.exp2(rng.integers(-4, 5, size=(B, H, W, 1)))
    hi, lo = q6_ref.split_halves(x)
    x22 = hi.astype(np.float64) + lo.astype(np.float64)
    keep, sc, flat, run = _one_conv_engine(w, bias, H, W, q6=True)
    got = run(_rows(x22), B)
    # the weights as stored: per-channel scaled halves
    n = C * 9 * C * 2
    st = flat[:n].view(torch.uint8).numpy().reshape(C, 3, 3, C // 32, 128)
    w_st = (st[..., :64].copy().view(np.float16).astype(np.float64) + st[..., 64:].copy().view(np.float16).astype(np.float64)).reshape(C, 3, 3, C)
    sel = [0, B - 1]
    ref = q6_ref.conv3x3_q6(x22[sel], w_st, None) * sc + bias
    ref = np.maximum(ref, 0)
    err = max(_rel(got[f], ref[i]) for i, f in enumerate(sel))
    print("q6 3x3 convolution vs emulation: %.2e" % err)
    assert err <= 2e-6
    one = run(_rows(x22[:1]), 1)          # few tiles: the plain kernel (k_conv) in its q6 form
    assert np.array_equal(one[0], got[0])
    assert np.array_equal(run(_rows(x22), B, halo=0), got)
    # and the three-MFMA engine on the same operands is close, not equal: the cross terms carry 4-bit operands
    keep3, _, _, run3 = _one_conv_engine(w, bias, H, W, q6=False)  # (keep3 owns the device weights)
    full = run3(_rows(x22[:2]), 2)
    d = _rel(got[:2], full)
    print("q6 vs three-MFMA convolution: %.2e" % d)
    assert 0 < d <= 2e-5
    del keep, keep3


def test_q6_engine_stages_and_output_vs_golden(shapes, golden_dir):
    """The reference module's golden forward (64x96): every named stage and the output planes within 1e-4 with option 11 on."""
    from rcvpose_amd.fcn_engine import FcnEngine

    g = np.load(os.path.join(golden_dir, "fcn1_64x96.npz"))
    x = torch.from_numpy(fcn_weights.make_input(int(g["input_seed"]), (1, 3, 64, 96))).cuda()
    eng = FcnEngine([_sd(shapes)], 64, 96, dtype="f16x3", reuse_slots=False, q6=True)
    assert sum(1 for o in eng.ops if o["wq_off"] and o["cout"] % 128 == 0) == 44   # (conv7 carries the composed tail's copy)
    for key in (3, 4):
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
        worst = max(worst, _rel(got, ref))
        assert _rel(got, ref) <= REL, (k, _rel(got, ref))
    er, es = _rel(rad[0].cpu().numpy(), g["radial"]), _rel(seg[0].cpu().numpy(), g["seg"])
    print("f16x3 + q6, 64x96: worst stage %.2e radial %.2e seg %.2e" % (worst, er, es))
    assert er <= REL_GUARD and es <= REL_GUARD
    eng.set_option(11, 0)
    seg0, rad0 = eng.forward(x)
    assert _rel(rad0[0].cpu().numpy(), g["radial"]) <= 3e-5 and not torch.equal(rad0, rad)


def test_q6_engine_full_size_whole_maps_and_batch_independence(shapes, golden_dir):
    """480x640, three networks: sampled pixels of the golden forward, whole maps against the fp32 engine, and a frame's maps do not
    depend on the batch (8 frames: k_conv_halo<.., true>; 1 frame: the plain kernel) or on the engine's kernel options."""
    from rcvpose_amd.fcn_engine import FcnEngine

    sds = [_sd(shapes, s) for s in range(3)]
    g = np.load(os.path.join(golden_dir, "fcn2_480x640_samples.npz"))
    x1 = torch.from_numpy(fcn_weights.make_input(int(g["input_seed"]), (1, 3, 480, 640))).cuda()
    x = torch.cat([x1, torch.from_numpy(fcn_weights.make_input(17, (7, 3, 480, 640))).cuda()])
    eng = FcnEngine(sds, 480, 640, dtype="f16x3", q6=True)
    eng.set_option(13, 0)   # (the parity-composed tail is another summation than the three-kernel tail the options below fall back to)
    seg, rad = [t.clone() for t in eng.forward(x)]
    pix = g["pix"]
    err_r = np.abs(rad[0, 0].cpu().numpy().reshape(-1)[pix] - g["radial"]).max() / float(g["radial_absmax"])
    err_s = np.abs(seg[0, 0].cpu().numpy().reshape(-1)[pix] - g["seg"]).max() / float(g["seg_absmax"])
    print("f16x3 + q6, 480x640 (sampled golden): radial %.2e seg %.2e" % (err_r, err_s))
    assert err_r <= REL_GUARD and err_s <= REL_GUARD
    seg_b, rad_b = FcnEngine(sds, 480, 640, dtype="fp32").forward(x[:4])
    for k in range(3):
        er = float((rad[k, :4] - rad_b[k]).abs().max() / rad_b[k].abs().max())
        es = float((seg[k, :4] - seg_b[k]).abs().max() / seg_b[k].abs().max())
        print("f16x3 + q6 vs fp32 engine, network %d: radial %.2e seg %.2e" % (k, er, es))
        assert er <= REL_GUARD and es <= REL_GUARD
    seg1, rad1 = eng.forward(x[:1])
    assert torch.equal(seg1, seg[:, :1]) and torch.equal(rad1, rad[:, :1])
    for key in (1, 3, 4, 7, 9):
        eng.set_option(key, 0)
    seg2, rad2 = eng.forward(x)
    assert torch.equal(seg2, seg) and torch.equal(rad2, rad)
    assert eng.range_status() == 0
