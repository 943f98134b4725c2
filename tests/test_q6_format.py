"""The q6 side band of the split-precision engine (engine option 11, include/rcvpose_hip.h) on the host: the weight packer of
rcvpose_amd/fcn_engine.py against the oracle's numpy restatement of the format (oracle/q6_ref.py) and against conversions the
hardware produced (tools/probe/fp6_mfma_probe.hip, profiles/r05_fp6_mfma_probe.txt)."""
import json
import os

import numpy as np
import torch

from oracle import q6_ref
from rcvpose_amd import _hip, fcn_engine

# v_cvt_scalef32_pk32_fp6_f16 at scale 1.0 on gfx950 (input -> value of the code it produced)
HW = [(0, 0), (0.0625, 0), (0.125, 0.125), (0.1875, 0.25), (0.3125, 0.25), (0.4375, 0.5), (0.9375, 1), (1, 1), (1.0625, 1),
      (1.1875, 1.25), (1.9, 1.875), (2, 2), (3.75, 3.75), (3.9, 4), (4, 4), (7.5, 7.5), (0.5625, 0.5), (0.6875, 0.75),
      (1.3125, 1.25), (1.4375, 1.5), (2.125, 2), (2.375, 2.5), (5.25, 5), (5.75, 6), (-0.19, -0.25), (-1.06, -1)]


def test_encoders_reproduce_the_hardware_conversion():
    vals = np.array([np.float16(v) for v, _ in HW], dtype=np.float64)
    want = np.array([w for _, w in HW])
    for lo in range(0, len(vals), 15):
        blk = np.concatenate([vals[lo:lo + 15], [7.5]])  # the block's largest element 7.5: scale 2^0
        blk = np.concatenate([blk, np.zeros(16 - len(blk))])
        c, s = q6_ref.encode(blk[None])
        assert s[0] == 127
        np.testing.assert_array_equal(q6_ref.decode(c, s)[0][:len(want[lo:lo + 15])], want[lo:lo + 15])
        c2, s2 = fcn_engine.q6_encode(torch.from_numpy(blk[None]))
        np.testing.assert_array_equal(c2.numpy(), c)
        assert int(s2[0]) == 127


def test_product_encoder_equals_oracle_encoder():
    rng = np.random.default_rng(3)
    v = rng.standard_normal((4000, 16)) * np.exp2(rng.integers(-22, 14, size=(4000, 1)))
    v = v.astype(np.float16).astype(np.float64)
    v[7] = 0
    v[8, 3:] = 0
    v[9] = -v[9]
    v[10] = np.float16(6.0e-8)   # half subnormals (lo halves are often subnormal)
    v[11, :] = 65504.0
    c, s = q6_ref.encode(v)
    c2, s2 = fcn_engine.q6_encode(torch.from_numpy(v))
    np.testing.assert_array_equal(c2.numpy(), c)
    np.testing.assert_array_equal(s2.numpy().astype(np.int64), s)
    assert s[7] == 1 and not c[7].any()
    # decoded error: at most half a step of the top binade (1/16 of the block's largest power of two ... 6.25 % of max)
    d = q6_ref.decode(c, s)
    top = np.abs(v).max(axis=1)
    assert np.all(np.abs(d - v).max(axis=1) <= top * 0.0626 + 1e-300)
    np.testing.assert_array_equal(fcn_engine.q6_decode(c2, s2).numpy(), d)
    # chunk packing: element j in bits [6j, 6j+6), scale byte
    for pos in (12, 14):
        ch = q6_ref.chunk_bytes(c, s, pos)
        np.testing.assert_array_equal(fcn_engine.q6_chunk(c2, s2, pos).numpy(), ch)
        cc, ss = q6_ref.unpack_chunk(ch, pos)
        np.testing.assert_array_equal(cc, c)
        np.testing.assert_array_equal(ss, s)


def test_q6_weight_rows_and_program_flags(golden_dir):
    """pack_weights: the 35 stride-1 3x3 convolutions of layer 3 get a q6 copy right behind their first copy, same size; its
    rows decode to the quantised halves of the first copy."""
    import fcn_weights

    shapes = json.load(open(os.path.join(golden_dir, "fcn_state_dict_keys.json")))
    sd = {k: torch.from_numpy(v) for k, v in fcn_weights.make_state_dict_numpy(shapes, seed=0).items()}
    b = fcn_engine.build_program(64, 96)
    ops, _ = fcn_engine.assign_slots(b)
    w, bias, ops = fcn_engine.pack_weights([sd], b, ops, _hip.RCV_F16X3, q6=True)
    flagged = [(o, sp) for o, sp in zip(ops, b.wspecs) if o.get("wq_off", 0) and o["cout"] % 128 == 0]   # (conv7's second copy is the composed tail's)
    assert len(flagged) == 44 and all(sp["conv"][:7] in ("block2.", "block3.", "block4.") and sp["conv"].endswith(".conv2") for _, sp in flagged)
    w_off = fcn_engine.pack_weights([sd], b, [dict(o) for o in ops], _hip.RCV_F16X3, q6=False)[0]
    assert w.numel() == w_off.numel() + (7 * 128 * 128 + 35 * 256 * 256 + 2 * 512 * 512) * 9 * 2   # (both carry the composed tail copy)
    o, sp = [f for f in flagged if f[1]["conv"].startswith("block3.")][0]
    n = 256 * 9 * 256 * 2
    first = w[o["w_off"]:o["w_off"] + n].view(torch.uint8).numpy().reshape(256, 9, 8, 128)
    q = w[o["wq_off"]:o["wq_off"] + n].view(torch.uint8).numpy().reshape(256, 9, 8, 128)
    assert o["wq_off"] == o["w_off"] + n
    np.testing.assert_array_equal(q[..., :64], first[..., :64])       # the hi halves, unchanged
    hi = first[..., :64].copy().view(np.float16).astype(np.float64)     # [256, 9, 8, 32]
    lo = first[..., 64:].copy().view(np.float16).astype(np.float64)
    for k, src in enumerate((lo[..., :16], lo[..., 16:], hi[..., :16], hi[..., 16:])):
        c, s = q6_ref.unpack_chunk(q[..., 64 + 16 * k:80 + 16 * k], 14)
        ce, se = q6_ref.encode(src)
        np.testing.assert_array_equal(c, ce)
        np.testing.assert_array_equal(s, se)
        assert not q[..., 64 + 16 * k + 12].any() and not q[..., 64 + 16 * k + 13].any() and not q[..., 64 + 16 * k + 15].any()


def test_composed_tail_weights_equal_upsampling_followed_by_the_convolution():
    """fcn_engine.compose_upsample_conv3 (engine option 13): a 3x3 convolution on the x2 bilinear upsampling (reference
    models/fcnresnet.py:177-183) == the composed 3x3 convolution on the replicate-padded low-resolution tensor, parity by parity,
    everywhere but on the outermost ring of output pixels (which k_tail_ring recomputes)."""
    import torch.nn.functional as F

    torch.manual_seed(0)
    w = torch.randn(6, 3, 3, 8, dtype=torch.float64)
    z = torch.randn(2, 8, 7, 9, dtype=torch.float64)
    ref = F.conv2d(F.interpolate(z, scale_factor=2, mode="bilinear", align_corners=False), w.permute(0, 3, 1, 2), padding=1)
    wc = fcn_engine.compose_upsample_conv3(w)
    low = F.conv2d(F.pad(z, (1, 1, 1, 1), mode="replicate"), wc.permute(0, 3, 1, 2))
    out = low.reshape(2, 2, 2, 6, 7, 9).permute(0, 3, 4, 1, 5, 2).reshape(2, 6, 14, 18)
    assert float((out - ref)[:, :, 1:-1, 1:-1].abs().max()) < 1e-12
    assert float((out - ref).abs().max()) > 1e-3       # ... the ring really differs
    b = fcn_engine.build_program(64, 96)
    assert [i for i in range(len(b.ops)) if fcn_engine.tail_composed_wanted(b, i)] == [len(b.ops) - 2]
    assert not any(fcn_engine.tail_composed_wanted(fcn_engine.build_program_res(64, 96), i) for i in range(10))
