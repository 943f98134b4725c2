"""BASELINE.json configs[4] (stress): 1 mm voxels, ~50k-point YCB-shaped cloud, float16 radial maps, 8 frames."""
import numpy as np
import pytest
import torch

import frames

pytestmark = pytest.mark.gpu


def _K(f):
    return np.array([[f["K"][0], 0, f["K"][2]], [0, f["K"][1], f["K"][3]], [0, 0, 1]])


def test_fp16_radial_seam_equals_oracle_on_rounded_maps(oracle):
    from rcvpose_amd import AccumulatorSpace as A
    from rcvpose_amd import _hip

    f = frames.synthetic_frame(1, shape="ycb")
    rad16 = f["radial"][0].astype(np.float16)
    rmax = float(rad16.astype(np.float32).max())
    seg, depth = torch.from_numpy(f["seg"]).cuda(), torch.from_numpy(f["depth"].view(np.int16)).cuda()
    ws = A.seam_compact(seg, torch.from_numpy(rad16).cuda(), depth, rmax, K=_K(f), mode=_hip.RCV_SEAM_RADIAL_F16)
    n, m = [int(v) for v in ws.counts.cpu()]
    xyz, rl = oracle.seam(f["seg"], rad16.astype(np.float32), f["depth"], rmax, K=f["K"])
    assert 47500 <= n <= 52500 and (n, m) == (len(xyz), len(rl))
    assert np.array_equal(ws.xyz[:n].cpu().numpy(), xyz) and np.array_equal(ws.radial[:m].cpu().numpy(), rl)


def test_one_millimetre_grid_both_engines_agree():
    """D ~ 1000 (10^9 voxels): the exact peak search and the LDS rasteriser must give the same peak, tie count and
    keypoint (the dense oracle grid would need > 4 GB; the engines are checked against the oracle at 5 mm)."""
    from rcvpose_amd import AccumulatorSpace as A

    f = frames.synthetic_frame(2, shape="ycb")
    seg, depth = torch.from_numpy(f["seg"]).cuda(), torch.from_numpy(f["depth"].view(np.int16)).cuda()
    out = {}
    ws = A.AccWorkspace(480 * 640)
    for mode in ("auto", "full"):
        A.set_vote_mode(mode)
        try:
            A.seam_compact(seg, torch.from_numpy(f["radial"][1]).cuda(), depth, float(f["max_radii_dm"][1]), K=_K(f), workspace=ws)
            A.accumulate_on_device(ws.xyz, ws.radial, ws.counts, ws.max_points, ws.max_points, ws, acc_unit=1.0)
            r = ws.read_result()
        finally:
            A.set_vote_mode("auto")
        assert r.status == 0 and r.dim > 900
        out[mode] = (r.peak_votes, r.n_ties, tuple(r.first_idx), tuple(r.center_mm))
    assert out["auto"] == out["full"]
    assert np.abs(np.array(out["auto"][3]) - f["keypoints_m"][1] * 1000).max() <= 5.0  # 1 mm voxels: within 5 voxels


def test_intermediate_grid_against_oracle(oracle):
    """2.5 mm voxels on a thinned cloud: both engines against the oracle's sparse grid (D ~ 420)."""
    from rcvpose_amd import AccumulatorSpace as A

    f = frames.synthetic_frame(3)
    xyz, rl = oracle.seam(f["seg"], f["radial"][2], f["depth"], f["max_radii_dm"][2])
    xyz, rl = xyz[::6], rl[::6]
    ref, votes, info, peak = oracle.accumulator_3d(xyz, rl, acc_unit=2.5, sparse=True, return_all=True)
    for mode in ("auto", "full"):
        A.set_vote_mode(mode)
        try:
            c = A.Accumulator_3D(xyz, rl, acc_unit=2.5)
        finally:
            A.set_vote_mode("auto")
        assert np.array_equal(c, ref), mode
