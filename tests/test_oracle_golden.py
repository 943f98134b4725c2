"""The CPU oracle (oracle/) against the golden vectors produced by the reference itself.

These tests pin the oracle; the GPU parity tests (tests/test_gpu_*.py) then compare the HIP path
with the oracle.  Everything here is bit-exact: integer vote grids, tie lists, float64 centres.
"""
import glob
import json
import os

import numpy as np
import pytest

ACC = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "acc*.npz")))


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


@pytest.mark.parametrize("name", ACC)
def test_prepare_matches_reference(oracle, golden_dir, name):
    g = _load(golden_dir, name)
    pts, rv, info = oracle.prepare(g["xyz_m"], g["radial_dm"])
    assert info.dim == g["votes"].shape[0]
    assert np.array_equal(pts, g["points_vox"])  # bit-exact float64, incl. numpy's pairwise mean
    assert np.array_equal(rv.astype(g["radii_vox"].dtype), g["radii_vox"])


@pytest.mark.parametrize("name", ACC)
@pytest.mark.parametrize("sparse", [False, True])
def test_votes_match_reference(oracle, golden_dir, name, sparse):
    g = _load(golden_dir, name)
    center, votes, info, peak = oracle.accumulator_3d(g["xyz_m"], g["radial_dm"], sparse=sparse, return_all=True)
    assert np.array_equal(votes, g["votes"].astype(np.uint32))
    assert center.shape == g["center"].shape  # every tied maximum, C order
    assert np.array_equal(center, g["center"])


def test_error_cases(oracle, golden_dir):
    idx = json.load(open(os.path.join(golden_dir, "acc_index.json")))
    assert idx["acc_empty"]["raises"] == "ValueError"
    with pytest.raises(ValueError):
        oracle.accumulator_3d(np.zeros((0, 3)), np.zeros((0,), dtype=np.float32))
    c = idx["acc7_single_point_tiny_radius"]
    assert c["raises"] == "ValueError"
    with pytest.raises(ValueError):
        oracle.accumulator_3d(np.array(c["xyz_m"]), np.array(c["radial_dm"], dtype=c["radial_dtype"]))


def test_inputs_not_mutated(oracle, golden_dir):
    g = _load(golden_dir, "acc2_noisy_f32")
    xyz, rad = g["xyz_m"].copy(), g["radial_dm"].copy()
    oracle.accumulator_3d(xyz, rad)
    assert np.array_equal(xyz, g["xyz_m"]) and np.array_equal(rad, g["radial_dm"])


def test_sparse_equals_bruteforce_random(oracle):
    rng = np.random.default_rng(5)
    for trial in range(6):
        n = int(rng.integers(5, 400))
        xyz = rng.normal(size=(n, 3)) * rng.uniform(0.005, 0.05) + np.array([0.1, -0.2, 0.9])
        rad = rng.uniform(-0.05, 0.9, size=n).astype(np.float32 if trial % 2 else np.float64)
        pts, rv, info = oracle.prepare(xyz, rad)
        if info.dim > 110:
            continue
        a = oracle.fast_for(pts, rv, info.dim, sparse=False)
        b = oracle.fast_for(pts, rv, info.dim, sparse=True)
        assert np.array_equal(a, b)


def test_pairwise_mean_is_numpys(oracle):
    rng = np.random.default_rng(0)
    for n in list(range(1, 140)) + [255, 256, 257, 1000, 8191, 8192, 8193, 15000, 16385, 50007, 307200]:
        a = rng.normal(size=(n, 3)) * 100 + 37.0
        for c in range(3):
            assert oracle.mean_strided(a, c) == float(np.mean(a[:, c])), (n, c)


def test_seam_golden(oracle, golden_dir):
    for name in ("seam1_aligned", "seam2_depth_holes"):
        g = _load(golden_dir, name)
        xyz, rl = oracle.seam(g["sem"], g["radial"], g["depth"], float(g["max_radius_dm"]))
        assert np.array_equal(xyz, g["xyz_m"]), name
        assert rl.dtype == g["radial_list"].dtype and np.array_equal(rl, g["radial_list"]), name
    g = _load(golden_dir, "seam3_radial_npy")
    xyz, rl = oracle.seam_radial_only(g["radial_est"], g["depth"], float(g["max_radius_dm"]))
    assert np.array_equal(xyz, g["xyz_m"]) and np.array_equal(rl, g["radial_list"])


def test_seam_variants_golden(oracle, golden_dir):
    """The Occlusion-LINEMOD (pre-computed radial maps, AccumulatorSpace.py:849-851,858-864) and YCB-Video (:1050-1057) seams
    against outputs of the reference's own lines (tests/golden/make_golden.py seamv): bit-exact lists."""
    for name in ("seam_lmo1", "seam_lmo3"):
        g = _load(golden_dir, name)
        xyz, rl = oracle.seam_lmo_radial(g["radial_est"], g["depth"], float(g["max_radius_dm"]))
        assert np.array_equal(xyz, g["xyz_m"]) and rl.dtype == g["radial_list"].dtype and np.array_equal(rl, g["radial_list"]), name
        assert len(rl) > len(xyz)  # radii over the mask, points over mask & depth != 0
    g = _load(golden_dir, "seam_lmo1")
    assert oracle.seam_lmo_radial(np.zeros_like(g["radial_est"]), g["depth"], 3.0) == (None, None)  # radial_out.max() == 0 (:858)
    for name in ("seam_ycb1", "seam_ycb2"):
        g = _load(golden_dir, name)
        K = g["K"]
        xyz, rl = oracle.seam_ycb(g["sem"], g["radial"], g["depth"], float(g["factor_depth"]), (K[0, 0], K[1, 1], K[0, 2], K[1, 2]))
        assert np.array_equal(xyz, g["xyz"]) and np.array_equal(rl, g["radial_list"]), name


def test_backproject_golden(oracle, golden_dir):
    g = _load(golden_dir, "seam1_aligned")
    mask = (g["sem"] > np.float32(0.8)) & (g["radial"].astype(np.float64) <= float(g["max_radius_dm"]))
    pts = oracle.backproject(g["depth"].astype(np.int64) * mask)
    assert np.array_equal(pts, g["xyz_mm"])


def test_chain_golden(oracle, golden_dir):
    """seam -> Accumulator_3D, the per-keypoint body AccumulatorSpace.py:596-654."""
    g = _load(golden_dir, "seam4_chain")
    xyz, rl = oracle.seam(g["sem"], g["radial"], g["depth"], float(g["max_radius_dm"]))
    assert np.array_equal(xyz, g["xyz_m"]) and np.array_equal(rl, g["radial_list"])
    center = oracle.accumulator_3d(xyz, rl)
    assert np.array_equal(center, g["center"])
    assert np.abs(center[0] - g["keypoint_m"] * 1000).max() <= 5.0  # within one 5 mm voxel of the truth


def test_horn_golden(oracle, golden_dir):
    """util/horn.py:75-181 restated in C: bit-exact on the reference's outputs (incl. the degenerate identical sets)."""
    g = _load(golden_dir, "horn_cases")
    for i in range(len(g["n"])):
        n = int(g["n"][i])
        assert np.array_equal(oracle.horn(g["P1"][i][:n], g["P2"][i][:n]), g["RT"][i]), i


@pytest.mark.parametrize("name", ACC)
def test_plane_wise_peak_equals_grid_arg_max(oracle, golden_dir, name):
    """orc_peak_planes (the 1 mm oracle: no grid in memory) against the golden grids of the reference: peak, number of
    maxima, first maximum in C order, row 0 of `center`, and probed single-voxel counts."""
    g = _load(golden_dir, name)
    votes = g["votes"].astype(np.uint32)
    D = votes.shape[0]
    rng = np.random.default_rng(1)
    probe = [tuple(int(v) for v in rng.integers(0, D, size=3)) for _ in range(5)] + [tuple(int(v) for v in np.argwhere(votes == votes.max())[0])]
    r = oracle.peak_without_grid(g["xyz_m"], g["radial_dm"], probe_idx=probe)
    assert r["dim"] == D and r["peak"] == int(votes.max()) and r["n_ties"] == int((votes == votes.max()).sum())
    assert r["first_idx"] == tuple(int(v) for v in np.argwhere(votes == votes.max())[0])
    assert np.array_equal(r["center_mm"], g["center"][0])
    assert [int(v) for v in r["probe_votes"]] == [int(votes[p]) for p in probe]


def test_plane_wise_peak_on_a_fine_grid(oracle):
    """... and against the grid oracle at 2 mm on a thinned frame cloud (D ~ 500)."""
    import frames

    f = frames.synthetic_frame(5)
    xyz, rl = oracle.seam(f["seg"], f["radial"][1], f["depth"], f["max_radii_dm"][1])
    xyz, rl = xyz[::40], rl[::40]
    ref, votes, info, peak = oracle.accumulator_3d(xyz, rl, acc_unit=2.0, sparse=True, return_all=True)
    r = oracle.peak_without_grid(xyz, rl, acc_unit=2.0)
    assert (r["peak"], r["n_ties"], r["dim"]) == (peak, len(ref), info.dim)
    assert np.array_equal(r["center_mm"], ref[0])


ND = ["nd1_f32", "nd2_f64", "nd3_three_points", "nd4_unshifted"]


@pytest.mark.parametrize("name", ND)
def test_no_depth_accumulator_matches_reference(oracle, golden_dir, name):
    """SURVEY f-5: Accumulator_3D_no_depth / fast_for_no_depth / rgbd_to_point_cloud_no_depth restated in C against the
    reference's own outputs, quirks included (only three points recentred, unrounded radii, no radius margin on the grid,
    int64 truncation of the shifted row, every row x5)."""
    g = _load(golden_dir, name)
    K = g["K"]
    pts = oracle.backproject_no_depth((K[0, 0], K[1, 1], K[0, 2], K[1, 2]), g["depth_m"])
    assert np.array_equal(pts, g["xyz_m"])
    center, votes, info, peak, pv, rv = oracle.accumulator_3d_no_depth(g["xyz_m"], g["radial_dm"], return_all=True)
    assert np.array_equal(pv, g["points_vox"])
    assert np.array_equal(rv[: len(g["radii_vox"])].astype(g["radii_vox"].dtype), g["radii_vox"])
    assert np.array_equal(votes, g["votes"].astype(np.uint32))
    assert center.dtype == np.int64 and np.array_equal(center, g["center"])


def test_no_depth_error_cases(oracle, golden_dir):
    idx = json.load(open(os.path.join(golden_dir, "nd_index.json")))
    assert idx["nd_two_points"]["raises"] == "IndexError" and idx["nd_empty"]["raises"] == "IndexError"
    with pytest.raises(IndexError):
        oracle.accumulator_3d_no_depth(np.zeros((2, 2, 3)) + 0.1, np.array([0.1, 0.1], dtype=np.float32))
    with pytest.raises(IndexError):
        oracle.accumulator_3d_no_depth(np.zeros((1, 0, 3)), np.zeros((0,), dtype=np.float32))
    assert any(json.load(open(os.path.join(golden_dir, "nd_index.json")))[n]["ties"] > 1 for n in ND)
