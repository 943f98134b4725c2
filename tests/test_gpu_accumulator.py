"""HIP voting path (through the C ABI) against the CPU oracle and the golden vectors.  Bit-exact."""
import glob
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ACC = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "acc*.npz")))


@pytest.fixture(scope="module")
def A():
    from rcvpose_amd import AccumulatorSpace

    return AccumulatorSpace


@pytest.mark.parametrize("name", ACC)
def test_golden_votes_and_centres(A, golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    xyz, rad = g["xyz_m"].copy(), g["radial_dm"].copy()
    center, votes, res = A.Accumulator_3D(xyz, rad, return_votes=True)
    assert res.dim == g["votes"].shape[0]
    assert np.array_equal(votes, g["votes"].astype(np.uint32))      # VoteMap_3D, exact integer counts
    assert center.shape == g["center"].shape                          # every tie, C order
    assert np.array_equal(center, g["center"])                        # float64, bit-exact
    assert np.array_equal(xyz, g["xyz_m"]) and np.array_equal(rad, g["radial_dm"])  # inputs untouched
    for mode in ("auto", "full"):                                     # hot path (no grid in HBM), both arg-max engines
        c2 = A.Accumulator_3D(xyz, rad, engine=mode)
        assert np.array_equal(c2, g["center"]), mode


@pytest.mark.parametrize("name", ACC)
def test_prepared_points_bit_exact(A, golden_dir, name):
    import torch

    from rcvpose_amd import _hip

    g = np.load(os.path.join(golden_dir, name + ".npz"))
    n, m = len(g["xyz_m"]), len(g["radial_dm"])
    ws = A.AccWorkspace(max(n, m))
    xyz_d = torch.from_numpy(g["xyz_m"]).cuda()
    rad_d = torch.from_numpy(g["radial_dm"]).cuda()
    A.accumulate_on_device(xyz_d, rad_d, None, n, m, ws)
    pts = torch.empty((max(n, m), 3), dtype=torch.float64, device="cuda")
    ri = torch.empty((max(n, m),), dtype=torch.int32, device="cuda")
    _hip.check(_hip.lib().rcv_acc_debug_points(_hip.ptr(ws.ws), _hip.ptr(pts), _hip.ptr(ri), max(n, m),
                                               _hip.current_stream()))
    assert np.array_equal(pts[:n].cpu().numpy(), g["points_vox"])
    assert np.array_equal(ri[:n].cpu().numpy(), np.rint(g["radii_vox"][:n].astype(np.float64)).astype(np.int32))


def test_error_behaviour(A, golden_dir):
    idx = json.load(open(os.path.join(golden_dir, "acc_index.json")))
    with pytest.raises(ValueError):
        A.Accumulator_3D(np.zeros((0, 3)), np.zeros((0,), dtype=np.float32))
    c = idx["acc7_single_point_tiny_radius"]
    with pytest.raises(ValueError):
        A.Accumulator_3D(np.array(c["xyz_m"]), np.array(c["radial_dm"], dtype=c["radial_dtype"]))


def _random_cloud(rng, n, spread, rmin, rmax, dtype):
    xyz = rng.normal(size=(n, 3)) * spread + np.array([0.1, -0.2, 0.9])
    rad = rng.uniform(rmin, rmax, size=n).astype(dtype)
    return xyz, rad


@pytest.mark.parametrize("seed,n,spread,rmax,dtype", [
    (1, 50, 0.01, 0.5, np.float32),
    (2, 700, 0.03, 1.2, np.float64),
    (3, 3000, 0.02, 2.0, np.float32),     # D ~ 100: several tiles per axis
    (4, 300, 0.05, 4.5, np.float32),      # D ~ 230: TI = 8 tiles, long columns
])
def test_random_clouds_vs_oracle(A, oracle, seed, n, spread, rmax, dtype):
    rng = np.random.default_rng(seed)
    xyz, rad = _random_cloud(rng, n, spread, -0.05, rmax, dtype)
    center, votes, res = A.Accumulator_3D(xyz, rad, return_votes=True)
    ref_center, ref_votes, info, peak = oracle.accumulator_3d(xyz, rad, sparse=True, return_all=True)
    assert res.dim == info.dim and res.zero_boundary == info.zero_boundary
    assert np.array_equal(votes, ref_votes)
    assert res.peak_votes == peak
    assert np.array_equal(center, ref_center)
    _check_hot_path_modes(A, xyz, rad, ref_center, peak)


def _check_hot_path_modes(A, xyz, rad, ref_center, peak):
    """The two arg-max engines of the hot path (exact peak search / LDS rasteriser) against the oracle."""
    import torch

    n, m = len(xyz), len(rad)
    xyz_d, rad_d = torch.from_numpy(np.ascontiguousarray(xyz)).cuda(), torch.from_numpy(np.ascontiguousarray(rad)).cuda()
    ws = A.AccWorkspace(max(n, m))
    for mode in ("auto", "full"):
        A.accumulate_on_device(xyz_d, rad_d, None, n, m, ws, engine=mode)
        res = ws.read_result()
        assert res.status == 0 and res.peak_votes == peak and res.n_ties == len(ref_center), (mode, res.peak_votes, res.n_ties)
        assert np.array_equal(np.array(res.center_mm[:]), ref_center[0]), mode


def test_full_size_frame_cloud(A, oracle):
    """~15k points, 5 mm voxels, radii up to ~0.47 m: exact grid equality at D ~ 236."""
    import frames

    f = frames.synthetic_frame(0)
    xyz, rad = oracle.seam(f["seg"], f["radial"][0], f["depth"], f["max_radii_dm"][0])
    assert 14000 <= len(xyz) <= 16000
    center, votes, res = A.Accumulator_3D(xyz, rad, return_votes=True)
    ref_center, ref_votes, info, peak = oracle.accumulator_3d(xyz, rad, sparse=True, return_all=True)
    assert res.dim == info.dim
    assert np.array_equal(votes, ref_votes)
    assert np.array_equal(center, ref_center)
    _check_hot_path_modes(A, xyz, rad, ref_center, peak)
    assert np.abs(center[0] - f["keypoints_m"][0] * 1000).max() <= 15.0  # sanity vs the truth; parity is the line above


def test_seam_golden(A, golden_dir):
    import torch

    for name in ("seam1_aligned", "seam2_depth_holes"):
        g = np.load(os.path.join(golden_dir, name + ".npz"))
        ws = A.seam_compact(torch.from_numpy(g["sem"]).cuda(), torch.from_numpy(g["radial"]).cuda(),
                            torch.from_numpy(g["depth"].view(np.int16)).cuda(), float(g["max_radius_dm"]))
        n, m = [int(v) for v in ws.counts.cpu()]
        assert np.array_equal(ws.xyz[:n].cpu().numpy(), g["xyz_m"]), name
        assert np.array_equal(ws.radial[:m].cpu().numpy(), g["radial_list"]), name
    g = np.load(os.path.join(golden_dir, "seam3_radial_npy.npz"))
    ws = A.seam_compact(None, torch.from_numpy(g["radial_est"]).cuda(), torch.from_numpy(g["depth"].view(np.int16)).cuda(),
                        float(g["max_radius_dm"]), mode=1)
    n, m = [int(v) for v in ws.counts.cpu()]
    assert n == m
    assert np.array_equal(ws.xyz[:n].cpu().numpy(), g["xyz_m"])
    assert np.array_equal(ws.radial[:m].cpu().numpy(), g["radial_list"])
    g = np.load(os.path.join(golden_dir, "seam1_aligned.npz"))
    mask = (g["sem"] > np.float32(0.8)) & (g["radial"].astype(np.float64) <= float(g["max_radius_dm"]))
    assert np.array_equal(A.rgbd_to_point_cloud(A.linemod_K, g["depth"].astype(np.int64) * mask), g["xyz_mm"])


def test_seam_variants_golden(A, golden_dir):
    """The seams of the Occlusion-LINEMOD (pre-computed radial maps: mode 0 with seg := radial > 0, threshold 0.5, as
    estimate_6d_pose_lmo calls it) and YCB-Video (mode 3) drivers against the lists the reference's own lines produced
    (AccumulatorSpace.py:849-851,858-864 and :1050-1057; tests/golden/make_golden.py seamv): bit-exact."""
    import torch

    for name in ("seam_lmo1", "seam_lmo3"):
        g = np.load(os.path.join(golden_dir, name + ".npz"))
        rad = torch.from_numpy(g["radial_est"]).cuda()
        seg = (rad > 0).to(torch.float32)
        ws = A.seam_compact(seg, rad, torch.from_numpy(g["depth"].view(np.int16)).cuda(), float(g["max_radius_dm"]), mode=0, seg_thr=0.5)
        n, m = [int(v) for v in ws.counts.cpu()]
        assert m > n
        assert np.array_equal(ws.xyz[:n].cpu().numpy(), g["xyz_m"]), name
        assert np.array_equal(ws.radial[:m].cpu().numpy(), g["radial_list"]), name
    for name in ("seam_ycb1", "seam_ycb2"):
        g = np.load(os.path.join(golden_dir, name + ".npz"))
        ws = A.seam_compact(torch.from_numpy(g["sem"]).cuda(), torch.from_numpy(g["radial"]).cuda(),
                            torch.from_numpy(g["depth"].view(np.int16)).cuda(), float(g["factor_depth"]), K=g["K"], mode=3)
        n, m = [int(v) for v in ws.counts.cpu()]
        assert np.array_equal(ws.xyz[:n].cpu().numpy(), g["xyz"]), name
        assert np.array_equal(ws.radial[:m].cpu().numpy(), g["radial_list"]), name


def test_seam_to_vote_chain_on_device(A, golden_dir):
    """seam -> Accumulator_3D without touching the host in between (AccumulatorSpace.py:596-654)."""
    import torch

    g = np.load(os.path.join(golden_dir, "seam4_chain.npz"))
    ws = A.seam_compact(torch.from_numpy(g["sem"]).cuda(), torch.from_numpy(g["radial"]).cuda(),
                        torch.from_numpy(g["depth"].view(np.int16)).cuda(), float(g["max_radius_dm"]))
    A.accumulate_on_device(ws.xyz, ws.radial, ws.counts, ws.max_points, ws.max_points, ws)
    res = ws.read_result()
    assert res.status == 0 and res.n_points == len(g["xyz_m"])
    assert np.array_equal(np.array(res.center_mm[:]), g["center"][0])


def test_depth_hole_misalignment_is_reproduced(A, oracle, golden_dir):
    """radial_list longer than xyz (SURVEY a-4): same pairing and same radius_max as the reference."""
    import torch

    g = np.load(os.path.join(golden_dir, "seam2_depth_holes.npz"))
    assert len(g["radial_list"]) > len(g["xyz_m"])
    center = A.Accumulator_3D(g["xyz_m"], g["radial_list"])
    assert np.array_equal(center, oracle.accumulator_3d(g["xyz_m"], g["radial_list"], sparse=True))


ND = ["nd1_f32", "nd2_f64", "nd3_three_points", "nd4_unshifted"]


@pytest.mark.parametrize("name", ND)
def test_no_depth_accumulator_golden(A, golden_dir, name):
    """SURVEY f-5: rgbd_to_point_cloud_no_depth + Accumulator_3D_no_depth on the device against the reference's own outputs
    (full vote grids, every tie row, int64 centres), both arg-max engines."""
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    pts = A.rgbd_to_point_cloud_no_depth(g["K"], g["depth_m"])
    assert pts.shape == g["xyz_m"].shape and np.array_equal(pts, g["xyz_m"])
    xyz, rad = g["xyz_m"].copy(), g["radial_dm"].copy()
    center, votes, res = A.Accumulator_3D_no_depth(xyz, rad, None, return_votes=True)
    assert res.dim == g["votes"].shape[0]
    assert np.array_equal(votes, g["votes"].astype(np.uint32))
    assert center.dtype == np.int64 and center.shape == g["center"].shape and np.array_equal(center, g["center"])
    assert np.array_equal(xyz, g["xyz_m"]) and np.array_equal(rad, g["radial_dm"])
    for mode in ("auto", "full"):
        assert np.array_equal(A.Accumulator_3D_no_depth(xyz, rad, None, engine=mode), g["center"]), mode


def test_no_depth_accumulator_vs_oracle_and_errors(A, oracle):
    """A larger hypothesis sweep (20 depth steps x 400 pixels, float32 radii) against the C oracle, and the error behaviour."""
    rng = np.random.default_rng(12)
    depth = np.zeros((48, 64))
    pix = rng.choice(48 * 64, size=400, replace=False)
    depth.reshape(-1)[pix] = 0.300 + rng.integers(0, 20, size=400) * 0.001
    depth.reshape(-1)[pix[0]], depth.reshape(-1)[pix[1]] = 0.300, 0.319
    K = np.array([[120.0, 0., 31.5], [0., 121.0, 23.5], [0., 0., 1.]])
    pts = A.rgbd_to_point_cloud_no_depth(K, depth)
    assert np.array_equal(pts, oracle.backproject_no_depth((120.0, 121.0, 31.5, 23.5), depth)) and pts.shape == (20, 400, 3)
    rad = rng.uniform(0.1, 0.6, size=400).astype(np.float32)
    ref, votes, info, peak, _, _ = oracle.accumulator_3d_no_depth(pts, rad, return_all=True)
    center, gv, res = A.Accumulator_3D_no_depth(pts, rad, None, return_votes=True)
    assert res.dim == info.dim and res.peak_votes == peak and np.array_equal(gv, votes)
    assert np.array_equal(center, ref)
    for mode in ("auto", "full"):
        assert np.array_equal(A.Accumulator_3D_no_depth(pts, rad, None, engine=mode), ref), mode
    with pytest.raises(IndexError):
        A.Accumulator_3D_no_depth(np.zeros((2, 2, 3)) + 0.1, np.array([0.1, 0.1], dtype=np.float32), None)
    with pytest.raises(ValueError):
        A.rgbd_to_point_cloud_no_depth(K, np.zeros((4, 4)))
