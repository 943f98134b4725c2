"""GPU nearest-neighbour distances and point-to-point ICP against the numpy/scipy restatement
of Open3D's algorithm in oracle/icp_oracle.py (parity unpinned against Open3D itself, see that file)."""
import numpy as np
import pytest

from oracle import icp_oracle

pytestmark = pytest.mark.gpu


def _rot(rng, max_deg):
    ax = rng.normal(size=3)
    ax /= np.linalg.norm(ax)
    a = np.deg2rad(rng.uniform(-max_deg, max_deg))
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    return np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * K @ K


def _model(rng, n):
    u = rng.normal(size=(n, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    return u * np.array([120.0, 100.0, 80.0]) * rng.uniform(0.9, 1.0, size=(n, 1))  # mm


def test_nn_distance_matches_kdtree():
    from rcvpose_amd.registration import compute_point_cloud_distance

    rng = np.random.default_rng(3)
    for na, nb in ((1, 1), (257, 1025), (5000, 3333), (1000, 7)):
        a, b = rng.normal(size=(na, 3)) * 100, rng.normal(size=(nb, 3)) * 100
        d, i = compute_point_cloud_distance(a, b, return_index=True)
        dr, ir = icp_oracle.nn_distance(a, b)
        assert np.allclose(d, dr, rtol=0, atol=1e-10)
        assert np.array_equal(i, ir) or np.allclose(np.linalg.norm(a - b[i], axis=1), dr, atol=1e-10)
    assert compute_point_cloud_distance(a, np.zeros((0, 3))).shape == (0,)


@pytest.mark.parametrize("seed,noise,max_deg,shift", [(0, 0.0, 3.0, 4.0), (1, 0.5, 5.0, 8.0), (2, 1.0, 2.0, 3.0)])
def test_icp_matches_oracle(seed, noise, max_deg, shift):
    from rcvpose_amd.registration import ICPConvergenceCriteria, registration_icp

    rng = np.random.default_rng(seed)
    model = _model(rng, 3000)
    Rg, tg = _rot(rng, 40), np.array([20.0, -30.0, 900.0])
    vis = (model @ Rg.T)[:, 2] < 10  # the camera-facing part is observed
    scene = model[vis] @ Rg.T + tg + rng.normal(size=(vis.sum(), 3)) * noise
    init = np.eye(4)
    init[:3, :3] = _rot(rng, max_deg) @ Rg
    init[:3, 3] = tg + rng.normal(size=3) * shift
    results = []
    for crit in (ICPConvergenceCriteria(), ICPConvergenceCriteria(relative_fitness=10.0, relative_rmse=10.0, max_iteration=30)):
        got = registration_icp(model, scene, 15.0, init, crit)
        results.append(got)
        ref = icp_oracle.registration_icp(model, scene, 15.0, init, crit.relative_fitness, crit.relative_rmse, crit.max_iteration)
        assert got.iterations == ref["iterations"] and got.n_correspondences == ref["n_correspondences"]
        assert abs(got.fitness - ref["fitness"]) < 1e-12 and abs(got.inlier_rmse - ref["inlier_rmse"]) < 1e-7
        assert np.allclose(got.transformation, ref["transformation"], rtol=0, atol=1e-6)
    # the reference's LINEMOD criteria (the defaults) iterate to convergence; its LM-O criteria (thresholds in mm used as
    # relative tolerances, AccumulatorSpace.py:935-937) stop after the first update
    assert results[1].iterations == 1 and results[0].iterations > 1
    d0 = icp_oracle.nn_distance(model @ init[:3, :3].T + init[:3, 3], scene)[0]
    T = results[0].transformation
    d1 = icp_oracle.nn_distance(model @ T[:3, :3].T + T[:3, 3], scene)[0]
    assert np.median(d1) < np.median(d0)


def test_icp_degenerate_inputs():
    from rcvpose_amd.registration import registration_icp

    rng = np.random.default_rng(9)
    a = rng.normal(size=(100, 3))
    far = registration_icp(a, a + 1000.0, 1.0)  # nothing within reach: the initial transform comes back
    assert far.n_correspondences == 0 and far.iterations == 0 and np.array_equal(far.transformation, np.eye(4))
    same = registration_icp(a, a, 0.5)
    assert same.fitness == 1.0 and same.inlier_rmse < 1e-9 and np.allclose(same.transformation, np.eye(4), atol=1e-9)


@pytest.mark.parametrize("seed,reach", [(3, 4.0), (4, 9.0), (5, 60.0)])
def test_icp_grid_search_equals_exhaustive_search(seed, reach, monkeypatch):
    """The hashed-grid neighbour search (targets binned at the correspondence distance; taken when that bins the cloud finely)
    must give the exhaustive search's correspondences, hence bit-identical ICP results -- also with duplicate target points
    (ties go to the lower index in both) and against the oracle."""
    from rcvpose_amd.registration import ICPConvergenceCriteria, registration_icp

    rng = np.random.default_rng(seed)
    model = _model(rng, 6000)
    Rg, tg = _rot(rng, 40), np.array([20.0, -30.0, 900.0])
    vis = (model @ Rg.T)[:, 2] < 25
    scene = model[vis] @ Rg.T + tg + rng.normal(size=(vis.sum(), 3)) * 0.3
    scene = np.concatenate([scene, scene[:200], rng.uniform(-300, 300, size=(500, 3)) + tg])  # duplicates + clutter
    assert scene.shape[0] >= 2048
    init = np.eye(4)
    init[:3, :3] = _rot(rng, 2.0) @ Rg
    init[:3, 3] = tg + rng.normal(size=3) * 2.0
    crit = ICPConvergenceCriteria()
    got = registration_icp(model, scene, reach, init, crit)
    monkeypatch.setenv("RCV_ICP_NO_GRID", "1")
    full = registration_icp(model, scene, reach, init, crit)
    monkeypatch.delenv("RCV_ICP_NO_GRID")
    assert got.iterations == full.iterations and got.n_correspondences == full.n_correspondences
    assert got.fitness == full.fitness and got.inlier_rmse == full.inlier_rmse
    assert np.array_equal(got.transformation, full.transformation)
    ref = icp_oracle.registration_icp(model, scene, reach, init, crit.relative_fitness, crit.relative_rmse, crit.max_iteration)
    assert got.iterations == ref["iterations"] and got.n_correspondences == ref["n_correspondences"]
    assert np.allclose(got.transformation, ref["transformation"], rtol=0, atol=1e-6)
