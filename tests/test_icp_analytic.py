"""SURVEY f-3: the restated Open3D calls against closed-form fixtures (tests/golden/make_icp_analytic.py): configurations
whose result follows from the definition of point-to-point ICP, not from a restatement of its implementation.  The CPU
restatement (oracle/icp_oracle.py) is checked here without a GPU, the HIP implementation in the -m gpu test below.
Parity with Open3D 0.14.1 itself stays unpinned (the library cannot be run here)."""
import os

import numpy as np
import pytest

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "icp_analytic.npz"))
NAMES = [str(n) for n in G["names"]]


def _check(i, T, fitness, rmse, n_corr, iterations):
    src, tgt, init, Texp = G["src_%d" % i], G["tgt_%d" % i], G["init_%d" % i], G["T_%d" % i]
    max_dist, fit_exp, rmse0, n_exp = G["scalars_%d" % i]
    name = NAMES[i]
    assert n_corr == int(n_exp) and abs(fitness - fit_exp) < 1e-12, name
    R = T[:3, :3]
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-9) and abs(np.linalg.det(R) - 1.0) < 1e-9, name   # a proper rotation
    if name == "out_of_reach":
        assert iterations == 0 and np.array_equal(T, init)
        return
    assert rmse < 1e-8, (name, rmse)                                   # every point lands on its partner
    moved = src @ T[:3, :3].T + T[:3, 3]
    near = np.linalg.norm(moved[:, None, :] - tgt[None, :, :], axis=2).min(axis=1)
    assert np.sort(near)[: int(n_exp)].max() < 1e-8, name
    if not np.isnan(Texp).any():
        assert np.allclose(T, Texp, rtol=0, atol=1e-9), (name, np.abs(T - Texp).max())
    # convergence rule (|d fitness| and |d rmse| below the tolerances after an update): the update that reaches rmse 0 is
    # followed by exactly one more (an identity update) unless the start was already perfect
    assert iterations == (1 if rmse0 == 0.0 else 2), (name, iterations)


@pytest.mark.parametrize("i", range(len(NAMES)), ids=NAMES)
def test_cpu_restatement_on_closed_form_cases(i):
    from oracle import icp_oracle

    max_dist = float(G["scalars_%d" % i][0])
    r = icp_oracle.registration_icp(G["src_%d" % i], G["tgt_%d" % i], max_dist, G["init_%d" % i])
    _check(i, r["transformation"], r["fitness"], r["inlier_rmse"], r["n_correspondences"], r["iterations"])


def test_cpu_nn_distance_closed_form():
    from oracle import icp_oracle

    d, _ = icp_oracle.nn_distance(G["nn_a"], G["nn_b"])
    assert np.allclose(d, G["nn_dist"], rtol=0, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("i", range(len(NAMES)), ids=NAMES)
def test_gpu_icp_on_closed_form_cases(i):
    from rcvpose_amd.registration import registration_icp

    r = registration_icp(G["src_%d" % i], G["tgt_%d" % i], float(G["scalars_%d" % i][0]), G["init_%d" % i])
    _check(i, r.transformation, r.fitness, r.inlier_rmse, r.n_correspondences, r.iterations)


@pytest.mark.gpu
def test_gpu_nn_distance_closed_form():
    from rcvpose_amd.registration import compute_point_cloud_distance

    assert np.allclose(compute_point_cloud_distance(G["nn_a"], G["nn_b"]), G["nn_dist"], rtol=0, atol=1e-12)
