#!/usr/bin/env python3
"""Closed-form fixtures for the two Open3D calls of the evaluation:
`compute_point_cloud_distance` and `registration_icp(..., TransformationEstimationPointToPoint(), criteria)`.

Open3D 0.14.1 is not installable here and the reference holds no fixture for these calls, so parity with the
library itself stays UNPINNED.  What this file pins instead does not depend on anybody's restatement of the
algorithm: configurations whose answer follows from the DEFINITION of point-to-point ICP --

  * every source point's nearest target point is its true partner (the motion is smaller than half the smallest
    point spacing), so the first least-squares update IS the rigid motion that was applied: transformation known in
    closed form, fitness = (points within reach) / (source points), inlier_rmse = 0 afterwards, and the initial
    rmse is sqrt(mean |R s + t - s|^2);
  * points out of reach do not take part (fitness is a known fraction, the transformation is unchanged);
  * degenerate sets (collinear, coplanar) must still come back as a proper rotation (det = +1) that maps every
    source point onto its partner;
  * identical clouds / nothing in reach.

    python tests/golden/make_icp_analytic.py      ->  tests/golden/icp_analytic.npz  (inputs + expected values)
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def rot(axis, deg):
    axis = np.asarray(axis, dtype=np.float64)
    axis = axis / np.linalg.norm(axis)
    a = np.deg2rad(deg)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * (K @ K)


def lattice(nx, ny, nz, h, jitter, rng):
    g = np.stack(np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij"), -1).reshape(-1, 3).astype(np.float64) * h
    return g + rng.uniform(-jitter, jitter, size=g.shape)


def main():
    rng = np.random.default_rng(31415)
    cases = []

    def add(name, src, tgt, max_dist, init, T, fitness, rmse0, n_corr, note):
        cases.append(dict(name=name, src=src, tgt=tgt, max_dist=max_dist, init=init, T=T, fitness=fitness, rmse0=rmse0, n_corr=n_corr,
                          note=note))

    def rigid(R, t):
        T = np.eye(4)
        T[:3, :3] = R
        T[:3, 3] = t
        return T

    # 1. small rigid motion of a jittered lattice (spacing 10, displacement < 2 everywhere): true partners are nearest
    S = lattice(6, 5, 4, 10.0, 1.0, rng) - np.array([25.0, 20.0, 15.0])
    R, t = rot([1, 2, 3], 1.5), np.array([0.6, -0.4, 0.5])
    Tg = S @ R.T + t
    assert np.linalg.norm(Tg - S, axis=1).max() < 2.5
    add("rigid_small", S, Tg, 5.0, np.eye(4), rigid(R, t), 1.0, float(np.sqrt((np.linalg.norm(Tg - S, axis=1) ** 2).mean())), len(S),
        "exact partners; first update = the applied motion")
    # 2. the same with 15 source points far out of reach: they never correspond
    far = rng.uniform(500, 600, size=(15, 3))
    S2 = np.vstack([S, far])
    add("rigid_with_outliers", S2, Tg, 5.0, np.eye(4), rigid(R, t), len(S) / len(S2),
        float(np.sqrt((np.linalg.norm(Tg - S, axis=1) ** 2).mean())), len(S), "fitness = 120/135; outliers do not move the fit")
    # 3. pure translation of a regular lattice (spacing 8, shift 3 < 4)
    G = lattice(5, 5, 5, 8.0, 0.0, rng)
    d = np.array([3.0, 0.0, 0.0])
    add("translation_lattice", G, G + d, 3.5, np.eye(4), rigid(np.eye(3), d), 1.0, 3.0, len(G), "rmse0 = 3 exactly")
    # 4. start from a non-identity initial transform: init followed by the residual motion
    Ri, ti = rot([0, 0, 1], 20.0), np.array([100.0, -50.0, 30.0])
    world = S @ Ri.T + ti                      # source as placed by init
    Rr, tr = rot([1, 0, 1], 1.0), np.array([0.3, 0.5, -0.2])
    c = world.mean(0)
    Tg4 = (world - c) @ Rr.T + c + tr          # small motion about the centroid
    T4 = rigid(Rr, c + tr - Rr @ c) @ rigid(Ri, ti)
    assert np.linalg.norm(Tg4 - world, axis=1).max() < 2.5
    add("with_init", S, Tg4, 5.0, rigid(Ri, ti), T4, 1.0, float(np.sqrt((np.linalg.norm(Tg4 - world, axis=1) ** 2).mean())), len(S),
        "result = residual motion composed with init")
    # 5. collinear points: the rotation about the line is free; every source point must land on its partner, det R = +1
    line = np.outer(np.arange(12) * 7.0, np.array([1.0, 2.0, -1.0]) / np.sqrt(6.0))
    R5, t5 = rot([0, 1, 0], 2.0), np.array([0.5, 0.2, -0.3])
    add("collinear", line, line @ R5.T + t5, 3.0, np.eye(4), None, 1.0,
        float(np.sqrt((np.linalg.norm(line @ R5.T + t5 - line, axis=1) ** 2).mean())), len(line), "transformation not unique: check the mapping")
    # 6. coplanar points (zero singular value): proper rotation required
    plane = np.stack(np.meshgrid(np.arange(6) * 9.0, np.arange(5) * 9.0, indexing="ij"), -1).reshape(-1, 2)
    plane = np.hstack([plane, np.zeros((len(plane), 1))]) + np.array([3.0, -4.0, 50.0])
    R6, t6 = rot([1, 1, 0], 2.0), np.array([-0.7, 0.4, 0.6])
    add("coplanar", plane, plane @ R6.T + t6, 4.0, np.eye(4), rigid(R6, t6), 1.0,
        float(np.sqrt((np.linalg.norm(plane @ R6.T + t6 - plane, axis=1) ** 2).mean())), len(plane), "planar set: unique proper rotation")
    # 7. identical clouds; 8. nothing within reach
    add("identical", S, S.copy(), 1.0, np.eye(4), np.eye(4), 1.0, 0.0, len(S), "identity")
    add("out_of_reach", S, S + 1000.0, 1.0, np.eye(4), np.eye(4), 0.0, 0.0, 0, "no correspondences: init comes back, zero updates")

    out = {"names": np.array([c["name"] for c in cases]), "notes": np.array([c["note"] for c in cases])}
    for i, c in enumerate(cases):
        out["src_%d" % i], out["tgt_%d" % i], out["init_%d" % i] = c["src"], c["tgt"], c["init"]
        out["T_%d" % i] = c["T"] if c["T"] is not None else np.full((4, 4), np.nan)
        out["scalars_%d" % i] = np.array([c["max_dist"], c["fitness"], c["rmse0"], c["n_corr"]], dtype=np.float64)
    # nearest-neighbour distances with known answers: lattice against the shifted lattice (distance = |shift|), and against a
    # lattice with every other point removed along x (removed points are 8 - 3 = 5 from the next survivor... of the shifted copy)
    out["nn_a"], out["nn_b"] = G, G + d
    out["nn_dist"] = np.full(len(G), 3.0)
    np.savez_compressed(os.path.join(HERE, "icp_analytic.npz"), **out)
    print("wrote %d ICP cases + 1 nearest-neighbour case" % len(cases))


if __name__ == "__main__":
    main()
