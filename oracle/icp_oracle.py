"""numpy/scipy restatement of Open3D 0.14's point-to-point ICP -- TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED against Open3D itself: open3d==0.14.1 (rcvpose.yml:176) is a third-party dependency that is neither in
/root/reference nor installable here, and the reference has no test or fixture for these calls.  This file restates the
published algorithm of `registration_icp` with `TransformationEstimationPointToPoint` (Registration.cpp:
RegistrationICP / GetRegistrationResultAndCorrespondences; TransformationEstimation.cpp: Eigen::umeyama without scaling)
at the reference's call sites AccumulatorSpace.py:689-736.  Only tests/ may import it.
"""
import numpy as np
from scipy.spatial import cKDTree


def nn_distance(a, b):
    d, i = cKDTree(np.asarray(b, dtype=np.float64)).query(np.asarray(a, dtype=np.float64))
    return d, i


def _evaluate(cur, tree, max_dist):
    d, j = tree.query(cur)
    ok = d <= max_dist
    n = int(ok.sum())
    fitness = n / len(cur)
    rmse = float(np.sqrt((d[ok] ** 2).sum() / n)) if n else 0.0
    return ok, j, fitness, rmse


def _kabsch(p, q):
    """Least-squares rigid transform q ~ R p + t (Umeyama, no scaling, reflection corrected)."""
    pm, qm = p.mean(0), q.mean(0)
    H = (q - qm).T @ (p - pm) / len(p)
    U, S, Vt = np.linalg.svd(H)
    D = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        D[2, 2] = -1
    R = U @ D @ Vt
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = qm - R @ pm
    return T


def registration_icp(source, target, max_dist, init=None, relative_fitness=1e-6, relative_rmse=1e-6, max_iteration=30):
    source = np.asarray(source, dtype=np.float64)
    target = np.asarray(target, dtype=np.float64)
    T = np.eye(4) if init is None else np.array(init, dtype=np.float64)
    tree = cKDTree(target)
    cur = source @ T[:3, :3].T + T[:3, 3]
    ok, j, fitness, rmse = _evaluate(cur, tree, max_dist)
    it = 0
    while it < max_iteration and ok.any():
        U = _kabsch(cur[ok], target[j[ok]]) if ok.sum() >= 3 else np.eye(4)
        T = U @ T
        cur = source @ T[:3, :3].T + T[:3, 3]
        pf, pr = fitness, rmse
        ok, j, fitness, rmse = _evaluate(cur, tree, max_dist)
        it += 1
        if abs(pf - fitness) < relative_fitness and abs(pr - rmse) < relative_rmse:
            break
    return dict(transformation=T, fitness=fitness, inlier_rmse=rmse, iterations=it, n_correspondences=int(ok.sum()))
