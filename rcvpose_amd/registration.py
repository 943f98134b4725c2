"""Nearest-neighbour distances and point-to-point ICP on the GPU.

Host-side mirror of the two Open3D calls the reference's evaluation makes after the keypoints
(AccumulatorSpace.py:689-736): `PointCloud.compute_point_cloud_distance` and
`pipelines.registration.registration_icp(..., TransformationEstimationPointToPoint(), criteria)`.  Same argument meaning and
result fields (`transformation`, `fitness`, `inlier_rmse`); the work is done by librcvpose_hip.so (rcv_nn_distance,
rcv_icp_point_to_point) -- there is no CPU path.
"""
import ctypes

import numpy as np

from . import _hip


class ICPConvergenceCriteria:
    """open3d.pipelines.registration.ICPConvergenceCriteria (defaults of Open3D 0.14)."""

    def __init__(self, relative_fitness=1e-6, relative_rmse=1e-6, max_iteration=30):
        self.relative_fitness = float(relative_fitness)
        self.relative_rmse = float(relative_rmse)
        self.max_iteration = int(max_iteration)


class RegistrationResult:
    def __init__(self, transformation, fitness, inlier_rmse, n_correspondences, iterations):
        self.transformation = transformation
        self.fitness = fitness
        self.inlier_rmse = inlier_rmse
        self.n_correspondences = n_correspondences
        self.iterations = iterations

    def __repr__(self):
        return "RegistrationResult(fitness=%.6f, inlier_rmse=%.6f, correspondences=%d, iterations=%d)" % (
            self.fitness, self.inlier_rmse, self.n_correspondences, self.iterations)


def _dev_points(a):
    import torch

    if isinstance(a, torch.Tensor):
        t = a.to(device="cuda", dtype=torch.float64).contiguous()
    else:
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()
    if t.dim() != 2 or t.shape[1] != 3:
        raise ValueError("points must have shape [n,3]")
    return t


def compute_point_cloud_distance(source, target, return_index=False):
    """For every point of `source` the distance to its nearest point of `target` (float64 [n])."""
    import torch

    lib = _hip.lib()
    a, b = _dev_points(source), _dev_points(target)
    if b.shape[0] == 0:
        return np.zeros((0,), dtype=np.float64)  # Open3D returns an empty vector when the other cloud is empty
    dist = torch.empty((a.shape[0],), dtype=torch.float64, device="cuda")
    idx = torch.empty((a.shape[0],), dtype=torch.int32, device="cuda") if return_index else None
    _hip.check(lib.rcv_nn_distance(_hip.ptr(a), a.shape[0], _hip.ptr(b), b.shape[0], _hip.ptr(dist),
                                   _hip.ptr(idx) if idx is not None else None, _hip.current_stream()), "rcv_nn_distance")
    if return_index:
        return dist.cpu().numpy(), idx.cpu().numpy()
    return dist.cpu().numpy()


def registration_icp(source, target, max_correspondence_distance, init=None, criteria=None):
    """Point-to-point ICP of `source` onto `target` (both [n,3]); returns a RegistrationResult."""
    import torch

    lib = _hip.lib()
    criteria = criteria or ICPConvergenceCriteria()
    a, b = _dev_points(source), _dev_points(target)
    T0 = np.eye(4) if init is None else np.ascontiguousarray(init, dtype=np.float64)
    if T0.shape != (4, 4):
        raise ValueError("init must be a 4x4 matrix")
    if a.shape[0] == 0 or b.shape[0] == 0:
        return RegistrationResult(T0.copy(), 0.0, 0.0, 0, 0)
    ws = torch.empty((lib.rcv_icp_workspace_bytes(a.shape[0]),), dtype=torch.uint8, device="cuda")
    out = _hip.IcpResult()
    t0 = (ctypes.c_double * 16)(*T0.reshape(-1))
    _hip.check(lib.rcv_icp_point_to_point(_hip.ptr(a), a.shape[0], _hip.ptr(b), b.shape[0], float(max_correspondence_distance), t0,
                                          criteria.relative_fitness, criteria.relative_rmse, criteria.max_iteration,
                                          ctypes.byref(out), _hip.ptr(ws), ws.numel(), _hip.current_stream()),
               "rcv_icp_point_to_point")
    return RegistrationResult(np.array(out.transformation[:], dtype=np.float64).reshape(4, 4), out.fitness, out.inlier_rmse,
                              out.n_correspondences, out.iterations)
