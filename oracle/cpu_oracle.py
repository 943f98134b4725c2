"""ctypes front-end of oracle/librcv_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
It wraps the plain-C restatement of the reference's voting path (oracle/rcv_oracle.c); see the
header of that file for the reference lines each function follows and for how the oracle is
pinned to the reference (tests/golden, tests/test_oracle_golden.py).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

LINEMOD_K = (572.4114, 573.57043, 325.2611, 242.04899)  # fx, fy, cx, cy; AccumulatorSpace.py:59-61


class GridInfo(ctypes.Structure):
    _fields_ = [
        ("mean", ctypes.c_double * 3),
        ("radius_max", ctypes.c_double),
        ("zero_boundary", ctypes.c_int32),
        ("length", ctypes.c_int32),
        ("dim", ctypes.c_int32),
        ("status", ctypes.c_int32),
    ]


def build(force=False):
    so = os.path.join(_HERE, "librcv_oracle.so")
    src = os.path.join(_HERE, "rcv_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "librcv_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "librcv_oracle.so")
        if not os.path.exists(so):
            build()
        L = ctypes.CDLL(so)
        L.orc_mean_strided.restype = ctypes.c_double
        L.orc_mean_strided.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_long]
        L.orc_backproject.restype = ctypes.c_long
        L.orc_argmax_all.restype = ctypes.c_long
        L.orc_acc_prepare.restype = ctypes.c_int
        L.orc_num_threads.restype = ctypes.c_int
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def num_threads():
    return int(lib().orc_num_threads())


def mean_strided(a2d, col):
    a2d = np.ascontiguousarray(a2d, dtype=np.float64)
    n, s = a2d.shape
    return float(lib().orc_mean_strided(ctypes.c_void_p(a2d.ctypes.data + 8 * col), n, s))


def backproject(depth_i64, K=LINEMOD_K):
    d = np.ascontiguousarray(depth_i64, dtype=np.int64)
    H, W = d.shape
    out = np.empty((H * W, 3), dtype=np.float64)
    Kd = np.asarray(K, dtype=np.float64)
    n = lib().orc_backproject(_p(d), H, W, _p(Kd), _p(out))
    return out[:n].copy()


def seam(sem, radial, depth, rmax_dm, seg_thr=0.8, K=LINEMOD_K):
    """AccumulatorSpace.py:603-610,619 -> (xyz_m [N,3] f64, radial_list [M] f32), M >= N."""
    sem = np.ascontiguousarray(sem, dtype=np.float32)
    radial = np.ascontiguousarray(radial, dtype=np.float32)
    depth = np.ascontiguousarray(depth, dtype=np.uint16)
    H, W = sem.shape
    xyz = np.empty((H * W, 3), dtype=np.float64)
    rl = np.empty((H * W,), dtype=np.float32)
    n = ctypes.c_long()
    m = ctypes.c_long()
    Kd = np.asarray(K, dtype=np.float64)
    lib().orc_seam(_p(sem), _p(radial), _p(depth), H, W, _p(Kd), ctypes.c_double(seg_thr), ctypes.c_double(rmax_dm),
                   _p(xyz), _p(rl), ctypes.byref(n), ctypes.byref(m))
    return xyz[: n.value].copy(), rl[: m.value].copy()


def seam_radial_only(radial_est, depth, rmax_dm, K=LINEMOD_K):
    """AccumulatorSpace.py:612-619 (using_ckpts=False)."""
    radial_est = np.ascontiguousarray(radial_est, dtype=np.float32)
    depth = np.ascontiguousarray(depth, dtype=np.uint16)
    H, W = radial_est.shape
    xyz = np.empty((H * W, 3), dtype=np.float64)
    rl = np.empty((H * W,), dtype=np.float32)
    n = ctypes.c_long()
    Kd = np.asarray(K, dtype=np.float64)
    lib().orc_seam_radial_only(_p(radial_est), _p(depth), H, W, _p(Kd), ctypes.c_double(rmax_dm), _p(xyz), _p(rl),
                               ctypes.byref(n))
    return xyz[: n.value].copy(), rl[: n.value].copy()


def seam_lmo_radial(radial_est, depth, rmax_dm, K=LINEMOD_K):
    """AccumulatorSpace.py:849-851,858-864 (Occlusion-LINEMOD, using_ckpts=False) -> (xyz_m, radial_list) or (None, None) when
    the thresholded map is all zero (the keypoint is skipped, :858)."""
    radial_est = np.ascontiguousarray(radial_est, dtype=np.float32)
    depth = np.ascontiguousarray(depth, dtype=np.uint16)
    H, W = radial_est.shape
    xyz = np.empty((H * W, 3), dtype=np.float64)
    rl = np.empty((H * W,), dtype=np.float32)
    n, m = ctypes.c_long(), ctypes.c_long()
    Kd = np.asarray(K, dtype=np.float64)
    L = lib()
    L.orc_seam_lmo_radial.restype = ctypes.c_int
    ok = L.orc_seam_lmo_radial(_p(radial_est), _p(depth), H, W, _p(Kd), ctypes.c_double(rmax_dm), _p(xyz), _p(rl),
                               ctypes.byref(n), ctypes.byref(m))
    if not ok:
        return None, None
    return xyz[: n.value].copy(), rl[: m.value].copy()


def seam_ycb(sem, radial, depth, factor_depth, K, seg_thr=0.8):
    """AccumulatorSpace.py:1050-1057 (YCB-Video): K = (fx, fy, cx, cy) of the frame's intrinsic matrix ->
    (xyz in depth / factor_depth units, radial_list)."""
    sem = np.ascontiguousarray(sem, dtype=np.float32)
    radial = np.ascontiguousarray(radial, dtype=np.float32)
    depth = np.ascontiguousarray(depth, dtype=np.uint16)
    H, W = sem.shape
    xyz = np.empty((H * W, 3), dtype=np.float64)
    rl = np.empty((H * W,), dtype=np.float32)
    n, m = ctypes.c_long(), ctypes.c_long()
    Kd = np.asarray(K, dtype=np.float64)
    lib().orc_seam_ycb(_p(sem), _p(radial), _p(depth), H, W, _p(Kd), ctypes.c_double(seg_thr), ctypes.c_double(factor_depth),
                       _p(xyz), _p(rl), ctypes.byref(n), ctypes.byref(m))
    return xyz[: n.value].copy(), rl[: m.value].copy()


def prepare(xyz_m, radial_dm, acc_unit=5.0):
    """AccumulatorSpace.py:374-401 -> (points_vox [N,3], radius_vox [M] (as f64), GridInfo)."""
    xyz_m = np.ascontiguousarray(xyz_m, dtype=np.float64)
    if xyz_m.ndim != 2 or xyz_m.shape[1] != 3:
        xyz_m = xyz_m.reshape(-1, 3)
    radial_dm = np.ascontiguousarray(radial_dm)
    if radial_dm.dtype not in (np.float32, np.float64):
        radial_dm = radial_dm.astype(np.float64)
    n, m = len(xyz_m), len(radial_dm)
    pts = np.empty((max(n, 1), 3), dtype=np.float64)
    rv = np.empty((max(m, 1),), dtype=np.float64)
    info = GridInfo()
    rc = lib().orc_acc_prepare(_p(xyz_m), n, _p(radial_dm), int(radial_dm.dtype == np.float64), m,
                               ctypes.c_double(acc_unit), _p(pts), _p(rv), ctypes.byref(info))
    if rc != 0:
        raise ValueError("oracle: reference would raise here (status %d)" % rc)
    return pts[:n], rv[:m], info


def fast_for(points_vox, radius_vox, D, sparse=False, threads=0):
    """AccumulatorSpace.py:325-341 -> uint32 [D,D,D] vote grid."""
    points_vox = np.ascontiguousarray(points_vox, dtype=np.float64)
    radius_vox = np.ascontiguousarray(radius_vox, dtype=np.float64)
    votes = np.zeros((D, D, D), dtype=np.uint32)
    fn = lib().orc_fast_for_sparse if sparse else lib().orc_fast_for
    fn(_p(points_vox), _p(radius_vox), len(points_vox), int(D), _p(votes), int(threads))
    return votes


def argmax_all(votes, cap=1 << 20):
    D = votes.shape[0]
    votes = np.ascontiguousarray(votes, dtype=np.uint32)
    peak = ctypes.c_uint32()
    idx = np.empty((cap,), dtype=np.int64)
    cnt = lib().orc_argmax_all(_p(votes), D, ctypes.byref(peak), _p(idx), cap)
    return int(peak.value), int(cnt), idx[: min(cnt, cap)].copy()


def accumulator_3d(xyz_m, radial_dm, acc_unit=5.0, sparse=False, threads=0, return_all=False):
    """Accumulator_3D, AccumulatorSpace.py:373-419 -> center float64 [M,3] (row 0 in mm)."""
    pts, rv, info = prepare(xyz_m, radial_dm, acc_unit)
    votes = fast_for(pts, rv[: len(pts)], info.dim, sparse=sparse, threads=threads)
    peak, cnt, idx = argmax_all(votes)
    D = info.dim
    center = np.stack([idx // (D * D), (idx // D) % D, idx % D], axis=1).astype(np.float64)
    if info.zero_boundary < 0:
        center = center + info.zero_boundary
    row0 = np.empty(3, dtype=np.float64)
    lib().orc_center_row0(ctypes.c_int64(int(idx[0])), ctypes.byref(info), ctypes.c_double(acc_unit), _p(row0))
    center[0] = row0
    if return_all:
        return center, votes, info, peak
    return center


def peak_without_grid(xyz_m, radial_dm, acc_unit=5.0, probe_idx=None, threads=0):
    """Accumulator_3D's peak for grids too large to hold (1 mm voxels): -> dict(center_mm (row 0 of the reference's
    `center`), peak, n_ties, first_idx (i, j, k), dim, zero_boundary, probe_votes).  Same arithmetic as accumulator_3d
    (prepare + the sparse candidate walk + the reference predicate), plane by plane; see orc_peak_planes."""
    pts, rv, info = prepare(xyz_m, radial_dm, acc_unit)
    D = info.dim
    peak = ctypes.c_uint32()
    first = ctypes.c_int64()
    probe = np.zeros((0,), dtype=np.int64) if probe_idx is None else np.ascontiguousarray(
        [(i * D + j) * D + k for i, j, k in probe_idx], dtype=np.int64)
    pv = np.zeros((max(len(probe), 1),), dtype=np.uint32)
    pts = np.ascontiguousarray(pts, dtype=np.float64)
    rvn = np.ascontiguousarray(rv[: len(pts)], dtype=np.float64)
    L = lib()
    L.orc_peak_planes.restype = ctypes.c_long
    cnt = L.orc_peak_planes(_p(pts), _p(rvn), ctypes.c_long(len(pts)), int(D), ctypes.byref(peak), ctypes.byref(first), _p(probe),
                            ctypes.c_long(len(probe)), _p(pv), int(threads))
    flat = int(first.value)
    row0 = np.empty(3, dtype=np.float64)
    L.orc_center_row0(ctypes.c_int64(flat), ctypes.byref(info), ctypes.c_double(acc_unit), _p(row0))
    return dict(center_mm=row0, peak=int(peak.value), n_ties=int(cnt), first_idx=(flat // (D * D), (flat // D) % D, flat % D),
                dim=int(D), zero_boundary=int(info.zero_boundary), probe_votes=pv[: len(probe)].copy())


def backproject_no_depth(K4, depth_m):
    """rgbd_to_point_cloud_no_depth, AccumulatorSpace.py:100-120: depth float64 [H,W] metres -> pts [Z,N,3]."""
    d = np.ascontiguousarray(depth_m, dtype=np.float64)
    H, W = d.shape
    n = int(np.count_nonzero(d))
    if n == 0:
        raise ValueError("zero-size array to reduction operation")  # zs.min() of an empty array
    nz = d[d != 0]
    Z = int(nz.max() * 1000) + 1 - int(nz.min() * 1000)
    out = np.empty((Z, n, 3), dtype=np.float64)
    Kd = np.asarray(K4, dtype=np.float64)
    nn = ctypes.c_long()
    L = lib()
    L.orc_backproject_no_depth.restype = ctypes.c_long
    z = L.orc_backproject_no_depth(_p(d), H, W, _p(Kd), _p(out), ctypes.c_long(Z * n), ctypes.byref(nn))
    assert z == Z and nn.value == n
    return out


def accumulator_3d_no_depth(xyz_m, radial_dm, return_all=False, threads=0):
    """Accumulator_3D_no_depth, AccumulatorSpace.py:436-479 -> center int64 [M,3] (all rows x5, row 0 shifted by the means)."""
    xyz_m = np.ascontiguousarray(xyz_m, dtype=np.float64)
    if xyz_m.ndim != 3 or xyz_m.shape[2] != 3 or xyz_m.shape[1] < 3:
        raise IndexError("Accumulator_3D_no_depth needs a [Z, N>=3, 3] hypothesis array")  # xyz_mm[:,2] in the reference
    radial_dm = np.ascontiguousarray(radial_dm)
    if radial_dm.dtype not in (np.float32, np.float64):
        radial_dm = radial_dm.astype(np.float64)
    Z, N = xyz_m.shape[:2]
    pts = np.empty((Z, N, 3), dtype=np.float64)
    rv = np.empty((max(len(radial_dm), 1),), dtype=np.float64)
    info = GridInfo()
    rc = lib().orc_nd_prepare(_p(xyz_m), ctypes.c_long(Z), ctypes.c_long(N), _p(radial_dm), int(radial_dm.dtype == np.float64),
                              ctypes.c_long(len(radial_dm)), _p(pts), _p(rv), ctypes.byref(info))
    if rc != 0:
        raise ValueError("oracle: reference would raise here (status %d)" % rc)
    D = info.dim
    votes = np.zeros((D, D, D), dtype=np.uint32)
    lib().orc_fast_for_nd(_p(pts), _p(rv), ctypes.c_long(Z), ctypes.c_long(N), int(D), _p(votes), int(threads))
    peak, cnt, idx = argmax_all(votes)
    center = np.empty((len(idx), 3), dtype=np.int64)
    lib().orc_nd_center(_p(idx), ctypes.c_long(len(idx)), ctypes.byref(info), _p(center))
    if return_all:
        return center, votes, info, peak, pts, rv
    return center


def horn(P1, P2):
    """HornPoseFitting.lmshorn, util/horn.py:75-181 -> 4x4 RT with P2 ~ R P1 + T."""
    P1 = np.ascontiguousarray(P1, dtype=np.float64)
    P2 = np.ascontiguousarray(P2, dtype=np.float64)
    A = np.zeros((4, 4), dtype=np.float64)
    lib().orc_horn(_p(P1), _p(P2), int(len(P1)), _p(A))
    return A
