"""Seeded fuzz of rcv_accumulate_3d: both arg-max engines (exact peak search, LDS rasteriser) against each other and, where
the cloud is small enough for it, against the oracle's vote grid -- on the shapes the reference's semantics make awkward:
a handful of points, coincident points (every shell voxel ties), radii that round to zero (nothing votes), radii on x.5
(half-even rounding), points on voxel centres (d2 exactly on a threshold), far-apart clusters (large, mostly empty grids),
outliers, float32 and float64 radii, 2.5 / 5 / 10 mm voxels.  Everything compared is integer or bit-exact float64.

RCV_FUZZ_CASES=<n> runs more cases (default 160, about a minute on the GPU box)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = int(os.environ.get("RCV_FUZZ_CASES", "160"))


def _surface(rng, n, centre):
    u = rng.normal(size=(n, 3))
    u[:, 2] = -np.abs(u[:, 2])
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    return centre + u * np.array([0.12, 0.10, 0.08]) * rng.uniform(0.97, 1.0, size=(n, 1))


def make_case(rng, case):
    """-> (xyz [N,3] metres float64, radial [M] decimetres, acc_unit mm, tag)"""
    kind = case % 10
    unit = float(rng.choice([5.0, 5.0, 5.0, 2.5, 10.0]))
    vox = unit / 1000.0  # metres per voxel
    centre = np.array([rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2), rng.uniform(0.5, 1.2)])
    dtype = np.float32 if rng.random() < 0.6 else np.float64
    if kind == 0:  # a handful of points, arbitrary radii
        n = int(rng.integers(1, 6))
        xyz = centre + rng.normal(size=(n, 3)) * 0.02
        rad = rng.uniform(0.02, 1.5, size=n)
        tag = "few"
    elif kind == 1:  # coincident points, one radius: the whole shell ties
        n = int(rng.integers(2, 40))
        xyz = np.repeat(centre[None], n, axis=0)
        rad = np.full(n, rng.uniform(0.1, 0.6))
        tag = "coincident"
    elif kind == 2:  # radii that round to zero voxels for some / all points
        n = int(rng.integers(3, 200))
        xyz = centre + rng.normal(size=(n, 3)) * 0.01
        rad = rng.uniform(0.0, 0.49 * vox * 10, size=n)
        if rng.random() < 0.5:
            rad[: max(1, n // 3)] = rng.uniform(0.05, 0.3, size=max(1, n // 3))
        tag = "zero-radii"
    elif kind == 3:  # radii exactly on x.5 voxels (np.around is half-to-even)
        n = int(rng.integers(20, 600))
        xyz = _surface(rng, n, centre)
        k = rng.integers(1, 40, size=n)
        rad = (k + 0.5) * vox * 10
        tag = "half-radii"
    elif kind == 4:  # points on a voxel lattice: squared distances hit thresholds exactly
        n = int(rng.integers(5, 300))
        xyz = (np.round(centre / vox) + rng.integers(-12, 13, size=(n, 3))) * vox
        rad = rng.integers(1, 15, size=n) * vox * 10
        tag = "lattice"
    elif kind == 5:  # two clusters far apart: a large, mostly empty grid
        n = int(rng.integers(50, 1500))
        half = n // 2
        xyz = np.concatenate([centre + rng.normal(size=(half, 3)) * 0.01,
                              centre + np.array([rng.uniform(0.3, 0.7), 0.0, 0.1]) + rng.normal(size=(n - half, 3)) * 0.01])
        rad = rng.uniform(0.05, 0.8, size=n)
        tag = "two-clusters"
    elif kind == 6:  # the real input: a surface patch, radii = distance to a keypoint + noise, some outliers
        n = int(rng.integers(200, 12000))
        xyz = _surface(rng, n, centre)
        kp = centre + rng.uniform(-0.25, 0.25, size=3)
        rad = np.linalg.norm(xyz - kp, axis=1) * 10 + rng.normal(size=n) * float(rng.choice([0.0, 0.02, 0.1]))
        m = n // int(rng.choice([5, 20, 1000]))
        rad[:m] = rng.uniform(0.05, 3.0, size=m)
        tag = "surface"
    elif kind == 7:  # more radii than points (the seam's misalignment quirk hands over len(radial) >= len(xyz))
        n = int(rng.integers(10, 800))
        xyz = _surface(rng, n, centre)
        rad = rng.uniform(0.1, 2.0, size=n + int(rng.integers(1, 50)))
        tag = "extra-radii"
    elif kind == 8:  # negative and huge radii mixed in
        n = int(rng.integers(10, 400))
        xyz = centre + rng.normal(size=(n, 3)) * 0.03
        rad = rng.uniform(-0.5, 2.5, size=n)
        tag = "negative-radii"
    else:  # collinear / coplanar points
        n = int(rng.integers(10, 500))
        t = rng.uniform(-0.1, 0.1, size=(n, 1))
        xyz = centre + t * rng.normal(size=(1, 3))
        if rng.random() < 0.5:
            xyz = xyz + rng.uniform(-0.1, 0.1, size=(n, 1)) * np.array([[0.0, 1.0, 0.0]])
        rad = rng.uniform(0.1, 1.5, size=n)
        tag = "flat"
    return np.ascontiguousarray(xyz, dtype=np.float64), np.ascontiguousarray(rad.astype(dtype)), unit, tag


def _fields(r):
    return (r.status, r.dim, r.zero_boundary, r.peak_votes, r.n_ties, tuple(r.first_idx[:]), tuple(r.center_mm[:]), r.n_points, r.n_radii)


def test_fuzz_both_engines_and_the_oracle(oracle):
    import torch

    from rcvpose_amd import AccumulatorSpace as A

    rng = np.random.default_rng(20261003)
    seen, with_oracle, errors = {}, 0, 0
    for case in range(CASES):
        xyz, rad, unit, tag = make_case(rng, case)
        n, m = len(xyz), len(rad)
        ws = A.AccWorkspace(max(n, m))
        xyz_d, rad_d = torch.from_numpy(xyz).cuda(), torch.from_numpy(rad).cuda()
        out = {}
        for mode in ("auto", "full"):
            A.accumulate_on_device(xyz_d, rad_d, None, n, m, ws, acc_unit=unit, engine=mode)
            out[mode] = _fields(ws.read_result())
        assert out["auto"] == out["full"], (case, tag, n, unit, out)
        status, dim, zb, peak, ties, first, centre_mm = out["auto"][:7]
        seen[tag] = seen.get(tag, 0) + 1
        # the oracle's sparse grid costs O(votes + D^3): keep it to moderate grids
        if status == 0 and dim <= 400:
            ref_center, votes, info, ref_peak = oracle.accumulator_3d(xyz, rad, acc_unit=unit, sparse=True, return_all=True)
            flat = np.flatnonzero(votes.ravel() == ref_peak)
            f0 = np.unravel_index(flat[0], votes.shape)
            assert (dim, zb, peak, ties, first) == (info.dim, info.zero_boundary, ref_peak, len(flat), tuple(int(v) for v in f0)), (case, tag)
            assert np.array_equal(np.array(centre_mm), ref_center[0]), (case, tag)
            with_oracle += 1
        elif status != 0:
            errors += 1
            with pytest.raises((ValueError, IndexError, MemoryError)):
                oracle.accumulator_3d(xyz, rad, acc_unit=unit, sparse=True, return_all=True)
    print("fuzz: %d cases %s, %d checked against the oracle, %d error cases" % (CASES, seen, with_oracle, errors))
    assert with_oracle >= CASES // 2


def test_fuzz_batched_entry_point_item_by_item():
    """rcv_accumulate_3d_batch on batches of unequal items (a few points next to thousands, an empty item, float32 radii,
    several voxel sizes across batches): every item's record equals the one-item entry point's on the same lists."""
    import ctypes

    import torch

    from rcvpose_amd import AccumulatorSpace as A
    from rcvpose_amd import _hip

    L = _hip.lib()
    rng = np.random.default_rng(77)
    rsz = ctypes.sizeof(_hip.AccResult)
    for batch in range(max(4, CASES // 20)):
        Bf, G = int(rng.integers(1, 5)), int(rng.integers(1, 4))
        unit = float(rng.choice([5.0, 5.0, 2.5, 10.0]))
        items = []
        for it in range(Bf * G):
            xyz, rad, _, tag = make_case(rng, int(rng.integers(0, 10)))
            if rng.random() < 0.08:
                xyz, rad = xyz[:0], rad[:0]  # nothing passed the mask
            items.append((xyz, rad.astype(np.float32)))
        cap = max(1, max(max(len(x), len(r)) for x, r in items))
        n_items = Bf * G
        xyz_b = torch.zeros((n_items, cap, 3), dtype=torch.float64, device="cuda")
        rad_b = torch.zeros((n_items, cap), dtype=torch.float32, device="cuda")
        cnt = torch.zeros((n_items, 2), dtype=torch.int32)
        for i, (x, r) in enumerate(items):
            xyz_b[i, : len(x)] = torch.from_numpy(x).cuda()
            rad_b[i, : len(r)] = torch.from_numpy(r).cuda()
            cnt[i, 0], cnt[i, 1] = len(x), len(r)
        cnt = cnt.cuda()
        ws_bytes = int(L.rcv_acc_workspace_bytes_min(cap))
        ws = torch.empty((n_items, ws_bytes), dtype=torch.uint8, device="cuda")
        res = torch.zeros((n_items, rsz), dtype=torch.uint8, device="cuda")
        S = lambda t, per: _hip.Strided(t.data_ptr(), G * per, per)  # buffers indexed [frame * G + keypoint]
        _hip.check(L.rcv_accumulate_3d_batch(S(xyz_b, cap * 24), S(rad_b, cap * 4), 0, S(cnt, 8), cap, unit, _hip.VOTE_ENGINES["auto"],
                                             S(res, rsz), S(ws, ws_bytes), ws_bytes, Bf, G, _hip.current_stream()), "rcv_accumulate_3d_batch")
        raw = res.cpu().numpy()
        for i, (x, r) in enumerate(items):
            got = _fields(_hip.AccResult.from_buffer_copy(raw[i].tobytes()))
            if len(x) == 0 or len(r) == 0:
                assert got[0] == _hip.RCV_E_EMPTY, (batch, i, got)
                continue
            one = A.AccWorkspace(max(len(x), len(r)))
            A.accumulate_on_device(xyz_b[i], rad_b[i], None, len(x), len(r), one, acc_unit=unit, engine="auto")
            assert got == _fields(one.read_result()), (batch, i, len(x), len(r), unit)


def test_fuzz_seam_against_oracle(oracle):
    """rcv_seam_compact (modes 0 and 1) on random small images of odd sizes: segmentation scores around the 0.8 threshold
    (and exactly on it), radii around max_radius (and exactly on it), NaN / inf entries, depth holes inside the mask (the
    reference's misaligned lists), empty masks, full masks.  Point list (float64) and radii list (float32) bit for bit,
    in np.nonzero's row-major order."""
    import torch

    from rcvpose_amd import AccumulatorSpace as A
    from rcvpose_amd import _hip

    rng = np.random.default_rng(4242)
    K4 = [572.4114, 573.57043, 325.2611, 242.04899]
    Km = np.array([[K4[0], 0, K4[2]], [0, K4[1], K4[3]], [0, 0, 1]])
    for case in range(max(30, CASES // 4)):
        H, W = int(rng.integers(1, 70)), int(rng.integers(1, 90))
        seg = rng.uniform(0.5, 1.1, size=(H, W)).astype(np.float32)
        rad = rng.uniform(0.0, 3.0, size=(H, W)).astype(np.float32)
        depth = rng.integers(300, 2000, size=(H, W)).astype(np.uint16)
        rmax = float(np.float32(rng.uniform(0.5, 3.2)))
        style = case % 6
        if style == 1:
            seg[:] = 0.0                                  # nothing selected
        elif style == 2:
            seg[:] = 1.0; rad[:] = rmax                   # everything selected, radii exactly on the bound (<=)
        elif style == 3:
            seg[rng.random((H, W)) < 0.3] = np.float32(0.8)   # exactly on the threshold (> fails)
            rad[rng.random((H, W)) < 0.1] = np.nan
            seg[rng.random((H, W)) < 0.05] = np.nan
            rad[rng.random((H, W)) < 0.05] = np.inf
        depth[rng.random((H, W)) < (0.3 if style >= 4 else 0.05)] = 0   # holes
        d_seg, d_rad = torch.from_numpy(seg).cuda(), torch.from_numpy(rad).cuda()
        d_dep = torch.from_numpy(depth.view(np.int16)).cuda()
        ws = A.seam_compact(d_seg, d_rad, d_dep, rmax, K=Km)
        n, m = [int(v) for v in ws.counts.cpu()]
        xyz, rl = oracle.seam(seg, rad, depth, rmax, K=K4)
        assert (n, m) == (len(xyz), len(rl)), (case, H, W, style)
        assert np.array_equal(ws.xyz[:n].cpu().numpy(), xyz) and np.array_equal(ws.radial[:m].cpu().numpy(), rl, equal_nan=True), (case, style)
        # mode 1: the radial-only mask of the pre-computed-map path (reference :612-619)
        ws = A.seam_compact(None, d_rad, d_dep, rmax, K=Km, mode=1)
        n, m = [int(v) for v in ws.counts.cpu()]
        xyz, rl = oracle.seam_radial_only(rad, depth, rmax, K=K4)
        assert (n, m) == (len(xyz), len(rl)), (case, H, W, style, "radial-only")
        assert np.array_equal(ws.xyz[:n].cpu().numpy(), xyz) and np.array_equal(ws.radial[:m].cpu().numpy(), rl, equal_nan=True), (case, style)
