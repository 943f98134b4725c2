#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE ITSELF.

Run in the build container only (needs /root/reference; the GPU box never sees it):

    python tests/golden/make_golden.py            # all cases
    python tests/golden/make_golden.py acc seam   # a subset

The reference (aaronWool/rcvpose) has no tests and no golden vectors,
so every pin below is produced by importing the reference's own modules in this process:

* numba / open3d / h5py are not installed; they are replaced by inert stubs
  (numba.jit -> identity, prange -> range), so AccumulatorSpace.fast_for
  (AccumulatorSpace.py:325-341) executes as sequential pure Python = the race-free
  semantics of that loop.
* only /root/reference and this directory's parent (for fcn_weights.py) are on sys.path;
  the repo root is NOT, so the same-named modules of the build are never imported here.
* nothing of the reference's text is written to the fixtures: they hold inputs and the
  outputs the reference computed from them.  The mask/back-projection seam lives inline in
  estimate_6d_pose_lm (AccumulatorSpace.py:602-619); it is executed here by reading those
  lines from the reference file at generation time and exec-ing them.
"""
import json
import os
import sys
import tempfile
import textwrap
import time
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _install_stubs():
    def _identity_decorator(*args, **kwargs):
        if len(args) == 1 and callable(args[0]) and not kwargs:
            return args[0]
        return lambda fn: fn

    numba = types.ModuleType("numba")
    numba.jit = _identity_decorator
    numba.njit = _identity_decorator
    numba.prange = range
    cuda = types.ModuleType("numba.cuda")
    cuda.jit = _identity_decorator
    cuda.grid = lambda n: 0
    numba.cuda = cuda
    sys.modules["numba"] = numba
    sys.modules["numba.cuda"] = cuda
    sys.modules["open3d"] = types.ModuleType("open3d")
    sys.modules["h5py"] = types.ModuleType("h5py")


def _import_reference():
    assert os.path.isdir(REF), "reference tree not present; fixtures can only be made in the build container"
    repo_root = os.path.dirname(os.path.dirname(HERE))
    sys.path[:] = [p for p in sys.path if os.path.abspath(p or ".") != repo_root]
    sys.path.insert(0, REF)
    sys.path.insert(1, os.path.dirname(HERE))  # tests/ for fcn_weights
    _install_stubs()
    import AccumulatorSpace  # noqa: the reference module

    assert AccumulatorSpace.__file__.startswith(REF)
    return AccumulatorSpace


# --------------------------------------------------------------------------------------
# ACC-*: Accumulator_3D (AccumulatorSpace.py:373-419) + fast_for (:325-341)
# --------------------------------------------------------------------------------------
def _sphere_cloud(rng, n, centre_m, semi_axes_m):
    """Points on the camera-facing half of an ellipsoid (metres)."""
    u = rng.normal(size=(n, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    u[:, 2] = -np.abs(u[:, 2])
    return centre_m + u * semi_axes_m


def acc_cases():
    rng = np.random.default_rng(20240601)
    cases = {}
    centre = np.array([0.02, -0.03, 0.90])
    axes = np.array([0.020, 0.016, 0.012])
    kp = centre + np.array([0.030, 0.024, 0.018])

    # ACC-1: noiseless radii, float64 list
    xyz = _sphere_cloud(rng, 160, centre, axes)
    rad = np.linalg.norm(xyz - kp, axis=1) * 10.0  # decimetres
    cases["acc1_noiseless_f64"] = (xyz, rad.astype(np.float64))

    # ACC-2: noisy radii, float32 list (what the FCN path produces, AccumulatorSpace.py:610)
    xyz = _sphere_cloud(rng, 200, centre, axes)
    rad = np.linalg.norm(xyz - kp, axis=1) * 10.0 + rng.normal(0, 0.02, size=200)
    cases["acc2_noisy_f32"] = (xyz, rad.astype(np.float32))

    # ACC-3: radii that land exactly on x.5 voxels (np.around half-to-even, :332)
    xyz = _sphere_cloud(rng, 96, centre, axes)
    halves = (rng.integers(4, 14, size=96) + 0.5) * 5.0 / 100.0  # dm s.t. r*100/5 = k+0.5
    cases["acc3_half_even_f32"] = (xyz, halves.astype(np.float32))

    # ACC-4: two points -> many tied maxima: all rows of `center` are pinned
    xyz = np.array([[0.000, 0.000, 0.800], [0.020, 0.005, 0.810]])
    cases["acc4_two_point_ties"] = (xyz, np.array([0.25, 0.31], dtype=np.float64))

    # ACC-5: radii larger than the cloud extent -> shells clipped by the grid faces, zb < 0
    xyz = _sphere_cloud(rng, 120, centre, axes * 0.5)
    rad = rng.uniform(0.3, 0.9, size=120)
    cases["acc5_clipped_f32"] = (xyz, rad.astype(np.float32))

    # ACC-6: degenerate radii (zero, negative, tiny) mixed with valid ones
    xyz = _sphere_cloud(rng, 80, centre, axes)
    rad = np.linalg.norm(xyz - kp, axis=1) * 10.0
    rad[::7] = 0.0
    rad[3::11] = -0.12
    rad[5::13] = 0.01
    cases["acc6_degenerate_f32"] = (xyz, rad.astype(np.float32))

    # ACC-7: one point, radius below one voxel: zero_boundary = +1 (the un-shifted branch,
    # AccumulatorSpace.py:396), grid dimension 0 -> the reference raises (pinned error behaviour)
    cases["acc7_single_point_tiny_radius"] = (np.array([[0.01, 0.02, 0.70]]), np.array([0.04], dtype=np.float32))

    # ACC-8: one point, ordinary radius: smallest legal grid
    cases["acc8_single_point"] = (np.array([[0.01, 0.02, 0.70]]), np.array([0.26], dtype=np.float32))

    # ACC-9: more radii than points (SURVEY a-4: mask covers a depth hole): the surplus radii still
    # enter radius_max (AccumulatorSpace.py:392) and only the first N pair with points (:329-331)
    xyz = _sphere_cloud(rng, 64, centre, axes)
    rad = np.concatenate([np.linalg.norm(xyz - kp, axis=1) * 10.0, [0.95, 0.20, 0.31]])
    cases["acc9_surplus_radii_f32"] = (xyz, rad.astype(np.float32))
    return cases


def gen_acc(A):
    captured = {}
    orig_fast_for = A.fast_for

    def spy(xyz_mm, radial_list_mm, vote_map):
        captured["points_vox"] = np.array(xyz_mm, dtype=np.float64, copy=True)
        captured["radii_vox"] = np.array(radial_list_mm, copy=True)
        out = orig_fast_for(xyz_mm, radial_list_mm, vote_map)
        captured["votes"] = np.array(out, copy=True)
        return out

    A.fast_for = spy
    index = {}
    for name, (xyz, rad) in acc_cases().items():
        xyz_in = np.array(xyz, dtype=np.float64, copy=True)
        rad_in = np.array(rad, copy=True)
        t0 = time.time()
        try:
            center = A.Accumulator_3D(xyz_in, rad_in)
        except Exception as e:  # pinned error behaviour
            index[name] = {"raises": type(e).__name__, "xyz_m": xyz.tolist(), "radial_dm": rad.tolist(),
                           "radial_dtype": str(rad.dtype)}
            print("  %-24s raises %s" % (name, type(e).__name__))
            continue
        assert np.array_equal(xyz_in, xyz), "reference mutated xyz"
        votes = captured["votes"]
        assert np.array_equal(votes, np.round(votes)) and votes.max() < 65536
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"),
            xyz_m=xyz,
            radial_dm=rad,
            votes=votes.astype(np.uint16),
            center=center,
            points_vox=captured["points_vox"],
            radii_vox=captured["radii_vox"],
        )
        index[name] = {
            "n": int(len(xyz)),
            "D": int(votes.shape[0]),
            "peak": int(votes.max()),
            "ties": int(center.shape[0]),
            "radial_dtype": str(rad.dtype),
            "seconds": round(time.time() - t0, 1),
        }
        print("  %-24s %s" % (name, index[name]))
    A.fast_for = orig_fast_for

    # error behaviour of an empty cloud (AccumulatorSpace.py:390 -> numpy ValueError)
    try:
        A.Accumulator_3D(np.zeros((0, 3)), np.zeros((0,), dtype=np.float32))
        index["acc_empty"] = {"raises": None}
    except Exception as e:
        index["acc_empty"] = {"raises": type(e).__name__}
    with open(os.path.join(HERE, "acc_index.json"), "w") as f:
        json.dump(index, f, indent=1, sort_keys=True)


# --------------------------------------------------------------------------------------
# SEAM-*: AccumulatorSpace.py:602-619 (inline in estimate_6d_pose_lm) + :77-85
# --------------------------------------------------------------------------------------
def _seam_source(using_ckpts_branch_only=False):
    with open(os.path.join(REF, "AccumulatorSpace.py")) as f:
        lines = f.read().split("\n")
    block = lines[601:619]  # 1-based lines 602..619
    assert block[0].strip() == "if opts.using_ckpts:", block[0]
    assert block[-1].strip() == "xyz = xyz_mm/1000", block[-1]
    return textwrap.dedent("\n".join(block))


def _run_seam(A, sem, radial, depth, max_radius_dm, using_ckpts=True, radial_est=None):
    ns = {
        "np": np,
        "os": os,
        "opts": types.SimpleNamespace(using_ckpts=using_ckpts, root_dataset=""),
        "sem_out": sem,
        "radial_out": radial,
        "depth_map1": depth,
        "max_radii_dm": np.array([max_radius_dm, 0.0, 0.0]),
        "keypoint_count": 1,
        "rgbd_to_point_cloud": A.rgbd_to_point_cloud,
        "linemod_K": A.linemod_K,
        "class_name": "x",
        "filename": "x.jpg",
    }
    if not using_ckpts:
        # the else-branch loads the radial map from disk (np.load of a path built with os.path.join)
        class _NP:
            def __getattr__(self, k):
                return getattr(np, k)

            @staticmethod
            def load(path):
                return np.array(radial_est, copy=True)

        ns["np"] = _NP()
    exec(_seam_source(), ns)
    return ns["xyz"], ns["radial_list"], ns["xyz_mm"]


def gen_seam(A):
    rng = np.random.default_rng(777)
    H, W = 48, 64
    v, u = np.mgrid[0:H, 0:W]
    inside = ((u - 30) / 14.0) ** 2 + ((v - 22) / 10.0) ** 2 <= 1.0
    sem = np.where(inside, 0.95, 0.05).astype(np.float32)
    sem += rng.normal(0, 0.02, size=sem.shape).astype(np.float32)
    sem[20, 28] = 0.8  # exactly at threshold -> excluded (> 0.8)
    radial = (2.0 + 0.01 * u + 0.02 * v + rng.normal(0, 0.01, size=(H, W))).astype(np.float32)
    radial[18:21, 40:43] = 9.0  # above max radius -> removed from the mask
    depth = (900 + 2 * u + 3 * v).astype(np.uint16)
    index = {}

    # SEAM-1: no depth holes -> lists aligned
    xyz, rl, xyz_mm = _run_seam(A, sem.copy(), radial.copy(), depth.copy(), 3.5)
    np.savez_compressed(os.path.join(HERE, "seam1_aligned.npz"), sem=sem, radial=radial, depth=depth,
                        max_radius_dm=3.5, xyz_m=xyz, radial_list=rl, xyz_mm=xyz_mm)
    index["seam1_aligned"] = {"n_points": int(len(xyz)), "n_radii": int(len(rl))}

    # SEAM-2: depth holes inside the mask -> radial_list longer than xyz (SURVEY a-4 quirk)
    depth2 = depth.copy()
    depth2[22, 25:29] = 0
    depth2[15, 33] = 0
    xyz, rl, xyz_mm = _run_seam(A, sem.copy(), radial.copy(), depth2.copy(), 3.5)
    np.savez_compressed(os.path.join(HERE, "seam2_depth_holes.npz"), sem=sem, radial=radial, depth=depth2,
                        max_radius_dm=3.5, xyz_m=xyz, radial_list=rl, xyz_mm=xyz_mm)
    index["seam2_depth_holes"] = {"n_points": int(len(xyz)), "n_radii": int(len(rl))}
    assert len(rl) > len(xyz)

    # SEAM-3: the using_ckpts=False branch (:611-618): pre-computed radial map, aligned gather
    est = np.where(inside, radial, 0).astype(np.float32)
    xyz, rl, xyz_mm = _run_seam(A, None, None, depth2.copy(), 3.5, using_ckpts=False, radial_est=est)
    np.savez_compressed(os.path.join(HERE, "seam3_radial_npy.npz"), radial_est=est, depth=depth2,
                        max_radius_dm=3.5, xyz_m=xyz, radial_list=rl, xyz_mm=xyz_mm)
    index["seam3_radial_npy"] = {"n_points": int(len(xyz)), "n_radii": int(len(rl))}

    # SEAM-4: seam -> Accumulator_3D chained (the per-keypoint body, :596-654, small N)
    depth4 = np.where(inside, depth, 0).astype(np.uint16)
    pts_all = A.rgbd_to_point_cloud(A.linemod_K, depth.astype(np.int64)) / 1000.0
    pts_obj = A.rgbd_to_point_cloud(A.linemod_K, depth4.astype(np.int64)) / 1000.0
    kp = pts_obj.mean(axis=0) + np.array([0.030, 0.020, 0.025])  # keypoint just outside the object
    rad_true = np.zeros((H, W), dtype=np.float32)
    vs, us = depth.nonzero()
    rad_true[vs, us] = (np.linalg.norm(pts_all - kp, axis=1) * 10.0).astype(np.float32)
    sub = inside & ((u + v) % 3 == 0)  # thin the mask so the pure-Python vote stays small
    sem4 = np.where(sub, 1.0, -1.0).astype(np.float32)
    xyz, rl, xyz_mm = _run_seam(A, sem4.copy(), rad_true.copy(), depth4.copy(), float(rad_true[sub].max()))
    center = A.Accumulator_3D(xyz, rl)
    np.savez_compressed(os.path.join(HERE, "seam4_chain.npz"), sem=sem4, radial=rad_true, depth=depth4,
                        max_radius_dm=float(rad_true[sub].max()), xyz_m=xyz, radial_list=rl, center=center,
                        keypoint_m=kp)
    index["seam4_chain"] = {"n_points": int(len(xyz)), "center0": [float(c) for c in center[0]],
                            "keypoint_mm": [float(c) * 1000 for c in kp]}
    with open(os.path.join(HERE, "seam_index.json"), "w") as f:
        json.dump(index, f, indent=1, sort_keys=True)
    print("  seam:", index)


# --------------------------------------------------------------------------------------
# SEAM-LMO / SEAM-YCB: the inline seams of estimate_6d_pose_lmo (AccumulatorSpace.py:835-851, :858-864, its
# using_ckpts=False branch -- the using_ckpts=True branch cannot run: `radial_out<=max_radii_dm` at :838 compares an [H,W]
# map with the 3-vector) and of estimate_6d_pose_ycb (:1050-1057), exec'd from the reference file like the LINEMOD seam
# --------------------------------------------------------------------------------------
def _ref_lines(first, last, head, tail):
    with open(os.path.join(REF, "AccumulatorSpace.py")) as f:
        lines = f.read().split("\n")
    block = lines[first - 1:last]
    assert block[0].strip() == head, block[0]
    assert block[-1].strip() == tail, block[-1]
    return textwrap.dedent("\n".join(block))


def _run_seam_lmo(A, radial_est, depth, max_radii_dm, kp):
    class _NP:
        def __getattr__(self, k):
            return getattr(np, k)

        @staticmethod
        def load(path):
            return np.array(radial_est, copy=True)

    ns = {"np": _NP(), "os": os, "opts": types.SimpleNamespace(using_ckpts=False), "rootPath": "", "class_name": "x",
          "filename": "color_00007.png", "keypoint_count": kp, "max_radii_dm": np.asarray(max_radii_dm, dtype=np.float64),
          "depth_map": np.array(depth, dtype=np.float64), "rgbd_to_point_cloud": A.rgbd_to_point_cloud, "linemod_K": A.linemod_K}
    exec(_ref_lines(835, 851, "if opts.using_ckpts:", "depth_map = depth_map*sem_out"), ns)
    ns["np"] = np
    exec(_ref_lines(858, 864, "if radial_out.max()!=0:", "xyz = xyz_mm/1000"), ns)
    return ns.get("xyz"), ns.get("radial_list"), ns.get("xyz_mm")


def _run_seam_ycb(A, sem, radial, depth_u16, factor_depth, K):
    ns = {"np": np, "sem_out": sem, "radial_out": radial, "depth_map1": depth_u16,
          "sceneInfo": {"factor_depth": np.array([[factor_depth]], dtype=np.uint16), "intrinsic_matrix": np.asarray(K, dtype=np.float64)},
          "rgbd_to_point_cloud": A.rgbd_to_point_cloud}
    exec(_ref_lines(1050, 1057, "sem_out = np.where(sem_out>0.8,1,0)", "xyz = rgbd_to_point_cloud(sceneInfo['intrinsic_matrix'],depth_map)"), ns)
    return ns["xyz"], ns["radial_list"]


def gen_seam_variants(A):
    rng = np.random.default_rng(4711)
    H, W = 48, 64
    v, u = np.mgrid[0:H, 0:W]
    inside = ((u - 33) / 15.0) ** 2 + ((v - 25) / 11.0) ** 2 <= 1.0
    index = {}
    # LM-O, pre-computed radial maps: radii above the bound are zeroed, the mask is radial > 0, radii are gathered over the
    # mask and points over mask & depth != 0 (two depth holes -> misaligned lists); one exact-zero and one negative radius
    radial = np.where(inside, 2.0 + 0.01 * u + 0.02 * v + rng.normal(0, 0.01, size=(H, W)), 0).astype(np.float32)
    radial[20:22, 45:47] = 7.5     # above the bound
    radial[25, 30] = 0.0           # a hole in the radial map
    radial[26, 31] = -0.25         # not > 0
    depth = (850 + 3 * u + 2 * v).astype(np.uint16)
    depth[24, 28:31] = 0
    depth[30, 36] = 0
    max_radii = [3.4, 3.3, 2.9]
    for kp in (1, 3):
        xyz, rl, xyz_mm = _run_seam_lmo(A, radial, depth, max_radii, kp)
        np.savez_compressed(os.path.join(HERE, "seam_lmo%d.npz" % kp), radial_est=radial, depth=depth, max_radius_dm=max_radii[kp - 1],
                            xyz_m=xyz, radial_list=rl, xyz_mm=xyz_mm)
        index["seam_lmo%d" % kp] = {"n_points": int(len(xyz)), "n_radii": int(len(rl))}
        assert len(rl) > len(xyz)
    xyz, rl, _ = _run_seam_lmo(A, np.zeros((H, W), dtype=np.float32), depth, max_radii, 1)   # radial_out.max() == 0: skipped (:858)
    assert xyz is None and rl is None
    index["seam_lmo_empty"] = {"skipped": True}
    # YCB-Video: mask = seg > 0.8 only, depth / factor_depth, the frame's own intrinsic matrix, xyz in the depth map's unit
    K = np.array([[1066.778, 0.0, 312.9869], [0.0, 1067.487, 241.3109], [0.0, 0.0, 1.0]])
    sem = np.where(inside, 0.93, 0.1).astype(np.float32) + rng.normal(0, 0.02, size=(H, W)).astype(np.float32)
    sem[22, 30] = 0.8   # exactly at the threshold: excluded
    rad = (1.5 + 0.02 * u + 0.01 * v).astype(np.float32)
    d16 = (7000 + 31 * u + 17 * v).astype(np.uint16)
    d16[27, 33:36] = 0
    for name, factor in (("seam_ycb1", 10000), ("seam_ycb2", 7919)):
        xyz, rl = _run_seam_ycb(A, sem.copy(), rad.copy(), d16.copy(), factor, K)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), sem=sem, radial=rad, depth=d16, factor_depth=factor, K=K,
                            xyz=xyz, radial_list=rl)
        index[name] = {"n_points": int(len(xyz)), "n_radii": int(len(rl))}
        assert len(rl) > len(xyz)
    with open(os.path.join(HERE, "seam_variants_index.json"), "w") as f:
        json.dump(index, f, indent=1, sort_keys=True)
    print("  seam variants:", index)


# --------------------------------------------------------------------------------------
# FCN-*: models/fcnresnet.py:48-191, FCResBackbone (AccumulatorSpace.py:122-162)
# --------------------------------------------------------------------------------------
def gen_fcn(A):
    import torch
    from PIL import Image

    import fcn_weights
    from models.fcnresnet import DenseFCNResNet152

    torch.manual_seed(0)
    model = DenseFCNResNet152(3, 2)
    sd = fcn_weights.make_state_dict(model, seed=0)
    model.load_state_dict(sd, strict=True)
    model.eval()
    keys = {k: list(v.shape) for k, v in model.state_dict().items()}
    with open(os.path.join(HERE, "fcn_state_dict_keys.json"), "w") as f:
        json.dump(keys, f)
    print("  fcn: %d state-dict keys, %d params" % (len(keys), sum(p.numel() for p in model.parameters())))

    stages = {}

    def hook(name):
        def fn(mod, inp, out):
            stages[name] = out.detach().clone()
        return fn

    names = ["bn1", "maxpool", "block1", "block2", "block3", "block4", "bn6", "conv_up5", "up5", "conv_up4", "up4",
             "conv_up3", "up3", "conv_up2", "up2", "conv_up1", "up1", "conv7", "conv8"]
    handles = [getattr(model, n).register_forward_hook(hook(n)) for n in names]

    # FCN-1: reduced input, full output
    x = torch.from_numpy(fcn_weights.make_input(1, (1, 3, 64, 96)))
    with torch.no_grad():
        seg, rad = model(x)
    st = {n: stages[n].numpy() for n in names}
    np.savez_compressed(os.path.join(HERE, "fcn1_64x96.npz"), input_seed=1, seg=seg.numpy(), radial=rad.numpy(),
                        **{"stage_" + n: v for n, v in st.items() if v.size <= 200000})
    stats = {n: [float(v.mean()), float(np.abs(v).mean()), float(np.abs(v).max())] for n, v in st.items()}
    print("  fcn1 stage |mean| :", {n: round(s[1], 3) for n, s in stats.items()})

    # FCN-2: full 480x640 input, sampled output pixels + per-stage statistics
    x = torch.from_numpy(fcn_weights.make_input(2, (1, 3, 480, 640)))
    t0 = time.time()
    with torch.no_grad():
        seg, rad = model(x)
    dt = time.time() - t0
    g = np.random.default_rng(4096)
    pix = g.choice(480 * 640, size=4096, replace=False)
    stage_stats = {}
    stage_samples = {}
    for n in names:
        v = stages[n].numpy()
        stage_stats[n] = [float(v.mean()), float(np.abs(v).mean()), float(np.abs(v).max())]
        flat = v.reshape(-1)
        idx = g.choice(flat.size, size=64, replace=False)
        stage_samples["idx_" + n] = idx
        stage_samples["val_" + n] = flat[idx]
    np.savez_compressed(os.path.join(HERE, "fcn2_480x640_samples.npz"), input_seed=2, pix=pix,
                        seg=seg.numpy().reshape(-1)[pix], radial=rad.numpy().reshape(-1)[pix],
                        seg_absmax=float(seg.abs().max()), radial_absmax=float(rad.abs().max()), **stage_samples)
    with open(os.path.join(HERE, "fcn2_stage_stats.json"), "w") as f:
        json.dump({"seconds_reference_forward": round(dt, 2), "stats_mean_absmean_absmax": stage_stats}, f, indent=1)
    print("  fcn2 forward %.2fs; |radial|max %.3f" % (dt, float(rad.abs().max())))
    for h in handles:
        h.remove()

    # FCN-3: FCResBackbone through a temporary PNG (pins the uint8 -> float64 -> float32 preprocessing)
    rgb = np.random.default_rng(31337).integers(0, 256, size=(64, 96, 3), dtype=np.uint8)
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "frame.png")
        Image.fromarray(rgb).save(p)
        sem_out, radial_out = A.FCResBackbone(model, p, [])
    np.savez_compressed(os.path.join(HERE, "fcn3_backbone_64x96.npz"), rgb=rgb, sem=sem_out, radial=radial_out)
    print("  fcn3 backbone:", sem_out.shape, sem_out.dtype, radial_out.dtype)


# --------------------------------------------------------------------------------------
# RESFCN-1: models/fcnresnet.py:194-323 (ResFCNResNet152: additive skips `up + conv_upK_1(skip)`), imported by the
# reference but never instantiated by its drivers
# --------------------------------------------------------------------------------------
def gen_resfcn(A):
    import torch

    import fcn_weights
    from models.fcnresnet import ResFCNResNet152

    torch.manual_seed(0)
    model = ResFCNResNet152(3, 2)
    sd = fcn_weights.make_state_dict(model, seed=7)
    model.load_state_dict(sd, strict=True)
    model.eval()
    keys = {k: list(v.shape) for k, v in model.state_dict().items()}
    with open(os.path.join(HERE, "resfcn_state_dict_keys.json"), "w") as f:
        json.dump(keys, f)
    print("  resfcn: %d state-dict keys, %d params" % (len(keys), sum(p.numel() for p in model.parameters())))
    stages = {}

    def hook(name):
        def fn(mod, inp, out):
            stages[name] = out.detach().clone()
        return fn

    names = ["block1", "block2", "block3", "block4", "conv_up4", "conv_up3", "conv_up2", "conv_up1", "conv7", "conv8"]
    handles = [getattr(model, n).register_forward_hook(hook(n)) for n in names]
    x = torch.from_numpy(fcn_weights.make_input(3, (1, 3, 64, 96)))
    with torch.no_grad():
        seg, rad = model(x)
    np.savez_compressed(os.path.join(HERE, "resfcn1_64x96.npz"), input_seed=3, weight_seed=7, seg=seg.numpy(), radial=rad.numpy(),
                        **{"stage_" + n: stages[n].numpy() for n in names if stages[n].numel() <= 50000})
    print("  resfcn1 |radial|max %.3f |seg|max %.3f" % (float(rad.abs().max()), float(seg.abs().max())))
    for h in handles:
        h.remove()


# --------------------------------------------------------------------------------------
# HORN-*: util/horn.py:75-181 (HornPoseFitting.lmshorn), called at AccumulatorSpace.py:659-661
# --------------------------------------------------------------------------------------
def gen_horn(A):
    from util.horn import HornPoseFitting

    horn = HornPoseFitting()
    rng = np.random.default_rng(99)
    P1s, P2s, RTs, ns = [], [], [], []
    for case in range(12):
        n = 3 if case < 8 else int(rng.integers(4, 9))
        P1 = rng.normal(size=(n, 3)) * 120.0                       # model keypoints, mm
        ang = rng.normal(size=3)
        th = np.linalg.norm(ang)
        k = ang / th
        K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        R = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
        t = rng.normal(size=3) * 300.0 + np.array([0, 0, 900.0])
        P2 = P1 @ R.T + t + (rng.normal(size=(n, 3)) * (0.0 if case % 2 == 0 else 2.0))  # estimated keypoints (+ noise)
        if case == 7:                                             # degenerate: identical sets
            P2 = P1.copy()
        RT = np.zeros((4, 4))
        horn.lmshorn(P1.copy(), P2.copy(), n, RT)
        pad = np.zeros((8, 3))
        P1s.append(np.vstack([P1, pad])[:8]); P2s.append(np.vstack([P2, pad])[:8]); RTs.append(RT); ns.append(n)
    np.savez_compressed(os.path.join(HERE, "horn_cases.npz"), P1=np.array(P1s), P2=np.array(P2s), RT=np.array(RTs), n=np.array(ns))
    print("  horn: %d cases" % len(ns))


# --------------------------------------------------------------------------------------
# ND-*: the RGB-only variant (SURVEY f-5): rgbd_to_point_cloud_no_depth (AccumulatorSpace.py:100-120),
# Accumulator_3D_no_depth (:436-479) + fast_for_no_depth (:421-434).  Pinned exactly as the reference computes it
# (it indexes the [Z,N,3] hypothesis array like an [N,3] cloud: only the first three points are recentred, by the
# mean over all hypotheses and coordinates of point 0 / 1 / 2; the radii are NOT rounded; the grid has no radius
# margin; `center` is an int64 array, so adding the float mean truncates; every row is multiplied by 5).
# --------------------------------------------------------------------------------------
def nd_cases(A):
    rng = np.random.default_rng(4242)
    cases = {}

    def hypotheses(n_pix, z0, dz_mm, H=12, W=16):
        depth = np.zeros((H, W))
        pix = rng.choice(H * W, size=n_pix, replace=False)
        depth.reshape(-1)[np.sort(pix)] = z0 + rng.integers(0, dz_mm + 1, size=n_pix) * 0.001
        depth.reshape(-1)[np.sort(pix)[0]] = z0
        depth.reshape(-1)[np.sort(pix)[-1]] = z0 + dz_mm * 0.001
        K = np.array([[57.24114, 0., 8.2], [0., 57.357043, 6.1], [0., 0., 1.]])
        return K, depth, A.rgbd_to_point_cloud_no_depth(K, depth)

    # ND-1: 9 pixels x 4 depth hypotheses, float32 radii (what the FCN would emit), zero_boundary < 0
    K, depth, xyz = hypotheses(9, 0.150, 3)
    cases["nd1_f32"] = (K, depth, xyz, rng.uniform(0.04, 0.30, size=9).astype(np.float32))
    # ND-2: float64 radii, some of them integers / half-integers in voxel units (no rounding happens: pinned), one zero
    K, depth, xyz = hypotheses(7, 0.120, 2)
    rad = rng.uniform(0.05, 0.25, size=7)
    rad[0], rad[1], rad[2], rad[3] = 0.25, 0.175, 0.0, 0.2
    cases["nd2_f64"] = (K, depth, xyz, rad.astype(np.float64))
    # ND-3: the smallest legal input: three pixels, one hypothesis
    K, depth, xyz = hypotheses(3, 0.100, 0)
    cases["nd3_three_points"] = (K, depth, xyz, np.array([0.11, 0.13, 0.09], dtype=np.float32))
    # ND-4: every coordinate beyond the largest radius -> zero_boundary > 0, the un-shifted branch
    depth = np.zeros((12, 16))
    for (v, u, z) in ((9, 12, 0.210), (10, 13, 0.211), (10, 14, 0.212), (11, 15, 0.210), (9, 15, 0.212)):
        depth[v, u] = z
    K = np.array([[30.0, 0., 2.0], [0., 30.0, 1.5], [0., 0., 1.]])
    cases["nd4_unshifted"] = (K, depth, A.rgbd_to_point_cloud_no_depth(K, depth), np.array([0.05, 0.06, 0.04, 0.055, 0.045], dtype=np.float32))
    return cases


def gen_nodepth(A):
    captured = {}
    orig = A.fast_for_no_depth

    def spy(xyz_mm, radial_list_mm, vote_map):
        captured["points_vox"] = np.array(xyz_mm, dtype=np.float64, copy=True)
        captured["radii_vox"] = np.array(radial_list_mm, copy=True)
        out = orig(xyz_mm, radial_list_mm, vote_map)
        captured["votes"] = np.array(out, copy=True)
        return out

    A.fast_for_no_depth = spy
    index = {}
    for name, (K, depth, xyz, rad) in nd_cases(A).items():
        xyz_in, rad_in = np.array(xyz, copy=True), np.array(rad, copy=True)
        t0 = time.time()
        center = A.Accumulator_3D_no_depth(xyz_in, rad_in, None)
        assert np.array_equal(xyz_in, xyz) and np.array_equal(rad_in, rad), "reference mutated its inputs"
        votes = captured["votes"]
        assert np.array_equal(votes, np.round(votes)) and votes.max() < 65536
        np.savez_compressed(os.path.join(HERE, name + ".npz"), K=K, depth_m=depth, xyz_m=xyz, radial_dm=rad, votes=votes.astype(np.uint16),
                            center=center, points_vox=captured["points_vox"], radii_vox=captured["radii_vox"])
        index[name] = {"Z": int(xyz.shape[0]), "N": int(xyz.shape[1]), "D": int(votes.shape[0]), "peak": int(votes.max()),
                       "ties": int(center.shape[0]), "center_dtype": str(center.dtype), "radial_dtype": str(rad.dtype),
                       "seconds": round(time.time() - t0, 1)}
        print("  %-20s %s" % (name, index[name]))
    A.fast_for_no_depth = orig
    for label, xyz, rad in (("nd_two_points", np.zeros((2, 2, 3)) + 0.1, np.array([0.1, 0.1], dtype=np.float32)),
                            ("nd_empty", np.zeros((1, 0, 3)), np.zeros((0,), dtype=np.float32))):
        try:
            A.Accumulator_3D_no_depth(xyz, rad, None)
            index[label] = {"raises": None}
        except Exception as e:
            index[label] = {"raises": type(e).__name__}
    with open(os.path.join(HERE, "nd_index.json"), "w") as f:
        json.dump(index, f, indent=1, sort_keys=True)


def main():
    which = set(sys.argv[1:]) or {"acc", "seam", "seamv", "fcn", "horn", "resfcn", "nd"}
    A = _import_reference()
    if "acc" in which:
        print("ACC cases (reference fast_for runs as pure Python, ~1 us per point*voxel test)")
        gen_acc(A)
    if "seam" in which:
        gen_seam(A)
    if "seamv" in which:
        gen_seam_variants(A)
    if "fcn" in which:
        gen_fcn(A)
    if "horn" in which:
        gen_horn(A)
    if "resfcn" in which:
        gen_resfcn(A)
    if "nd" in which:
        print("ND cases (RGB-only accumulator; pure-Python fast_for_no_depth)")
        gen_nodepth(A)


if __name__ == "__main__":
    main()
