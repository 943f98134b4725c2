"""estimate_6d_pose_lm on a synthetic LINEMOD-layout dataset written to a temp dir."""
import os
import types

import numpy as np
import pytest

import frames

pytestmark = pytest.mark.gpu


def _write_dataset(root, n_frames=2):
    cls = "ape"
    pv = os.path.join(root, "LINEMOD", cls)
    orig = os.path.join(root, "LINEMOD_ORIG", cls)
    for d in (os.path.join(pv, "Split"), os.path.join(pv, "pose"), os.path.join(pv, "JPEGImages"), os.path.join(orig, "data")):
        os.makedirs(d)
    from PIL import Image

    f0 = frames.synthetic_frame(0)
    a, b, c = 0.12, 0.10, 0.08
    rng = np.random.default_rng(0)
    u = rng.normal(size=(3000, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    model = u * np.array([a, b, c])                      # object frame, metres
    with open(os.path.join(pv, cls + ".ply"), "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\nend_header\n" % len(model))
        for p in model:
            f.write("%.9f %.9f %.9f\n" % tuple(p))
    sign = np.array([[1, 1, 1], [-1, 1, -1], [1, -1, -1]], dtype=np.float64)
    kps_obj = np.zeros((9, 3))
    kps_obj[1:4] = 2.0 * sign * np.array([a, b, c])      # rows 1..3 are the ones the driver uses (reference :541-548)
    np.save(os.path.join(pv, "Outside9.npy"), kps_obj)
    names, poses = [], {}
    for i in range(n_frames):
        fr = frames.synthetic_frame(i)
        stem = "%06d" % i
        names.append(stem)
        Image.fromarray(fr["rgb"]).save(os.path.join(pv, "JPEGImages", stem + ".jpg"))
        centre = fr["keypoints_m"][0] - kps_obj[1]       # object centre in the camera frame
        RT = np.concatenate([np.eye(3), centre[:, None]], axis=1)
        np.save(os.path.join(pv, "pose", "pose%d.npy" % i), RT)
        poses[stem] = RT
        with open(os.path.join(orig, "data", "depth%d.dpt" % i), "wb") as f:
            np.array(fr["depth"].shape, dtype=np.uint32).tofile(f)
            fr["depth"].astype(np.uint16).tofile(f)
        for k in range(3):
            d = os.path.join(root, "LINEMOD_ORIG", "estRadialMap", cls, "Out_pt%d_dm" % (k + 1))
            os.makedirs(d, exist_ok=True)
            np.save(os.path.join(d, stem + ".npy"), fr["radial"][k])
    with open(os.path.join(pv, "Split", "val.txt"), "w") as f:
        f.write("\n".join(names) + "\n")
    return cls, poses


def test_driver_with_precomputed_radial_maps(tmp_path, oracle):
    from rcvpose_amd import AccumulatorSpace as A

    root = str(tmp_path) + "/"
    cls, poses = _write_dataset(root)
    opts = types.SimpleNamespace(root_dataset=root, model_dir=root, using_ckpts=False, demo_mode=False, test_occ=False,
                                 class_names=[cls])
    results, details = A.estimate_6d_pose_lm(opts, return_details=True)
    assert results[cls]["frames"] == 2
    kps_obj = np.load(os.path.join(root, "LINEMOD", cls, "Outside9.npy"))
    model = A.read_ply_points(os.path.join(root, "LINEMOD", cls, cls + ".ply"))
    max_radii_dm = [(((model - kps_obj[i + 1]) ** 2).sum(axis=1) ** 0.5).max() * 10 for i in range(3)]
    for stem, d in details[cls].items():
        i = int(stem)
        fr = frames.synthetic_frame(i)
        # every keypoint equals what the oracle recovers from the same files (the reference's using_ckpts=False branch)
        for k in range(3):
            xyz, rl = oracle.seam_radial_only(fr["radial"][k], fr["depth"], max_radii_dm[k])
            ref = oracle.accumulator_3d(xyz, rl, sparse=True)
            assert np.array_equal(d["estimated_kpts"][k], ref[0]), (stem, k)
        assert np.array_equal(d["RT"], oracle.horn(kps_obj[1:4] * 1000, d["estimated_kpts"]))
        # and the pose is close to the ground truth (keypoints are within ~2 voxels of the truth)
        t_gt = poses[stem][:, 3] * 1000
        assert np.abs(d["RT"][:3, 3] - t_gt).max() < 25.0
        assert d["add_distance_mm"] < 15.0
        # ADD before ICP, the scene cloud, the ICP refinement and ADD after ICP against the CPU restatements
        from oracle import icp_oracle

        RTGT_mm = np.array(poses[stem], dtype=np.float64)
        RTGT_mm[:, 3] *= 1000
        gt_model = model * 1000 @ RTGT_mm[:, :3].T + RTGT_mm[:, 3]
        est_model = model * 1000 @ d["RT"][:3, :3].T + d["RT"][:3, 3]
        add0 = icp_oracle.nn_distance(gt_model, est_model)[0].mean()
        assert abs(d["add_distance_mm"] - add0) < 1e-9
        union = np.zeros(fr["depth"].shape, dtype=bool)
        for k in range(3):
            r = fr["radial"][k]
            union |= (r.astype(np.float64) <= max_radii_dm[k]) & (r != 0) & (fr["depth"] != 0)
        scene = oracle.backproject((fr["depth"] * union).astype(np.int64))
        assert d["scene_points"] == len(scene)
        ref = icp_oracle.registration_icp(model * 1000, scene, add0, d["RT"])
        assert d["icp_iterations"] == ref["iterations"]
        assert np.allclose(d["icp_transformation"], ref["transformation"], rtol=0, atol=1e-6)
        cad = model * 1000 @ ref["transformation"][:3, :3].T + ref["transformation"][:3, 3]
        assert abs(d["add_distance_after_icp_mm"] - icp_oracle.nn_distance(gt_model, cad)[0].mean()) < 1e-5
    assert results[cls]["add_before_icp"] == 1.0
    assert 0.0 <= results[cls]["add_after_icp"] <= 1.0


def _write_lm_checkpoints(root, cls, golden_dir, radii_dm):
    """Three checkpoints `<class>_pt{1,2,3}.pth.tar` (reference :522) with a zeroed last layer: every network answers seg = 1 and
    radial = its own constant -- known maps through the real FCN path (no trained weights offline)."""
    import json

    import torch

    import fcn_weights

    shapes = json.load(open(os.path.join(golden_dir, "fcn_state_dict_keys.json")))
    for k in range(3):
        sd = {n: torch.from_numpy(v) for n, v in fcn_weights.make_state_dict_numpy(shapes, seed=k).items()}
        sd["conv8.weight"] = torch.zeros_like(sd["conv8.weight"])
        sd["conv8.bias"] = torch.tensor([1.0, radii_dm[k]], dtype=torch.float32)
        torch.save(dict(epoch=1, model_state_dict=sd, optim_state_dict={}, loss=[]), os.path.join(root, cls + "_pt%d.pth.tar" % (k + 1)))


def test_lm_driver_with_checkpoints_batched(tmp_path, golden_dir, oracle):
    """The north-star branch of estimate_6d_pose_lm (using_ckpts=True, reference :520-530, :596-610): checkpoints -> FCN ->
    seam (seg > 0.8, radial <= max_radii_dm[k]) -> voting -> Horn -> ADD / ICP, with the frames travelling in batches of
    three (five frames: one full batch and a short one) -- against the oracle, and identical to the frame-by-frame run."""
    import time

    from rcvpose_amd import AccumulatorSpace as A

    root = str(tmp_path) + "/"
    cls, poses = _write_dataset(root, n_frames=5)
    radii = (3.0, 3.5, 4.0)
    _write_lm_checkpoints(root, cls, golden_dir, radii)
    base = dict(root_dataset=root, model_dir=root, using_ckpts=True, demo_mode=False, test_occ=False, class_names=[cls])
    t0 = time.perf_counter()
    res_b, det_b = A.estimate_6d_pose_lm(types.SimpleNamespace(frames_per_step=3, **base), return_details=True)
    t_b = time.perf_counter() - t0
    t0 = time.perf_counter()
    res_1, det_1 = A.estimate_6d_pose_lm(types.SimpleNamespace(frames_per_step=1, **base), return_details=True)
    t_1 = time.perf_counter() - t0
    print("estimate_6d_pose_lm, 5 frames with checkpoints: %.2f s in batches of 3, %.2f s frame by frame" % (t_b, t_1))
    assert res_b[cls]["frames"] == 5 and res_1[cls]["frames"] == 5
    kps_obj = np.load(os.path.join(root, "LINEMOD", cls, "Outside9.npy"))
    model = A.read_ply_points(os.path.join(root, "LINEMOD", cls, cls + ".ply"))
    max_radii_dm = [(((model - kps_obj[i + 1]) ** 2).sum(axis=1) ** 0.5).max() * 10 for i in range(3)]
    assert all(r <= m for r, m in zip(radii, max_radii_dm))
    for stem in sorted(det_b[cls]):
        db, d1 = det_b[cls][stem], det_1[cls][stem]
        for key in ("estimated_kpts", "RT", "icp_transformation"):
            assert np.array_equal(db[key], d1[key]), (stem, key)
        for key in ("add_distance_mm", "add_distance_after_icp_mm", "scene_points", "icp_iterations"):
            assert db[key] == d1[key], (stem, key)
        fr = frames.synthetic_frame(int(stem))
        ones = np.ones(fr["depth"].shape, dtype=np.float32)
        for k in range(3):
            xyz, rl = oracle.seam(ones, ones * np.float32(radii[k]), fr["depth"], max_radii_dm[k])
            assert len(rl) == ones.size and len(xyz) == fr["n_silhouette"]    # radii over the whole mask, points where depth != 0
            ref = oracle.accumulator_3d(xyz, rl, sparse=True)
            assert np.array_equal(db["estimated_kpts"][k], ref[0]), (stem, k)
        assert np.array_equal(db["RT"], oracle.horn(kps_obj[1:4] * 1000, db["estimated_kpts"]))
    assert res_b[cls]["add_before_icp"] == res_1[cls]["add_before_icp"] and res_b[cls]["net_time_ms"] > 0


def test_main_entry_runs_the_driver(tmp_path):
    from rcvpose_amd.main import main

    root = str(tmp_path) + "/"
    _write_dataset(root, n_frames=1)
    res = main(["--mode", "test", "--root_dataset", root, "--model_dir", root, "--using_ckpts", "false"])
    assert res["ape"]["frames"] == 1
    # the module's own command line (reference AccumulatorSpace.py:1202-1227)
    from rcvpose_amd import AccumulatorSpace as A

    res2 = A.main(["--dataset", "lm", "--root_dataset", root, "--model_dir", root])
    strip = lambda r: {c: {k: v for k, v in d.items() if not k.endswith("_s")} for c, d in r.items()}  # (wall-clock entries differ)
    assert strip(res2) == strip(res)
    # a root without a single class directory must not look like a successful evaluation (the reference dies with
    # FileNotFoundError on the first class's files): no YCB classes under this root, and the reference's default
    # --root_dataset without the trailing separator
    with pytest.raises(FileNotFoundError):
        A.main(["--dataset", "ycb", "--root_dataset", root, "--model_dir", root])
    with pytest.raises(FileNotFoundError):
        main(["--mode", "test", "--root_dataset", root.rstrip("/"), "--model_dir", root, "--using_ckpts", "false"])


def _write_lmo_dataset(root, n_frames=2):
    """Occlusion-LINEMOD layout (reference AccumulatorSpace.py:745-848) around the same synthetic frames; frame 1 has a depth
    hole under the mask (radius and point lists of different length), one extra image has no pose file."""
    cls = "ape"
    pv = os.path.join(root, "LINEMOD", cls)
    occ = os.path.join(root, "OCCLUSION_LINEMOD")
    for d in (pv, os.path.join(occ, "RGB-D", "rgb_noseg"), os.path.join(occ, "RGB-D", "depth_noseg"), os.path.join(occ, "blender_poses", cls)):
        os.makedirs(d)
    from PIL import Image

    a, b, c = 0.12, 0.10, 0.08
    rng = np.random.default_rng(0)
    u = rng.normal(size=(3000, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    model = u * np.array([a, b, c])
    with open(os.path.join(pv, cls + ".ply"), "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\nend_header\n" % len(model))
        for p in model:
            f.write("%.9f %.9f %.9f\n" % tuple(p))
    sign = np.array([[1, 1, 1], [-1, 1, -1], [1, -1, -1]], dtype=np.float64)
    kps_obj = np.zeros((9, 3))
    kps_obj[1:4] = 2.0 * sign * np.array([a, b, c])
    np.save(os.path.join(pv, "Outside9.npy"), kps_obj)
    depths, poses = {}, {}
    for i in range(n_frames):
        fr = frames.synthetic_frame(i)
        depth = fr["depth"].copy()
        if i == 1:
            ys, xs = np.nonzero(depth)
            depth[ys[:40], xs[:40]] = 0
        depths[i] = depth
        Image.fromarray(fr["rgb"]).save(os.path.join(occ, "RGB-D", "rgb_noseg", "color_%05d.png" % i))
        Image.fromarray(depth.astype(np.uint16)).save(os.path.join(occ, "RGB-D", "depth_noseg", "depth_%05d.png" % i))
        centre = fr["keypoints_m"][0] - kps_obj[1]
        RT = np.concatenate([np.eye(3), centre[:, None]], axis=1)
        np.save(os.path.join(occ, "blender_poses", cls, "pose%d.npy" % i), RT)
        poses[i] = RT
        for k in range(3):
            d = os.path.join(occ, "estRadialMap", cls, "Out_pt%d_dm" % (k + 1))
            os.makedirs(d, exist_ok=True)
            np.save(os.path.join(d, "_%05d.npy" % i), fr["radial"][k])
    Image.fromarray(frames.synthetic_frame(0)["rgb"]).save(os.path.join(occ, "RGB-D", "rgb_noseg", "color_%05d.png" % 7))
    return cls, poses, depths


def test_lmo_driver_with_precomputed_radial_maps(tmp_path, oracle):
    """estimate_6d_pose_lmo, the branch of the reference that runs (using_ckpts=False, :842-849)."""
    from rcvpose_amd import AccumulatorSpace as A
    from rcvpose_amd.train import Trainer

    root = str(tmp_path) + "/"
    cls, poses, depths = _write_lmo_dataset(root)
    opts = types.SimpleNamespace(root_dataset=root, model_dir=root, using_ckpts=False, demo_mode=False, test_occ=True, class_names=[cls])
    results, details = A.estimate_6d_pose_lmo(opts, return_details=True)
    assert results[cls]["frames"] == 3 and results[cls]["evaluated"] == 2  # the image without a pose file still counts (:963)
    kps_obj = np.load(os.path.join(root, "LINEMOD", cls, "Outside9.npy"))
    model = A.read_ply_points(os.path.join(root, "LINEMOD", cls, cls + ".ply"))
    max_radii_dm = [(((model - kps_obj[i + 1]) ** 2).sum(axis=1) ** 0.5).max() * 10 for i in range(3)]
    for i in range(2):
        d = details[cls]["color_%05d" % i]
        fr = frames.synthetic_frame(i)
        for k in range(3):
            r = fr["radial"][k]
            xyz, rl = oracle.seam((r > 0).astype(np.float32), r, depths[i], max_radii_dm[k], seg_thr=0.5)
            if i == 1 and k == 0:
                assert len(rl) == len(xyz) + 40  # the depth hole: lists of different length, paired like the reference
            ref = oracle.accumulator_3d(xyz, rl, sparse=True)
            assert np.array_equal(d["estimated_kpts"][k], ref[0]), (i, k)
        assert np.array_equal(d["RT"], oracle.horn(kps_obj[1:4] * 1000, d["estimated_kpts"]))
        assert d["icp_iterations"] == 1  # the LM-O criteria (:935) accept the first update
    # frame 0 passes ADD; frame 1 pairs radii and points with an offset of up to 40 (the depth hole), like the reference would
    assert details[cls]["color_00000"]["add_distance_mm"] < 15.0 and results[cls]["add_before_icp"] >= 1 / 3 - 1e-12
    assert Trainer([None, None], opts).Test()[cls]["frames"] == 3


def _write_ycb_dataset(root, golden_dir, radii_dm=(2.0, 2.5, 3.0)):
    """YCB-Video layout (reference AccumulatorSpace.py:986-1057) around the YCB-shaped synthetic frames.  The three
    checkpoints carry a zeroed last layer, so every network answers seg = 1 and radial = its own constant: known maps
    through the real FCN path.  Saved with `module.` keys, as `nn.DataParallel` checkpoints are (:1005)."""
    import json

    import scipy.io
    import torch
    from PIL import Image

    import fcn_weights

    cls, cls_id = "003_cracker_box", 2
    for d in (os.path.join(root, "Split", cls), os.path.join(root, "models", cls), os.path.join(root, "data", "0048")):
        os.makedirs(d)
    a, b, c = 0.0975, 0.079, 0.065
    rng = np.random.default_rng(0)
    u = rng.normal(size=(2000, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    model = u * np.array([a, b, c])
    np.savetxt(os.path.join(root, "models", cls, "points.xyz"), model, fmt="%.9f")
    model = np.loadtxt(os.path.join(root, "models", cls, "points.xyz"))     # the rounded text is what the driver sees
    sign = np.array([[1, 1, 1], [-1, 1, -1], [1, -1, -1]], dtype=np.float64)
    kps_obj = np.zeros((9, 3))
    kps_obj[1:4] = 2.0 * sign * np.array([a, b, c])
    np.save(os.path.join(root, "models", cls, "Outside9.npy"), kps_obj)
    fr = frames.synthetic_frame(0, shape="ycb")
    raw = (fr["depth"].astype(np.uint32) * 10).astype(np.uint16)       # factor_depth 10000: units of 0.1 mm
    Image.fromarray(fr["rgb"]).save(os.path.join(root, "data", "0048", "000001-color.png"))
    Image.fromarray(raw).save(os.path.join(root, "data", "0048", "000001-depth.png"))
    centre = fr["keypoints_m"][0] - kps_obj[1]
    RT = np.concatenate([np.eye(3), centre[:, None]], axis=1)
    poses = np.stack([np.concatenate([np.eye(3), np.ones((3, 1))], axis=1), RT], axis=2)   # another object first
    fx, fy, cx, cy = frames.YCB_K
    Kmat = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
    scipy.io.savemat(os.path.join(root, "data", "0048", "000001.mat"),
                     dict(poses=poses, cls_indexes=np.array([[5], [cls_id]], dtype=np.uint8), factor_depth=np.array([[10000]], dtype=np.uint16),
                          intrinsic_matrix=Kmat))
    with open(os.path.join(root, "Split", cls, "val.txt"), "w") as f:
        f.write("0048_000001\n")
    shapes = json.load(open(os.path.join(golden_dir, "fcn_state_dict_keys.json")))
    for k in range(3):
        sd = {n: torch.from_numpy(v) for n, v in fcn_weights.make_state_dict_numpy(shapes, seed=k).items()}
        sd["conv8.weight"] = torch.zeros_like(sd["conv8.weight"])
        sd["conv8.bias"] = torch.tensor([1.0, radii_dm[k]], dtype=torch.float32)
        torch.save(dict(epoch=1, model_state_dict={"module." + n: v for n, v in sd.items()}, loss=[]),
                   os.path.join(root, cls + "_pt%d.pth.tar" % (k + 1)))
    return cls, model, kps_obj, RT, raw, Kmat


def test_ycb_driver_end_to_end(tmp_path, golden_dir, oracle):
    """estimate_6d_pose_ycb: checkpoints -> FCN -> seam (seg > 0.8, depth / factor_depth, the frame's own K)
    -> voting -> Horn -> ADD / ICP / AUC, against the CPU restatements on the same files."""
    from oracle import icp_oracle
    from rcvpose_amd import AccumulatorSpace as A

    root = str(tmp_path) + "/"
    radii = (2.0, 2.5, 3.0)
    cls, model, kps_obj, RT_gt, raw, Kmat = _write_ycb_dataset(root, golden_dir, radii)
    opts = types.SimpleNamespace(root_dataset=root, model_dir=root, using_ckpts=True, demo_mode=False, class_names=[cls])
    results, details = A.estimate_6d_pose_ycb(opts, return_details=True)
    assert results[cls]["frames"] == 1
    d = details[cls]["0048_000001"]
    # the reference's seam for this loop (:1051-1059): every pixel is in the mask, depth in metres = raw / factor_depth
    depth_m = raw.astype(np.float64) / 10000.0
    vs, us = depth_m.nonzero()
    zs = depth_m[vs, us]
    xyz = np.stack([(us - Kmat[0, 2]) * zs / Kmat[0, 0], (vs - Kmat[1, 2]) * zs / Kmat[1, 1], zs], axis=1)
    assert len(xyz) > 45000
    for k in range(3):
        ref = oracle.accumulator_3d(xyz, np.full(raw.size, radii[k], dtype=np.float32), sparse=True)
        assert np.array_equal(d["estimated_kpts"][k], ref[0]), k    # seam mode 3 divides by factor_depth first, like :1052: identical
    # constant-radius maps put the three peaks nearly on one line: the rotation about it is ill-conditioned, so the pose is
    # checked through what Horn minimises rather than element by element
    kp_mm = kps_obj[1:4] * 1000
    fit = lambda T: np.linalg.norm(kp_mm @ T[:3, :3].T + T[:3, 3] - d["estimated_kpts"])
    assert fit(d["RT"]) <= fit(oracle.horn(kp_mm, d["estimated_kpts"])) + 1e-6
    assert abs(np.linalg.det(d["RT"][:3, :3]) - 1.0) < 1e-9
    gt_model = model * 1000 @ RT_gt[:, :3].T + RT_gt[:, 3] * 1000
    est_model = model * 1000 @ d["RT"][:3, :3].T + d["RT"][:3, 3]
    add0 = icp_oracle.nn_distance(gt_model, est_model)[0].mean()
    assert abs(d["add_distance_mm"] - add0) < 1e-9
    assert d["scene_points"] == len(xyz)
    ref = icp_oracle.registration_icp(model * 1000, xyz * 1000, add0, d["RT"], max_iteration=2000000)
    assert d["icp_iterations"] == ref["iterations"]
    assert np.allclose(d["icp_transformation"], ref["transformation"], rtol=0, atol=1e-5)
    cad = model * 1000 @ ref["transformation"][:3, :3].T + ref["transformation"][:3, 3]
    add1 = icp_oracle.nn_distance(gt_model, cad)[0].mean()
    assert abs(d["add_distance_after_icp_mm"] - add1) < 1e-4
    # summary figures: 1 % of the bounding-box diagonal, AUC over 0 ... 0.1 m
    diag = results[cls]["bbox_diagonal_m"]
    assert abs(diag - 2 * np.linalg.norm([0.0975, 0.079, 0.065])) < 0.01
    thr = np.array([0, 0.02, 0.04, 0.06, 0.08, 0.1]) * 1000
    for key, val in (("auc_before_icp", add0), ("auc_after_icp", add1)):
        y = (val <= thr).astype(np.float64)
        assert abs(results[cls][key] - np.trapezoid(y, thr / 1000) / 0.1) < 1e-12
    assert results[cls]["add_before_icp"] == float(add0 <= diag * 10)
