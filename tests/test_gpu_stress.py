"""BASELINE.json configs[4] (stress): 1 mm voxels, ~50k-point YCB-shaped cloud, float16 radial maps, 8 frames."""
import numpy as np
import pytest
import torch

import frames

pytestmark = pytest.mark.gpu


def _K(f):
    return np.array([[f["K"][0], 0, f["K"][2]], [0, f["K"][1], f["K"][3]], [0, 0, 1]])


def test_fp16_radial_seam_equals_oracle_on_rounded_maps(oracle):
    from rcvpose_amd import AccumulatorSpace as A
    from rcvpose_amd import _hip

    f = frames.synthetic_frame(1, shape="ycb")
    rad16 = f["radial"][0].astype(np.float16)
    rmax = float(rad16.astype(np.float32).max())
    seg, depth = torch.from_numpy(f["seg"]).cuda(), torch.from_numpy(f["depth"].view(np.int16)).cuda()
    ws = A.seam_compact(seg, torch.from_numpy(rad16).cuda(), depth, rmax, K=_K(f), mode=_hip.RCV_SEAM_RADIAL_F16)
    n, m = [int(v) for v in ws.counts.cpu()]
    xyz, rl = oracle.seam(f["seg"], rad16.astype(np.float32), f["depth"], rmax, K=f["K"])
    assert 47500 <= n <= 52500 and (n, m) == (len(xyz), len(rl))
    assert np.array_equal(ws.xyz[:n].cpu().numpy(), xyz) and np.array_equal(ws.radial[:m].cpu().numpy(), rl)


def test_configs4_eight_frames_fp16_radial_one_millimetre_against_oracle(oracle, golden_dir):
    """BASELINE.json configs[4] as stated: a batch of 8 YCB-shaped frames (~50k points each) x 3 keypoints in ONE batched
    call sequence (rcv_seam_compact_batch with RCV_SEAM_RADIAL_F16, rcv_accumulate_3d_batch with acc_unit = 1 mm), the
    radial maps written as float16 by the FCN engine's head (option 8) resp. injected as float16.  Checked against the
    oracle AT 1 mm (plane-wise peak: D ~ 900, the dense grid would need 3 GB): peak, tie count, first index, centre."""
    import json
    import os

    import fcn_weights
    from rcvpose_amd.pipeline import FramePipeline

    B, G, H, W = 8, 3, 480, 640
    fr = [frames.synthetic_frame(300 + i, shape="ycb") for i in range(B)]
    shapes = json.load(open(os.path.join(golden_dir, "fcn_state_dict_keys.json")))
    sds = [{k: torch.from_numpy(v) for k, v in fcn_weights.make_state_dict_numpy(shapes, seed=s).items()} for s in range(G)]
    pipe = FramePipeline(sds, H, W, frames_per_step=B, K=_K(fr[0]), acc_unit=1.0, radial_f16=True)  # (default arithmetic: f16x3)
    assert pipe.radial.dtype == torch.float16
    rgb = torch.from_numpy(np.stack([f["rgb"] for f in fr])).cuda()
    depth = torch.from_numpy(np.stack([f["depth"] for f in fr]).view(np.int16)).cuda()
    rad16 = np.stack([np.stack([f["radial"][k] for f in fr]) for k in range(G)]).astype(np.float16)      # [G,B,H,W]
    seg = torch.from_numpy(np.stack([np.stack([f["seg"] for f in fr])] * G))[:, :, None].contiguous().cuda()
    rmax = rad16.astype(np.float32).max(axis=(2, 3)).T.astype(np.float64)                                  # [B,G]
    pipe.process(rgb, depth, torch.from_numpy(rmax).cuda(), inject=(seg, torch.from_numpy(rad16)[:, :, None].contiguous().cuda()))
    kpts, res = pipe.read_results()
    for b in range(B):
        for k in range(G):
            r = res[b * G + k]
            assert r.status == 0 and r.dim > 800 and 47500 <= r.n_points <= 52500, (b, k, r.status, r.dim, r.n_points)
            assert np.abs(kpts[b, k] - fr[b]["keypoints_m"][k] * 1000).max() <= 5.0      # 1 mm voxels: within 5 voxels of the truth
    # both arg-max engines on all 24 items (r05): the LDS rasteriser alone (engine 'full', ~170 ms per item at 1 mm) must return the
    # search's (peak, ties, first index, centre) bit for bit
    del pipe
    torch.cuda.empty_cache()
    pipe_r = FramePipeline(sds, H, W, frames_per_step=B, K=_K(fr[0]), acc_unit=1.0, radial_f16=True, engine="full")
    pipe_r.process(rgb, depth, torch.from_numpy(rmax).cuda(), inject=(seg, torch.from_numpy(rad16)[:, :, None].contiguous().cuda()))
    kpts_r, res_r = pipe_r.read_results()
    assert np.array_equal(kpts_r, kpts)
    for a, c in zip(res, res_r):
        assert (a.status, a.dim, a.n_points, a.peak_votes, a.n_ties, tuple(a.first_idx[:])) == (c.status, c.dim, c.n_points, c.peak_votes, c.n_ties, tuple(c.first_idx[:]))
    del pipe_r
    torch.cuda.empty_cache()
    for b, k in ((0, 0), (7, 2), (3, 1), (5, 0)):  # (four of the 24 items against the 1 mm oracle: ~35 s each on the box's host cores)
        xyz, rl = oracle.seam(fr[b]["seg"], rad16[k, b].astype(np.float32), fr[b]["depth"], rmax[b, k], K=fr[b]["K"])
        r = res[b * G + k]
        assert (r.n_points, r.n_radii) == (len(xyz), len(rl))
        ref = oracle.peak_without_grid(xyz, rl, acc_unit=1.0)
        assert (r.dim, r.zero_boundary) == (ref["dim"], ref["zero_boundary"])
        assert (r.peak_votes, r.n_ties, tuple(r.first_idx[:])) == (ref["peak"], ref["n_ties"], ref["first_idx"]), (b, k)
        assert np.array_equal(kpts[b, k], ref["center_mm"]), (b, k)


def test_one_millimetre_rasteriser_against_oracle(oracle):
    """The LDS rasteriser (engine 'full') at 1 mm on one YCB-shaped item against the same 1 mm oracle, with the full
    workspace (replica area: a tile's point list split over several workgroups) and with the minimal one."""
    from rcvpose_amd import AccumulatorSpace as A
    from rcvpose_amd import _hip

    f = frames.synthetic_frame(2, shape="ycb")
    seg, depth = torch.from_numpy(f["seg"]).cuda(), torch.from_numpy(f["depth"].view(np.int16)).cuda()
    rmax = float(f["max_radii_dm"][1])
    xyz, rl = oracle.seam(f["seg"], f["radial"][1], f["depth"], rmax, K=f["K"])
    ref = oracle.peak_without_grid(xyz, rl, acc_unit=1.0)
    ws = A.AccWorkspace(480 * 640)
    small = int(_hip.lib().rcv_acc_workspace_bytes_min(480 * 640))
    assert small < ws.bytes // 3
    for nbytes in (ws.bytes, small):
        ws.bytes = nbytes
        A.seam_compact(seg, torch.from_numpy(f["radial"][1]).cuda(), depth, rmax, K=_K(f), workspace=ws)
        A.accumulate_on_device(ws.xyz, ws.radial, ws.counts, ws.max_points, ws.max_points, ws, acc_unit=1.0, engine="full")
        r = ws.read_result()
        assert r.status == 0 and (r.dim, r.peak_votes, r.n_ties, tuple(r.first_idx[:])) == (ref["dim"], ref["peak"], ref["n_ties"], ref["first_idx"])
        assert np.array_equal(np.array(r.center_mm[:]), ref["center_mm"])


def test_intermediate_grid_against_oracle(oracle):
    """2.5 mm voxels on a thinned cloud: both engines against the oracle's sparse grid (D ~ 420)."""
    from rcvpose_amd import AccumulatorSpace as A

    f = frames.synthetic_frame(3)
    xyz, rl = oracle.seam(f["seg"], f["radial"][2], f["depth"], f["max_radii_dm"][2])
    xyz, rl = xyz[::6], rl[::6]
    ref, votes, info, peak = oracle.accumulator_3d(xyz, rl, acc_unit=2.5, sparse=True, return_all=True)
    for mode in ("auto", "full"):
        c = A.Accumulator_3D(xyz, rl, acc_unit=2.5, engine=mode)
        assert np.array_equal(c, ref), mode


def test_engines_agree_on_many_structured_clouds():
    """Exact peak search (float32 wide levels + exact candidates) against the LDS rasteriser on 40 seeded clouds that
    look like the real input: points on a curved surface patch, radii = distance to a keypoint plus noise, different
    sizes, distances, noise levels and voxel sizes.  Peak, tie count and first index must agree exactly."""
    import torch

    from rcvpose_amd import AccumulatorSpace as A

    rng = np.random.default_rng(2024)
    for case in range(40):
        n = int(rng.integers(200, 20000))
        unit = float(rng.choice([5.0, 5.0, 2.5, 10.0]))
        centre = np.array([rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2), rng.uniform(0.6, 1.2)])
        u = rng.normal(size=(n, 3))
        u[:, 2] = -np.abs(u[:, 2])
        u /= np.linalg.norm(u, axis=1, keepdims=True)
        xyz = centre + u * np.array([0.12, 0.10, 0.08]) * rng.uniform(0.97, 1.0, size=(n, 1))
        kp = centre + rng.uniform(-0.25, 0.25, size=3)
        noise = float(rng.choice([0.0, 0.02, 0.1]))
        rad = (np.linalg.norm(xyz - kp, axis=1) * 10 + rng.normal(size=n) * noise).astype(np.float32)
        if case % 7 == 0:  # a few outliers with wrong radii
            rad[: n // 20] = rng.uniform(0.1, 3.0, size=n // 20)
        xyz_d, rad_d = torch.from_numpy(xyz).cuda(), torch.from_numpy(rad).cuda()
        ws = A.AccWorkspace(n)
        out = {}
        for mode in ("auto", "full"):
            A.accumulate_on_device(xyz_d, rad_d, None, n, n, ws, acc_unit=unit, engine=mode)
            out[mode] = ws.read_result()
        a, f = out["auto"], out["full"]
        assert a.status == 0 and f.status == 0, (case, a.status, f.status)
        assert (a.peak_votes, a.n_ties, tuple(a.first_idx[:]), a.dim) == (f.peak_votes, f.n_ties, tuple(f.first_idx[:]), f.dim), case
        assert tuple(a.center_mm[:]) == tuple(f.center_mm[:]), case
