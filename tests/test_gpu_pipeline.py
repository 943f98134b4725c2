"""Frame pipeline (FCN -> seam -> vote -> arg-max, 3 keypoints) on the GPU against the oracle."""
import json
import os

import numpy as np
import pytest
import torch

import fcn_weights
import frames

pytestmark = pytest.mark.gpu


def _state_dicts(golden_dir, n=3):
    shapes = json.load(open(os.path.join(golden_dir, "fcn_state_dict_keys.json")))
    return [{k: torch.from_numpy(v) for k, v in fcn_weights.make_state_dict_numpy(shapes, seed=s).items()} for s in range(n)]


def test_full_frames_three_keypoints(golden_dir, oracle):
    """2 synthetic 640x480 frames x 3 keypoints, ideal maps injected after the forward:
    recovered keypoints equal the oracle's bit for bit (the north star's gate is <= 1 voxel from the reference)
    and land near the true keypoints (noisy integer-radius shells seen from ~0.4 m: a few voxels)."""
    from rcvpose_amd.pipeline import FramePipeline

    B = 2
    fr = [frames.synthetic_frame(i) for i in range(B)]
    pipe = FramePipeline(_state_dicts(golden_dir), 480, 640, frames_per_step=B)  # (default arithmetic: f16x3)
    rgb = torch.from_numpy(np.stack([f["rgb"] for f in fr])).cuda()
    depth = torch.from_numpy(np.stack([f["depth"] for f in fr]).view(np.int16)).cuda()
    seg = torch.from_numpy(np.stack([np.stack([f["seg"] for f in fr])] * 3))[:, :, None].contiguous().cuda()
    rad = torch.from_numpy(np.stack([np.stack([f["radial"][k] for f in fr]) for k in range(3)]))[:, :, None].contiguous().cuda()
    # all frames share max_radii (the reference derives them per class from the model, AccumulatorSpace.py:540-549)
    rmax = np.max([f["max_radii_dm"] for f in fr], axis=0)
    pipe.process(rgb, depth, rmax, inject=(seg, rad))
    kpts, res = pipe.read_results()
    for b in range(B):
        for k in range(3):
            r = res[b * 3 + k]
            assert r.status == 0
            xyz, rl = oracle.seam(fr[b]["seg"], fr[b]["radial"][k], fr[b]["depth"], rmax[k])
            assert r.n_points == len(xyz) and r.n_radii == len(rl)
            ref = oracle.accumulator_3d(xyz, rl, sparse=True)
            assert np.array_equal(kpts[b, k], ref[0]), (b, k, kpts[b, k], ref[0])
            assert np.abs(kpts[b, k] - fr[b]["keypoints_m"][k] * 1000).max() <= 15.0
    assert torch.equal(pipe.keypoints_tensor().cpu(), torch.from_numpy(kpts))


def test_small_frame_fcn_outputs_feed_the_vote(golden_dir, oracle):
    """No injection: whatever the (random-weight) fp32 FCN emits goes through seam + vote on the device, and the
    same maps pushed through the oracle give the same keypoint (or the same error)."""
    from rcvpose_amd.pipeline import FramePipeline

    H, W = 64, 96
    sds = _state_dicts(golden_dir, 1)
    pipe = FramePipeline(sds, H, W, dtype="fp32", frames_per_step=1)
    rng = np.random.default_rng(3)
    rgb = torch.from_numpy(rng.integers(0, 256, size=(1, H, W, 3), dtype=np.uint8)).cuda()
    depth_np = (800 + rng.integers(0, 40, size=(1, H, W))).astype(np.uint16)
    depth = torch.from_numpy(depth_np.view(np.int16)).cuda()
    pipe.forward_fcn(rgb)
    # shift the random seg map so that a few hundred pixels pass the 0.8 threshold, keep radii small and positive
    seg = pipe.seg[0, 0, 0]
    thr = torch.quantile(seg.flatten(), 0.95)
    pipe.seg.copy_((seg - thr + 0.8)[None, None, None])
    pipe.radial.copy_((pipe.radial[0, 0, 0].abs() % 0.9)[None, None, None])
    pipe.vote(depth, [0.85])
    kpts, res = pipe.read_results()
    xyz, rl = oracle.seam(pipe.seg[0, 0, 0].cpu().numpy(), pipe.radial[0, 0, 0].cpu().numpy(), depth_np[0], 0.85)
    assert res[0].n_points == len(xyz) and res[0].n_radii == len(rl) and len(xyz) > 50
    ref = oracle.accumulator_3d(xyz, rl, sparse=True)
    assert res[0].status == 0 and np.array_equal(kpts[0, 0], ref[0])


def _keypoints_under_engine_error(golden_dir, dtype, B=8):
    """The low-precision engine's error transferred onto ideal maps (see test below) -> (deviation in voxels [B,G],
    relative map errors per network, perturbed maps, frames, rmax, the low-precision pipeline's keypoints)."""
    from rcvpose_amd.pipeline import FramePipeline

    G, H, W = 3, 480, 640
    fr = [frames.synthetic_frame(200 + i) for i in range(B)]
    sds = _state_dicts(golden_dir)
    pb = FramePipeline(sds, H, W, dtype=dtype, frames_per_step=B)
    pf = FramePipeline(sds, H, W, dtype="fp32", frames_per_step=B)
    rgb = torch.from_numpy(np.stack([f["rgb"] for f in fr])).cuda()
    depth = torch.from_numpy(np.stack([f["depth"] for f in fr]).view(np.int16)).cuda()
    ideal_seg = torch.from_numpy(np.stack([np.stack([f["seg"] for f in fr])] * G))[:, :, None].contiguous().cuda()
    ideal_rad = torch.from_numpy(np.stack([np.stack([f["radial"][k] for f in fr]) for k in range(G)]))[:, :, None].contiguous().cuda()
    pb.forward_fcn(rgb)
    pf.forward_fcn(rgb)
    d_rad, d_seg = pb.radial - pf.radial, pb.seg - pf.seg
    rel = [(float(d_rad[k].abs().max() / pf.radial[k].abs().max()), float(d_seg[k].abs().max() / pf.seg[k].abs().max())) for k in range(G)]
    scale_r = (ideal_rad.amax(dim=(1, 2, 3, 4)) / pf.radial.abs().amax(dim=(1, 2, 3, 4))).view(G, 1, 1, 1, 1)
    scale_s = (1.0 / pf.seg.abs().amax(dim=(1, 2, 3, 4))).view(G, 1, 1, 1, 1)
    pert_rad, pert_seg = ideal_rad + scale_r * d_rad, ideal_seg + scale_s * d_seg
    step = float((scale_r * d_rad).abs().max()) * 20, float((scale_r * d_rad).pow(2).mean().sqrt()) * 20   # voxels (5 mm)
    pb.seg.copy_(pert_seg)
    pb.radial.copy_(pert_rad)
    pf.seg.copy_(ideal_seg)
    pf.radial.copy_(ideal_rad)
    rmax = np.max([f["max_radii_dm"] for f in fr], axis=0) + 0.5   # the perturbation must not push radii over the bound
    pb.vote(depth, rmax)
    pf.vote(depth, rmax)
    kb, rb = pb.read_results()
    kf, rf = pf.read_results()
    assert all(r.status == 0 for r in rb + rf)
    return np.abs(kb - kf).max(axis=2) / 5.0, rel, step, (pert_seg.cpu().numpy(), pert_rad.cpu().numpy()), fr, rmax, kb


def test_bf16_keypoints_against_fp32_keypoints(golden_dir, oracle):
    """What the bf16 engine's error does to the KEYPOINTS (VERDICT r01 #1b): 8 frames x 3 networks at 480x640 through the
    fp32 and the bf16 engine.  With random weights (no checkpoints offline) the maps carry no object, so the error field
    of the real bf16 forward, delta = map_bf16 - map_fp32, is transferred onto the ideal maps of the synthetic frames,
    scaled by range(ideal) / range(fp32 map) (an affine rescale of the network output to the radial range of a trained
    network): the fp32 pipeline votes on `ideal`, the bf16 pipeline on `ideal + scale * delta`.

    Measured: the bf16 error field (rms ~0.6 voxel of radius, spatially coherent where it stems from
    the coarse decoder levels) moves keypoints by 1-2 voxels on average and by up to ~7: bf16 does NOT keep the keypoints
    within north_star's 1 voxel under this model -- it is the throughput mode; fp32 (the drivers' default) is the parity
    mode.  The bound asserted here is a regression guard on that measurement, not a parity claim.  The voting itself is
    exact on whatever maps it is given (checked against the oracle on the perturbed maps)."""
    dev, rel, step, (ps, pr), fr, rmax, kb = _keypoints_under_engine_error(golden_dir, "bf16")
    for k, (er, es) in enumerate(rel):
        print("network %d: bf16 - fp32 maps, max |delta| / max |map|: radial %.3e seg %.3e" % (k, er, es))
    print("radial perturbation: max %.2f voxels, rms %.2f voxels" % step)
    print("bf16 vs fp32 keypoints: max %.2f voxels, mean %.3f voxels over %d keypoints" % (dev.max(), dev.mean(), dev.size))
    # regression guards at ~1.3x the measured figures (max 7 voxels, mean 1.7; map error 1.5-5.5e-2; radius perturbation
    # 0.62 voxel rms): a numeric regression of the 16-bit kernels fails here
    assert dev.max() <= 9.5 and dev.mean() <= 2.3
    assert max(er for er, _ in rel) <= 7.5e-2 and step[1] <= 0.85
    for b, k in ((0, 0), (3, 1), (7, 2)):             # the device voted exactly like the oracle on the perturbed maps
        xyz, rl = oracle.seam(ps[k, b, 0], pr[k, b, 0], fr[b]["depth"], rmax[k])
        assert np.array_equal(kb[b, k], oracle.accumulator_3d(xyz, rl, sparse=True)[0])


def test_fp16_keypoints_against_fp32_keypoints(golden_dir):
    """The same measurement for the fp16 storage mode: map error ~7x below bf16's, keypoints closer to the fp32 ones."""
    dev, rel, step, _, _, _, _ = _keypoints_under_engine_error(golden_dir, "fp16")
    for k, (er, es) in enumerate(rel):
        print("network %d: fp16 - fp32 maps, max |delta| / max |map|: radial %.3e seg %.3e" % (k, er, es))
        assert er <= 1.2e-2 and es <= 1.2e-2
    print("radial perturbation: max %.2f voxels, rms %.2f voxels" % step)
    print("fp16 vs fp32 keypoints: max %.2f voxels, mean %.3f voxels over %d keypoints" % (dev.max(), dev.mean(), dev.size))
    assert dev.max() <= 3.0 and dev.mean() <= 0.7   # ~1.3x the measured 2 / 0.5 voxels


def test_f16x3_keypoints_against_fp32_keypoints(golden_dir):
    """The same measurement for the split-precision engine (the drivers' and bench.py's default): its map error is four
    orders of magnitude below one voxel of radius, and the keypoints stay within north_star's 1 voxel of the fp32 engine's."""
    dev, rel, step, _, _, _, _ = _keypoints_under_engine_error(golden_dir, "f16x3")
    for k, (er, es) in enumerate(rel):
        print("network %d: f16x3 - fp32 maps, max |delta| / max |map|: radial %.3e seg %.3e" % (k, er, es))
        assert er <= 1e-4 and es <= 1e-4
    print("radial perturbation: max %.5f voxels, rms %.6f voxels" % step)
    print("f16x3 vs fp32 keypoints: max %.2f voxels, mean %.3f voxels over %d keypoints, %d identical" %
          (dev.max(), dev.mean(), dev.size, int((dev == 0).sum())))
    assert dev.max() <= 1.0


def test_keypoints_under_false_positives_and_noise_equal_the_oracle(golden_dir, oracle):
    """The peak search's second start (robust seed, line search along the sheet normal, restart of the lower bound from the
    candidate block with the largest upper bound -- DESIGN 4.2, r03) is only a place to start counting: on frames with
    false-positive segmentation on a background plane, gross radius outliers and one voxel of radial noise the keypoints are
    still the oracle's, bit for bit, and the search (not the rasteriser) decides them."""
    from rcvpose_amd.pipeline import FramePipeline

    cases = [dict(noise_dm=0.02, outlier_frac=0.0, seg_false_pos=0.1), dict(noise_dm=0.05, outlier_frac=0.2, seg_false_pos=0.1)]
    fr = [frames.synthetic_frame(300 + i, **cases[i]) for i in range(2)]
    B, G, H, W = 2, 3, 480, 640
    pipe = FramePipeline(_state_dicts(golden_dir), H, W, frames_per_step=B)   # (only the voting half is used)
    pipe.seg.copy_(torch.from_numpy(np.stack([np.stack([f["seg"] for f in fr])] * G))[:, :, None].cuda())
    pipe.radial.copy_(torch.from_numpy(np.stack([np.stack([f["radial"][k] for f in fr]) for k in range(G)]))[:, :, None].cuda())
    depth = torch.from_numpy(np.stack([f["depth"] for f in fr]).view(np.int16)).cuda()
    rmax = torch.tensor(np.stack([f["max_radii_dm"] for f in fr]), dtype=torch.float64, device="cuda")
    pipe.vote(depth, rmax)
    kp, res = pipe.read_results()
    hdr = pipe.ws[:, :512].cpu().numpy().view(np.int32)
    for b in range(B):
        assert fr[b]["depth"].astype(bool).sum() > fr[b]["n_silhouette"]  # the false positives are in the cloud
        for k in range(G):
            r = res[b * G + k]
            assert r.status == 0 and int(hdr[b * G + k, 116 // 4]) == 1   # AccHeader::bb_done: decided by the search
            xyz, rl = oracle.seam(fr[b]["seg"], fr[b]["radial"][k], fr[b]["depth"], fr[b]["max_radii_dm"][k])
            ref, _, info, _ = oracle.accumulator_3d(xyz, rl, sparse=True, return_all=True)
            assert np.array_equal(kp[b, k], ref[0]), (b, k)
            assert r.dim == info.dim and r.n_points == len(xyz)


def test_flat_vote_landscape_is_handed_to_the_rasteriser_and_equals_the_oracle(golden_dir, oracle):
    """Four voxels of radial noise flatten the vote landscape until certifying its arg-max by branch and bound costs more than
    voting the whole grid.  The search hands such an item to the LDS rasteriser: as soon as its lower bound says so (k_bb_gate(2)
    behind the restart on the 8-voxel candidates, r05: best exact count / N times r_max below BB_FLAT_F_RMAX = 7.8 --
    AccHeader::bb_handover 2) or, failing that prediction,
    once more than BB_HANDOVER_R2 = 2.5 r_max^2 four-voxel candidates survive (d_bb_restart, r04 -- bb_handover 1).  Same keypoints
    as the oracle either way; a benign frame in the same batch stays with the search."""
    from rcvpose_amd.pipeline import FramePipeline

    fr = [frames.synthetic_frame(310, noise_dm=0.2), frames.synthetic_frame(311)]
    B, G, H, W = 2, 3, 480, 640
    pipe = FramePipeline(_state_dicts(golden_dir), H, W, frames_per_step=B)   # (only the voting half is used)
    pipe.seg.copy_(torch.from_numpy(np.stack([np.stack([f["seg"] for f in fr])] * G))[:, :, None].cuda())
    pipe.radial.copy_(torch.from_numpy(np.stack([np.stack([f["radial"][k] for f in fr]) for k in range(G)]))[:, :, None].cuda())
    depth = torch.from_numpy(np.stack([f["depth"] for f in fr]).view(np.int16)).cuda()
    rmax = torch.tensor(np.stack([f["max_radii_dm"] for f in fr]), dtype=torch.float64, device="cuda")
    pipe.vote(depth, rmax)
    kp, res = pipe.read_results()
    hdr = pipe.ws[:, :512].cpu().numpy().view(np.int32)
    done = [[int(hdr[b * G + k, 116 // 4]) for k in range(G)] for b in range(B)]   # AccHeader::bb_done
    hand = [[int(hdr[b * G + k, 324 // 4]) for k in range(G)] for b in range(B)]   # AccHeader::bb_handover
    print("decided by the search (1) / by the rasteriser (0):", done, "hand-over kind:", hand)
    assert sum(done[0]) == 0 and all(h in (1, 2) for h in hand[0]) and 2 in hand[0]   # (peak / N ~ 0.05, r_max ~ 94: f r_max ~ 4.5)
    assert sum(done[1]) == G and not any(hand[1])
    for b in range(B):
        for k in range(G):
            r = res[b * G + k]
            xyz, rl = oracle.seam(fr[b]["seg"], fr[b]["radial"][k], fr[b]["depth"], fr[b]["max_radii_dm"][k])
            ref, _, info, peak = oracle.accumulator_3d(xyz, rl, sparse=True, return_all=True)
            assert r.status == 0 and np.array_equal(kp[b, k], ref[0]), (b, k)
            assert (r.dim, r.peak_votes) == (info.dim, peak)


def test_pipeline_reports_a_forward_that_left_the_half_range(golden_dir):
    """The batched pipeline stays asynchronous, so an activation beyond the half range (a checkpoint whose stem weights are
    3e5 times too large here) surfaces where the results are read: FramePipeline.read_results raises (RCV_E_RANGE) instead of
    handing out keypoints voted on clipped maps; the same frames through the fp32 engine pass."""
    from rcvpose_amd import _hip
    from rcvpose_amd.pipeline import FramePipeline

    H, W = 64, 96
    sds = _state_dicts(golden_dir, 1)
    bad = dict(sds[0])
    bad["conv1.weight"] = bad["conv1.weight"] * 3.0e5
    rng = np.random.default_rng(5)
    rgb = torch.from_numpy(rng.integers(0, 256, size=(1, H, W, 3), dtype=np.uint8)).cuda()
    depth = torch.from_numpy((800 + rng.integers(0, 40, size=(1, H, W))).astype(np.uint16).view(np.int16)).cuda()
    pipe = FramePipeline([bad], H, W, frames_per_step=1)           # default arithmetic: f16x3
    pipe.process(rgb, depth, [50.0])
    with pytest.raises(_hip.HalfRangeError):
        pipe.read_results()
    ok = FramePipeline([bad], H, W, dtype="fp32", frames_per_step=1)
    ok.process(rgb, depth, [50.0])
    ok.read_results()                                               # fp32 has the reference's range: no flag


def test_read_results_refuses_non_finite_maps(golden_dir):
    """The range guard sees magnitudes, not NaNs (v_max and the next ReLU drop them): FramePipeline.read_results checks the maps the
    votes were cast on.  A NaN written into the radial plane after the forward surfaces as FloatingPointError."""
    from rcvpose_amd.pipeline import FramePipeline

    H, W = 64, 96
    pipe = FramePipeline(_state_dicts(golden_dir, 1), H, W, frames_per_step=1)
    rng = np.random.default_rng(2)
    rgb = torch.from_numpy(rng.integers(0, 256, size=(1, H, W, 3), dtype=np.uint8)).cuda()
    depth = torch.from_numpy((800 + rng.integers(0, 40, size=(1, H, W))).astype(np.uint16).view(np.int16)).cuda()
    seg = torch.full((1, 1, 1, H, W), 1.0, device="cuda")
    rad = torch.full((1, 1, 1, H, W), 0.5, device="cuda")
    pipe.process(rgb, depth, [1.0], inject=(seg, rad))
    pipe.read_results()                       # finite maps: fine
    rad[0, 0, 0, 3, 5] = float("nan")
    pipe.process(rgb, depth, [1.0], inject=(seg, rad))
    with pytest.raises(FloatingPointError):
        pipe.read_results()
    pipe.process(rgb, depth, [1.0], inject=(seg, torch.full_like(rad, 0.5)))
    pipe.read_results()                       # the flag was consumed with the batch it belonged to
    # double-buffered stream of batches: the NaN batch is followed by two clean ones, which take both output buffers before the
    # results are read -- the seam counted the non-finite pixels of the maps ITS votes were cast on, so the batch is still refused
    over = FramePipeline(_state_dicts(golden_dir, 1), H, W, frames_per_step=1, overlap=True)
    clean = torch.full_like(rad, 0.5)
    for r in (rad, clean, clean):
        over.enqueue_step(rgb, depth, [1.0], inject=(seg, r))
    with pytest.raises(FloatingPointError):
        over.read_results()
    for r in (clean, clean, clean):
        over.enqueue_step(rgb, depth, [1.0], inject=(seg, r))
    over.read_results()
    # a NaN in the segmentation plane is dropped by `seg > 0.8` without a trace: counted as well
    seg_bad = seg.clone()
    seg_bad[0, 0, 0, 7, 9] = float("inf")
    pipe.process(rgb, depth, [1.0], inject=(seg_bad, clean))
    with pytest.raises(FloatingPointError):
        pipe.read_results()


def test_mfma_calibration_returns_a_plausible_rate():
    """rcv_mfma_calibrate (bench.py's yardstick for the box): a bare fp16 MFMA loop on every SIMD -- between a third of and the
    whole dense peak, at a shader clock the chip can hold."""
    import ctypes

    from rcvpose_amd import _hip

    tf, mhz = ctypes.c_double(), ctypes.c_double()
    _hip.check(_hip.lib().rcv_mfma_calibrate(0.05, ctypes.byref(tf), ctypes.byref(mhz), _hip.current_stream()), "rcv_mfma_calibrate")
    print("bare MFMA loop: %.0f TFLOP/s at %.0f MHz" % (tf.value, mhz.value))
    assert 800.0 < tf.value < 2600.0 and 1000.0 < mhz.value < 2600.0
    assert _hip.lib().rcv_mfma_calibrate(0.0, ctypes.byref(tf), None, None) == _hip.RCV_E_INVALID
