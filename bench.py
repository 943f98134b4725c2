#!/usr/bin/env python3
"""Benchmark of the RCVPose `--mode=test` hot path on MI355X: end-to-end frames/s.

One *step* = one batch of `--frames-per-step` synthetic 640x480 RGB-D frames per GPU pushed through the
whole hot path: uint8 RGB -> normalise -> 3 x DenseFCNResNet152 forward (one network per keypoint,
reference AccumulatorSpace.py:520-530) -> per (frame, keypoint): mask/back-project seam -> radial voting
on a 5 mm voxel grid -> arg-max keypoint -> (multi-GPU) one RCCL all_gather of the keypoint coordinates.
Inputs are resident in HBM when the timed region starts.  Workload = BASELINE.json configs[2]
("1-GPU full pipeline: FCN + HIP radial accumulator, 3 keypoints, ~15k back-projected pts, 5 mm voxel grid").

Because the weights are random (no checkpoints without a network), the FCN output cannot select the
object: after the (timed, complete) forward the ideal seg / radial maps of the synthetic frame are copied
over the output buffers (also timed) and the seam/vote stages run on those.

Usage:  python bench.py --gpus N --steps K --warmup W
For N > 1 the driver launches one process per GPU through torch.distributed.run; a plain
`python bench.py --gpus N` starts those N processes itself (fresh children, before this process touches a GPU)
and relays rank 0's line.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# dense peaks, /opt/skills/guides/MI355X_MICROARCH.md.  f16x3 (the timed default) runs on the fp16 matrix pipe and is priced
# against ITS peak with the ALGORITHMIC flops (265.568 GFLOP per forward); it executes three MFMAs per product, so its
# ceiling is a third of that (roofline.mfma_flop_executed_per_algorithmic_flop)
MFMA_PEAK_TFLOPS = {"f16x3": 2500.0, "bf16": 2500.0, "fp16": 2500.0, "fp32": 157.3}
PARITY_DTYPES = ("f16x3", "fp32")   # the storage / arithmetic modes that carry north_star's 1e-4 map tolerance
MAP_TOLERANCE = 1e-4
HBM_PEAK_GBS = 8000.0
REGIONS = 3                # the timed region (exactly --steps steps) is run this many times; `value` is the median region
CALIB_REF_TFLOPS = 2000.0  # reference matrix rate the normalised convolution time is quoted at (a bare fp16 MFMA loop at ~1.9 GHz)
GATE_VOXELS = 1.0          # north_star: keypoints within 1 voxel of the reference's
TRUTH_GATE_VOXELS = 4.0    # sanity only: recovered keypoint vs the synthetic frame's true keypoint (0.02 dm radial noise)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames-per-step", type=int, default=53, help="frames per GPU per step (53: the dominant launches of the f16x3 forward land just under whole rounds of 256 workgroups, and the largest tensor stays below the 2 GiB a DMA descriptor addresses)")
    ap.add_argument("--dtype", default="f16x3", choices=["f16x3", "bf16", "fp16", "fp32"],
                    help="FCN arithmetic: f16x3 (default; fp16 hi/lo pairs, 3 MFMAs per product: the fast mode within north_star's "
                         "1e-4 map tolerance), fp32 (v_mfma_f32_16x16x4_f32), bf16 / fp16 (throughput modes outside the tolerance)")
    ap.add_argument("--overlap", action="store_true", help="FCN of batch s+1 and voting of batch s on two HIP streams")
    ap.add_argument("--q6", action="store_true", help="f16x3 only: fp6 cross terms in the stride-1 3x3 convolutions of layers 2-4 (engine option 11; "
                                                      "-3 % forward time for 2-3x the map error, still inside the 1e-4 tolerance: off by default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the fp32-mode and batch-1 (configs[1]) side measurements")
    ap.add_argument("--cpu-seconds", type=float, default=9.0, help="budget of the CPU vote samples (all frames together)")
    ap.add_argument("--parity-frames", type=int, default=8, help="frames of the last step checked against the oracle")
    ap.add_argument("--rccl-one-rank", action="store_true",
                    help="with --gpus 1: build the RCCL communicator anyway and run the per-step all_gather, barrier and "
                         "all_reduce of the N-rank path in a one-rank group (exercises that path on a one-GPU box)")
    ap.add_argument("--selftest-launcher", action="store_true",
                    help="CPU only: exercise the launcher / rendezvous / gather with gloo ranks and no GPU work (tests)")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` without torch.distributed.run around it
# ---------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args, argv):
    """Start N fresh ranks of this script (one per GPU) and return their exit code.  Runs before this process has made any
    HIP call: a process that initialised the GPU must never be replaced or forked into the ranks."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL across processes on this pool)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(1, args.gpus))))
    return subprocess.run(cmd, env=env).returncode


def selftest(args):
    """The N > 1 plumbing without a GPU: rendezvous from the torchrun environment, the per-step collective
    (sharding.gather_keypoints) on deterministic stand-in keypoints, the max-over-ranks timing reduction, rank 0's line."""
    import numpy as np
    import torch
    import torch.distributed as dist

    from rcvpose_amd import sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")
    assert args.gpus == world, "--gpus must equal WORLD_SIZE"
    B = args.frames_per_step

    def fake(frame):  # stands in for the pipeline's result for global frame `frame`
        return np.random.default_rng(1000 + frame).normal(size=(3, 3)) * 100

    t0 = time.perf_counter()
    ok = True
    for s in range(args.steps):
        local = torch.from_numpy(np.stack([fake((s * B + i) * world + rank) for i in range(B)]))
        full = sharding.gather_keypoints(local, B * world)
        expect = np.stack([fake(s * B * world + f) for f in range(B * world)])
        ok = ok and np.array_equal(full.numpy(), expect)
    if world > 1:
        dist.barrier()
    tmax = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"selftest": True, "n_gpus": world, "ranks": dist.get_world_size() if world > 1 else 1, "steps": args.steps,
                          "gather_equals_world1": bool(ok), "seconds": float(tmax.item())}))
    if world > 1:
        dist.destroy_process_group()
    return 0 if ok else 1


# ---------------------------------------------------------------------------------------------------------
# CPU baseline (the oracle timed on the host cores)
# ---------------------------------------------------------------------------------------------------------
def cpu_baseline(frames, budget_s):
    """The oracle (kind "port") timed on this box's host cores on a bounded sample of the same workload, per frame:
    one torch-CPU fp32 FCN forward (all cores) and the brute-force N x D^3 vote on a subset of the frame's points
    (cost is exactly proportional to the number of points, so it is scaled to the full cloud).  Median over the frames."""
    import numpy as np
    import torch

    from oracle import cpu_oracle, fcn_torch
    from rcvpose_amd import synthetic_weights

    shapes = json.load(open(os.path.join(ROOT, "tests", "golden", "fcn_state_dict_keys.json")))
    sd = {k: torch.from_numpy(v) for k, v in synthetic_weights.make_state_dict_numpy(shapes, seed=0).items()}
    fcn_torch.forward(sd, fcn_torch.preprocess(frames[0]["rgb"]))  # warm-up
    per_frame, notes, rates = [], [], []
    for f in frames:
        x = fcn_torch.preprocess(f["rgb"])
        t0 = time.perf_counter()
        fcn_torch.forward(sd, x)
        t_fcn = time.perf_counter() - t0
        xyz, rl = cpu_oracle.seam(f["seg"], f["radial"][0], f["depth"], f["max_radii_dm"][0])
        pts, rv, info = cpu_oracle.prepare(xyz, rl)
        n = len(pts)
        probe = 64
        t0 = time.perf_counter()
        cpu_oracle.fast_for(pts[:probe], rv[:probe], info.dim)
        per_point = (time.perf_counter() - t0) / probe
        sample = int(max(probe, min(n, budget_s / len(frames) / max(per_point, 1e-9))))
        idx = np.linspace(0, n - 1, sample).astype(np.int64)
        t0 = time.perf_counter()
        cpu_oracle.fast_for(pts[idx], rv[idx], info.dim)
        t_vote_sample = time.perf_counter() - t0
        t_vote = t_vote_sample * n / sample
        per_frame.append(1.0 / (3.0 * t_fcn + 3.0 * t_vote))
        rates.append(sample * float(info.dim) ** 3 / t_vote_sample)
        notes.append("fwd %.2f s, vote %d/%d pts D=%d %.1f s -> %.0f s" % (t_fcn, sample, n, info.dim, t_vote_sample, t_vote))
    return {
        "value": float(np.median(per_frame)), "unit": "frames/s",
        "cores": int(max(cpu_oracle.num_threads(), torch.get_num_threads())), "kind": "port",
        "sample": "median over %d frames of: 1 torch-CPU fp32 FCN forward x3 networks + brute-force N*D^3 vote on a point "
                  "subset scaled by N, x3 keypoints [%s]" % (len(frames), "; ".join(notes)),
        "brute_force_tests_per_s": float(np.median(rates)),
    }


# ---------------------------------------------------------------------------------------------------------
# side measurements that the driver should see (fp32-mode fps, configs[1] batch-1 latency)
# ---------------------------------------------------------------------------------------------------------
def _pipeline_fps(sds, frames, H, W, dtype, B, steps, engine="auto", stress=False, stages=None, q6=False, precision=None):
    """Whole-path frames/s of one mode at its own batch: FramePipeline(dtype, engine), ideal maps injected as in the timed run.
    stress: BASELINE configs[4] -- 1 mm voxels, the frames' own (YCB-Video) intrinsics, radial maps as float16 end to end.
    stages: optional dict that receives the per-step stage times (ms)."""
    import numpy as np
    import torch

    from rcvpose_amd.pipeline import FramePipeline

    B = min(B, len(frames))
    kw = {}
    if stress:
        fx, fy, cx, cy = frames[0]["K"]
        kw = dict(K=np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]]), acc_unit=1.0, radial_f16=True)
    pipe = FramePipeline(sds, H, W, dtype=dtype, frames_per_step=B, engine=engine, q6=q6, **kw)
    rgb = torch.from_numpy(np.stack([f["rgb"] for f in frames[:B]])).cuda()
    depth = torch.from_numpy(np.stack([f["depth"] for f in frames[:B]]).view(np.int16)).cuda()
    seg = torch.from_numpy(np.stack([f["seg"] for f in frames[:B]])).cuda()[None].expand(3, B, H, W)[:, :, None]
    rad_np = np.stack([f["radial"] for f in frames[:B]])
    if stress:  # the maps are float16 from the head on; the radius bound is taken on the rounded values
        rad_np = rad_np.astype(np.float16)
        rmax = torch.tensor(rad_np.astype(np.float32).reshape(B, 3, -1).max(axis=2).astype(np.float64), device="cuda")
    else:
        rmax = torch.tensor(np.stack([f["max_radii_dm"] for f in frames[:B]]), dtype=torch.float64, device="cuda")
    rad = torch.from_numpy(rad_np).cuda().permute(1, 0, 2, 3)[:, :, None]
    pipe.enqueue_step(rgb, depth, rmax, inject=(seg, rad))
    pipe.finish()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        pipe.enqueue_step(rgb, depth, rmax, inject=(seg, rad))
    pipe.finish()
    torch.cuda.synchronize()
    fps = steps * B / (time.perf_counter() - t0)
    kp, _ = pipe.read_results()
    if stages is not None:
        pipe.set_stage_timing(True)
        pipe.enqueue_step(rgb, depth, rmax, inject=(seg, rad))
        stages.update(pipe.stage_times())
        pipe.set_stage_timing(False)
    if precision is not None:
        precision.update(measured_precision(pipe, sds, H, W))
    del pipe
    torch.cuda.empty_cache()
    return fps, B, kp


def extras(sds, frames, H, W, dtype, B_timed, kp_timed):
    """Side measurements next to the timed mode: BASELINE configs[1] (bf16, batch 1, forward only), the other arithmetic
    modes end to end at their own best batch, and the LDS-rasteriser arg-max engine (the engine north_star describes; the
    timed default is the exact branch-and-bound search in front of it -- identical results)."""
    import numpy as np
    import torch

    from rcvpose_amd.fcn_engine import FcnEngine

    out = {}
    # configs[1]: one DenseFCNResNet152, bf16, batch 1, 640x480, forward only.  The forward's ~175 launches are replayed as one HIP
    # graph (FcnEngine.capture): launched one by one from Python they take the same device time when the host keeps up
    # (2.41 ms both ways) and 50 % more on a box whose host does not -- the eager figure is kept beside it.
    def batch1(mode, split_k=False):
        eng = FcnEngine(sds[:1], H, W, dtype=mode)
        if split_k:
            eng.set_option(12, 1)  # opt-in latency option: split-K in the few-tile launches (another summation order: tolerance, not bits)
        x = torch.randn((1, 3, H, W), device="cuda")
        for _ in range(5):
            eng.forward(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            eng.forward(x)
        e1.record()
        torch.cuda.synchronize()
        eager = e0.elapsed_time(e1) / 30
        graph, _, _ = eng.capture(x)
        graph.replay()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(30):
            graph.replay()
        e1.record()
        torch.cuda.synchronize()
        replay = e0.elapsed_time(e1) / 30
        flops = eng.flops_per_image_group
        del graph, eng
        torch.cuda.empty_cache()
        return replay, eager, flops

    ms, eager, flops = batch1("bf16")
    out["configs1_batch1_bf16_forward_ms"] = ms
    out["configs1_batch1_bf16_forward_ms_launched_from_python"] = eager
    out["configs1_batch1_bf16_tflops"] = flops / ms / 1e9
    # the same single forward in the parity-grade mode
    ms, eager, _ = batch1("f16x3")
    out["batch1_f16x3_forward_ms"] = ms
    out["batch1_f16x3_forward_ms_launched_from_python"] = eager
    # ... and both with the opt-in latency option (engine option 12: deterministic split-K; maps within tolerance, not bit-identical)
    out["configs1_batch1_bf16_forward_ms_split_k_option"] = batch1("bf16", True)[0]
    out["batch1_f16x3_forward_ms_split_k_option"] = batch1("f16x3", True)[0]
    # the other arithmetic modes, whole path (same frames, ideal maps injected)
    for mode, bb, steps in (("bf16", 72, 4), ("fp16", 72, 4), ("fp32", 24, 2), ("f16x3", 36, 3)):
        if mode == dtype:
            continue
        fps, b, _ = _pipeline_fps(sds, frames, H, W, mode, bb, steps)
        out["%s_mode_frames_per_s" % mode] = fps
        out["%s_mode_frames_per_step" % mode] = b
    out["mode_note"] = ("f16x3 and fp32 meet north_star's 1e-4 map tolerance (tests/test_gpu_fcn_x3.py, test_gpu_fcn.py); bf16 / fp16 are "
                        "throughput modes outside it (map error ~2e-2 / ~3e-3, keypoints move 1.7 / 0.5 voxels on average: DESIGN 2)")
    # the LDS shell rasteriser as the arg-max engine (rcv_accumulate_3d engine RCV_VOTE_RASTER) under the timed FCN mode
    fps, b, kp = _pipeline_fps(sds, frames, H, W, dtype, min(B_timed, 24), 2, engine="full")
    out["rasteriser_engine_frames_per_s"] = fps
    out["rasteriser_engine_frames_per_step"] = b
    if kp_timed is not None:
        n = min(b, kp_timed.shape[0])
        out["rasteriser_engine_keypoints_equal_timed_engine"] = bool(np.array_equal(kp[:n], kp_timed[:n]))
    if dtype == "f16x3":
        # the split engine with fp6 cross terms in the stride-1 3x3 convolutions of layers 2-4 (engine option 11, --q6): faster and
        # inside the tolerance, but it spends most of it -- reported here, not the timed default
        pr = {}
        fps, b, kp = _pipeline_fps(sds, frames, H, W, dtype, B_timed, 4, q6=True, precision=pr)
        out["q6_mode_frames_per_s"] = fps
        out["q6_mode"] = {"frames_per_step": b, "radial_map_rel_err_vs_fp32_engine": pr.get("radial_map_rel_err_vs_fp32_engine"),
                          "seg_map_rel_err_vs_fp32_engine": pr.get("seg_map_rel_err_vs_fp32_engine"), "tolerance": MAP_TOLERANCE}
    # one point off the benign end of the voting cost (profiles/*_vote_sweep.json): the timed step on frames whose radial maps
    # carry one voxel (0.05 dm) of noise instead of 0.02 dm
    from rcvpose_amd import synthetic

    noisy = [synthetic.synthetic_frame(5000 + i, noise_dm=0.05) for i in range(B_timed)]
    st = {}
    fps, b, _ = _pipeline_fps(sds, noisy, H, W, dtype, B_timed, 3, stages=st)
    out["noise_0p05dm_frames_per_s"] = fps
    out["noise_0p05dm_frames_per_step"] = b
    out["noise_0p05dm_acc_time_ms_per_step"] = st.get("acc")
    del noisy
    # BASELINE configs[4] (stress): 1 mm voxels, ~50k-point YCB-Video-shaped clouds, 8 frames per step, float16 radial maps
    ycb = [synthetic.synthetic_frame(300 + i, shape="ycb") for i in range(8)]
    st = {}
    fps, b, kp = _pipeline_fps(sds, ycb, H, W, dtype, 8, 2, stress=True, stages=st)
    out["configs4_stress_frames_per_s"] = fps
    out["configs4_stress"] = {"frames_per_step": b, "voxel_mm": 1.0, "points_per_item": int(np.mean([f["n_silhouette"] for f in ycb])),
                              "radial_maps": "float16", "net_time_ms_per_step": st.get("net"), "seam_ms_per_step": st.get("seam"),
                              "acc_time_ms_per_step": st.get("acc"),
                              "max_distance_to_true_keypoint_mm": float(max(np.abs(kp[i] - ycb[i]["keypoints_m"] * 1000).max() for i in range(b)))}
    return out


def measured_precision(pipe, sds, H, W, n_frames=2):
    """Map error of the TIMED arithmetic against the fp32 engine (v_mfma_f32_16x16x4_f32, itself within 1e-5 of the reference's
    golden forward: tests/test_gpu_fcn.py) on frames of the last step: max |d| / max |fp32| per network, seg and radial."""
    import torch

    from rcvpose_amd.fcn_engine import FcnEngine

    x = pipe.x[:n_frames].clone()
    seg_t, rad_t = pipe.engine.forward(pipe.x)
    seg_t, rad_t = seg_t[:, :n_frames].float().clone(), rad_t[:, :n_frames].float().clone()
    ref = FcnEngine(sds, H, W, dtype="fp32")
    seg_r, rad_r = ref.forward(x)
    err_r = [float((rad_t[g] - rad_r[g]).abs().max() / rad_r[g].abs().max()) for g in range(len(sds))]
    err_s = [float((seg_t[g] - seg_r[g]).abs().max() / seg_r[g].abs().max()) for g in range(len(sds))]
    del ref
    torch.cuda.empty_cache()
    return {"radial_map_rel_err_vs_fp32_engine": max(err_r), "seg_map_rel_err_vs_fp32_engine": max(err_s),
            "per_network_radial": err_r, "frames": n_frames, "tolerance": MAP_TOLERANCE,
            "definition": "max |map_timed - map_fp32| / max |map_fp32| over whole 480x640 maps (SURVEY 8d parity gate); the fp32 engine "
                          "is the golden-pinned one: within 1e-5 of the reference module's golden forward (tests/test_gpu_fcn.py), as is "
                          "the timed f16x3 engine itself (6-8e-6, tests/test_gpu_fcn_x3.py)",
            "range_flags_of_timed_engine": pipe.engine.range_status(clear=False)}


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, argv))  # nothing in this process has touched a GPU yet
    if args.selftest_launcher:
        sys.exit(selftest(args))
    # stdout carries the JSON line and nothing else: libraries that print to file descriptor 1 (RCCL announces its path there
    # when the communicator is torn down) are sent to stderr; the line itself goes out through a copy of the real stdout
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    torch.cuda.set_device(local)
    use_dist = world > 1 or args.rccl_one_rank
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))

    from rcvpose_amd import sharding, synthetic, synthetic_weights
    from rcvpose_amd.pipeline import FramePipeline

    B, G, H, W = args.frames_per_step, 3, 480, 640
    shapes = json.load(open(os.path.join(ROOT, "tests", "golden", "fcn_state_dict_keys.json")))
    sds = [{k: torch.from_numpy(v) for k, v in synthetic_weights.make_state_dict_numpy(shapes, seed=s).items()}
           for s in range(G)]
    pipe = FramePipeline(sds, H, W, dtype=args.dtype, frames_per_step=B, overlap=args.overlap, q6=args.q6 and args.dtype == "f16x3")

    # frames are sharded over ranks: global frame f -> rank f % world; local frame i of rank r is the
    # synthetic frame r + world * i.  A pool of distinct frames per rank is cycled, resident in HBM before the timed region.
    pool = max(2 * B, 4)  # two distinct frame sets alternate across the steps (a step never repeats the previous one's frames)
    fr = [synthetic.synthetic_frame(rank + world * i) for i in range(pool)]

    def upload(frames):
        return (torch.from_numpy(np.stack([f["rgb"] for f in frames])).cuda(),
                torch.from_numpy(np.stack([f["depth"] for f in frames]).view(np.int16)).cuda(),
                torch.from_numpy(np.stack([f["seg"] for f in frames])).cuda(),                      # [P,H,W]
                torch.from_numpy(np.stack([f["radial"] for f in frames])).cuda(),                   # [P,3,H,W]
                torch.tensor(np.stack([f["max_radii_dm"] for f in frames]), dtype=torch.float64, device="cuda"))  # [P,3]

    rgb_pool, depth_pool, seg_pool, rad_pool, rmax_pool = upload(fr)

    def batch_of(pools, ids):
        rgb_p, depth_p, seg_p, rad_p, rmax_p = pools
        it = torch.tensor(ids, device="cuda")
        seg = seg_p[it][None].expand(G, B, H, W)[:, :, None]
        rad = rad_p[it].permute(1, 0, 2, 3)[:, :, None]
        return rgb_p[it], depth_p[it], rmax_p[it], (seg, rad)

    def ids_of(step):
        return [(step * B + i) % pool for i in range(B)]

    # the batches of all steps are assembled (gathered out of the pool) BEFORE the timed region: inputs resident in HBM
    batches = {}
    for s in range(args.warmup + args.steps):
        key = tuple(ids_of(s))
        if key not in batches:
            batches[key] = batch_of((rgb_pool, depth_pool, seg_pool, rad_pool, rmax_pool), list(key))

    # results of every step land in pinned host memory through an async copy on the voting stream
    host_out = [torch.empty((B * world, G, 3), dtype=torch.float64).pin_memory() for _ in range(args.steps + args.warmup)]

    def step_fn(step):
        """One batch: preprocess + 3 FCN forwards + injection, then seam + voting + the single collective of the path (one
        RCCL all_gather of 72 B per frame over xGMI, frames back in global order f = i*world + rank) + the result copy.
        One HIP stream by default; --overlap puts the forward of batch s+1 and the voting of batch s on two streams.
        Nothing waits on the host inside the timed region."""
        ids = ids_of(step)
        rgb, depth, rmax, inject = batches[tuple(ids)]

        def collect(k):
            out = sharding.gather_keypoints(k, B * world, always_collective=use_dist)
            host_out[step].copy_(out, non_blocking=True)

        pipe.enqueue_step(rgb, depth, rmax, inject=inject, collect=collect)
        return ids

    def fence():
        pipe.finish()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def calibrate():
        """The dense fp16 matrix rate THIS device holds in a bare MFMA loop (rcv_mfma_calibrate): boxes of the pool differ by
        several per cent there, which is as much as a round of kernel work moves."""
        import ctypes

        from rcvpose_amd import _hip

        tf, mhz = ctypes.c_double(), ctypes.c_double()
        _hip.check(_hip.lib().rcv_mfma_calibrate(0.25, ctypes.byref(tf), ctypes.byref(mhz), _hip.current_stream()), "rcv_mfma_calibrate")
        return tf.value, mhz.value

    for s in range(args.warmup):
        step_fn(s)
    fence()
    pipe.engine.set_profiling(True)   # HIP events around every FCN op, on the launch stream
    # EXACTLY `steps` steps are timed, bracketed by barrier + synchronize, max over ranks -- and that region is run REGIONS times
    # back to back: `value` is the median region, the others are reported (a 3 % kernel gain must not drown in run-to-run spread)
    regions = []
    last_ids = None
    for rg in range(REGIONS):
        pipe.set_stage_timing(True)   # ... and around the stages (net / inject / seam / acc / collect)
        fence()
        t0 = time.perf_counter()
        for s in range(args.steps):
            last_ids = step_fn(args.warmup + s)
        fence()
        dt = time.perf_counter() - t0
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        if use_dist:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        op_ms = pipe.engine.get_op_times()  # (of the region's last forward)
        # the launches that run at the matrix pipe's rate (k_conv_halo / k_tail_x3: every 3x3 / stride-1 convolution, 0.81-0.85 busy);
        # the rest of the step (1x1 chains, strided launches, glue, voting) is bound by LDS, HBM or the VALU
        mfma_ms = sum(t for o, t in zip(pipe.engine.ops, op_ms) if o["kind"] == 3 and o["ksize"] == 3 and o["stride"] == 1)
        regions.append(dict(seconds=float(tmax.item()), prof=pipe.engine.get_profile(), stages=pipe.stage_times(), mfma_bound_ms=mfma_ms))
        pipe.set_stage_timing(False)
    pipe.engine.set_profiling(False)
    # (r06: both samples AFTER the regions -- three quarters of a second of full-chip MFMA load right in front of region 0 would
    # pre-heat the device the regions are then timed on)
    cal = [calibrate(), calibrate()]
    order = sorted(range(REGIONS), key=lambda i: regions[i]["seconds"])
    med = regions[order[REGIONS // 2]]
    dt, prof, stages, mfma_bound_ms = med["seconds"], med["prof"], med["stages"], med["mfma_bound_ms"]
    frames_total = args.steps * B * world
    fps = frames_total / dt
    calib_tf = float(np.mean([c[0] for c in cal]))
    calib_mhz = float(np.mean([c[1] for c in cal]))
    out = host_out[args.warmup + args.steps - 1]     # [B*world, G, 3], global frame order f = i*world + rank
    ids = last_ids
    mine = out[rank::world]

    # sanity of the last step on every rank: recovered keypoints near the synthetic truth
    truth_vox = 0.0
    for i, fid in enumerate(ids):
        truth_vox = max(truth_vox, float(np.abs(mine[i].numpy() - fr[fid]["keypoints_m"] * 1000).max() / 5.0))

    # world-N == world-1: rank 0 recomputes frames of the other ranks on its own GPU and compares the gathered
    # rows bit for bit (a frame's result must not depend on which rank produced it)
    sharded_equal, sharded_checked = None, 0
    if world > 1 and rank == 0:
        nchk = min(8, B)
        todo = [(r, i) for r in range(1, world) for i in range(nchk)]
        sharded_equal = True
        for c0 in range(0, len(todo), B):
            chunk = todo[c0:c0 + B]
            frames = [synthetic.synthetic_frame(r + world * ids[i]) for r, i in chunk]
            frames += [fr[0]] * (B - len(frames))
            pools = upload(frames)
            rgb, depth, rmax, inject = batch_of(pools, list(range(B)))
            pipe.process(rgb, depth, rmax, inject=inject)
            torch.cuda.synchronize()
            kp = pipe.keypoints_tensor().cpu()
            for j, (r, i) in enumerate(chunk):
                sharded_equal = sharded_equal and bool(torch.equal(kp[j], out[i * world + r]))
                sharded_checked += 1

    ok = True
    if rank == 0:
        conv_flop_step = prof["conv_flop_per_image_group"] * B * G
        conv_s = prof["conv_ms"] * 1e-3
        achieved = conv_flop_step / conv_s / 1e12
        peak = MFMA_PEAK_TFLOPS[args.dtype]
        # HBM bytes of the conv family: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the FCN of this mode
        # (gfx950 correction applied), committed under profiles/ with the batch they were taken at -- NOT counters of this run
        traffic, traffic_src = None, None
        for name, modes in (("r06_conv_hbm_traffic_%s.json" % args.dtype, (args.dtype,)), ("r05_conv_hbm_traffic_%s.json" % args.dtype, (args.dtype,)), ("r04_conv_hbm_traffic_%s.json" % args.dtype, (args.dtype,)),
                            ("r03_conv_hbm_traffic_%s.json" % args.dtype, (args.dtype,)),
                            ("r02_conv_hbm_traffic.json", ("bf16", "fp16"))):
            tpath = os.path.join(ROOT, "profiles", name)
            if os.path.exists(tpath) and args.dtype in modes:
                tj = json.load(open(tpath))["conv_family"]
                traffic = tj["hbm_bytes_per_network_forward"] * B * G  # bytes per step on this GPU
                traffic_src = "profiles/%s (PMC passes at %s frames/step, a constant read from that file; per-forward bytes x %d forwards)" % (
                    name, tj.get("frames_per_step", "8"), B * G)
                break
        n_steps = float(args.steps)
        stage_ms = {k: v / n_steps for k, v in stages.items()}
        # ---- second roofline: seam + voting (HBM-bound by definition) -------------------------------
        n_pts = [f["n_silhouette"] for f in (fr[i] for i in ids)]
        alg_bytes = sum(H * W * 10 + 40 * n + 96 for n in n_pts) * G           # per step on this GPU
        votes = 0.0
        for i in ids[:8]:                                                       # sum 5.44 r^2 (voxels) over points, 8 frames sampled
            f = fr[i]
            for k in range(G):
                r = f["radial"][k][f["depth"] != 0].astype(np.float64) * 100.0 / 5.0
                votes += 5.44 * float((r * r).sum())
        votes = votes / min(8, len(ids)) * B                                    # per step
        vote_s = (stage_ms.get("seam", 0.0) + stage_ms.get("acc", 0.0)) * 1e-3
        roof_vote = {
            "bound": "hbm", "achieved": alg_bytes / vote_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": alg_bytes / vote_s / 1e9 / HBM_PEAK_GBS, "traffic": None,
            "kernel": "seam + Accumulator_3D (k_seam_*, k_prep, k_setup, k_bb_*; dominant: k_bb_wide), %d (frame, keypoint) items per step" % (B * G),
            "ms_per_step": vote_s * 1e3, "algorithmic_bytes_per_step": alg_bytes,
            "algorithmic_bytes_per_item": "H*W*10 B seam read + 40 B/point list written+read + 96 B result (DESIGN 4.2)",
            "equivalent_votes_per_s": votes / vote_s,
            "equivalent_votes_note": "sum of 5.44 r^2 shell voxels per point / stage time: the votes a rasteriser would cast; the timed "
                                     "engine (exact branch-and-bound peak search) never casts them",
        }
        vp = next((q for q in (os.path.join(ROOT, "profiles", n) for n in ("r06_vote_pmc.json", "r05_vote_pmc.json", "r04_vote_pmc.json", "r03_vote_pmc.json", "r02_vote_pmc.json")) if os.path.exists(q)), None)
        if vp:  # separate rocprofv3 --pmc passes of the voting stage (tools/collect_profiles.sh; r04: at the bench batch)
            pj = json.load(open(vp))
            roof_vote["pmc"] = {"source": "profiles/%s: " % os.path.basename(vp) + pj["source"], "definitions": pj["definitions"],
                                "kernels": {k: {q: v[q] for q in ("ms_per_batch", "valu_issue_frac", "hbm_bytes_per_batch") if q in v}
                                            for k, v in pj["kernels"].items() if v["ms_per_batch"] >= 0.1}}
            if B == int(pj.get("frames_per_batch", 72)):
                roof_vote["traffic"] = pj["hbm_bytes_per_batch_sum"]
                roof_vote["traffic_note"] = ("HBM-side bytes per step of all voting kernels (FETCH_SIZE / WRITE_SIZE passes, Infinity-Cache hits "
                                             "included): the point lists are re-streamed by every pass of every search level out of L2 / "
                                             "Infinity Cache; the stage is VALU-bound (valu_issue_frac of k_bb_wide), not traffic-bound")
        line = {
            "metric": "end-to-end frames/sec (FCN+3-kpt voting), 640x480 RGB-D",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "timed_regions": {"n": REGIONS, "steps_each": args.steps, "value_is": "median region",
                              "frames_per_s": [frames_total / r["seconds"] for r in regions],
                              "min": frames_total / max(r["seconds"] for r in regions), "max": frames_total / min(r["seconds"] for r in regions)},
            "calibration": {"mfma_loop_tflops": calib_tf, "shader_clock_mhz_in_that_loop": calib_mhz, "samples": [list(c) for c in cal],
                            "what": "rcv_mfma_calibrate: bare v_mfma_f32_16x16x32_f16 loop on every SIMD, 0.25 s, twice, after the timed regions "
                                    "(rank 0's device): the dense fp16 rate THIS box holds; boxes differ by several per cent",
                            "reference_tflops": CALIB_REF_TFLOPS,
                            "mfma_bound_ms_per_step": mfma_bound_ms,
                            "ms_per_step_normalised": mfma_bound_ms * calib_tf / CALIB_REF_TFLOPS + (dt / args.steps * 1e3 - mfma_bound_ms),
                            "fcn_conv_ms_per_step_normalised": mfma_bound_ms * calib_tf / CALIB_REF_TFLOPS + (prof["conv_ms"] - mfma_bound_ms),
                            "ms_per_step_times_loop_rate_over_reference": dt / args.steps * 1e3 * calib_tf / CALIB_REF_TFLOPS,
                            "note": "normalised = (time of the launches bound by the matrix pipe: the 3x3 / stride-1 convolutions) x (this box's "
                                    "loop rate / reference rate) + the rest of the step unscaled -- boxes differ in the clock they hold under MFMA "
                                    "load, and only that part of the step follows it (three boxes of r05: raw 126.1 / 126.4 / 130.1 ms at loop rates "
                                    "1549 / 1527 / 1458: scaling the WHOLE step over-corrects, 97.7 / 96.5 / 94.9).  Compare THESE across boxes and rounds"},
            "dtype": args.dtype, "data": "synthetic", "rccl_ranks": dist.get_world_size() if use_dist else 1,
            "config": {"workload": "1-GPU full pipeline (BASELINE configs[2]): 3x DenseFCNResNet152 forward + seam + radial-vote arg-max, "
                                   "3 keypoints, ~15.5k back-projected pts, 5 mm voxel grid (D~228), 640x480",
                       "fcn_arithmetic": {"f16x3": "fp16 hi/lo operand pairs, 3 MFMAs per product, fp32 accumulate (maps within 1e-4 of fp32)",
                                          "fp32": "v_mfma_f32_16x16x4_f32", "bf16": "bf16 storage, fp32 accumulate (outside the 1e-4 tolerance)",
                                          "fp16": "fp16 storage, fp32 accumulate (outside the 1e-4 tolerance)"}[args.dtype],
                       "vote_engine": "bb: exact branch-and-bound peak search (k_bb_*) returning the rasteriser's (peak, ties, first index); "
                                      "the LDS shell rasteriser (k_vote) is its fallback and is timed in extras.rasteriser_engine_frames_per_s",
                       "frames_per_step_per_gpu": B, "distinct_frames_cycled": pool,
                       "weights": "random (seeded He), ideal maps injected after the forward",
                       "overlap": bool(args.overlap), "q6_cross_terms": bool(args.q6 and args.dtype == "f16x3"), "parallelism": "frames sharded f%%%d, 1 RCCL all_gather/step" % world},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "traffic": traffic, "traffic_unit": "HBM bytes per step (all conv launches)", "traffic_source": traffic_src,
                         "algorithmic_flop_per_step": conv_flop_step,
                         "mfma_flop_executed_per_algorithmic_flop": 3 if args.dtype == "f16x3" else 1,
                         "frac_of_mode_ceiling": achieved / peak * (3 if args.dtype == "f16x3" else 1),
                         "kernel": "implicit-GEMM convolution family of the %s engine (k_conv_halo, k_conv_dma, k_conv<..>, %s), %d convolutions per "
                                   "forward, algorithmic 265.568 GFLOP per network forward" %
                                   (args.dtype, {"f16x3": "k_chain2_x3, k_tail_x3", "fp32": "k_head"}.get(args.dtype, "k_conv_block, k_tail"),
                                    sum(1 for o in pipe.engine.ops if o["kind"] == 3)),
                         "fcn_conv_ms_per_step": prof["conv_ms"], "fcn_other_ms_per_step": prof["other_ms"]},
            "roofline_vote": roof_vote,
            "stage_ms_per_step": {"net_time (preprocess + 3 forwards)": stage_ms.get("net"), "inject (measurement only)": stage_ms.get("inject"),
                                  "seam": stage_ms.get("seam"), "acc_time (Accumulator_3D)": stage_ms.get("acc"),
                                  "collect (gather + D2H)": stage_ms.get("collect")},
            "sanity": {"max_distance_to_true_keypoint_voxels": truth_vox, "gate_voxels": TRUTH_GATE_VOXELS},
        }
        ok = ok and truth_vox <= TRUTH_GATE_VOXELS
        if world > 1:
            line["sharding"] = {"world_n_equals_world_1_bit_for_bit": sharded_equal, "frames_checked": sharded_checked}
            ok = ok and bool(sharded_equal)
        if not args.no_cpu_baseline and world == 1:
            # parity of the timed path against the oracle (the reference's algorithm) on frames of the last step
            from oracle import cpu_oracle

            worst, dims, checked = 0.0, [], 0
            for i in range(min(args.parity_frames, B)):
                f = fr[ids[i]]
                for k in range(G):
                    xyz, rl = cpu_oracle.seam(f["seg"], f["radial"][k], f["depth"], f["max_radii_dm"][k])
                    ref, _, info, _ = cpu_oracle.accumulator_3d(xyz, rl, sparse=True, return_all=True)
                    worst = max(worst, float(np.abs(mine[i, k].numpy() - ref[0]).max() / 5.0))
                    dims.append(info.dim)
                checked += 1
            line["vote_parity_on_injected_maps"] = {
                "keypoints_vs_oracle_max_voxels": worst, "gate_voxels": GATE_VOXELS, "frames_checked": checked, "keypoints_checked": checked * G,
                "note": "seam + voting of the timed path against the oracle on the injected ideal maps; the forward's own precision is `precision`"}
            ok = ok and worst <= GATE_VOXELS
            d3 = float(np.mean([d ** 3 for d in dims]))
            roof_vote["brute_force_equivalent_tests_per_s"] = float(np.mean(n_pts)) * d3 * B * G / vote_s
            line["cpu_baseline"] = cpu_baseline([fr[ids[i]] for i in range(min(3, B))], args.cpu_seconds)
            line["speedup_vs_cpu_baseline"] = fps / line["cpu_baseline"]["value"]
        # precision of the timed forward itself, measured here against the fp32 engine on frames of the last step
        line["precision"] = measured_precision(pipe, sds, H, W)
        line["precision_meets_tolerance"] = bool(line["precision"]["radial_map_rel_err_vs_fp32_engine"] <= MAP_TOLERANCE and
                                                 line["precision"]["seg_map_rel_err_vs_fp32_engine"] <= MAP_TOLERANCE)
        line["precision_gate_applies"] = args.dtype in PARITY_DTYPES  # (bf16 / fp16 are throughput modes outside the tolerance: reported, not gated)
        ok = ok and (line["precision_meets_tolerance"] or args.dtype not in PARITY_DTYPES) and line["precision"]["range_flags_of_timed_engine"] == 0
        sweep = next((q for q in (os.path.join(ROOT, "profiles", n) for n in ("r05_vote_sweep.json", "r04_vote_sweep.json", "r03_vote_sweep.json")) if os.path.exists(q)), "")
        if os.path.exists(sweep):  # data dependence of the peak search (tools/vote_sweep.py), a committed measurement, not this run
            sj = json.load(open(sweep))
            roof_vote["sensitivity"] = {"source": "profiles/%s (%s)" % (os.path.basename(sweep), sj.get("command", "")), "benign_ms_per_216_items": sj.get("benign_ms"),
                                        "worst_ms_per_216_items": sj.get("worst_ms"), "worst_case": sj.get("worst_case"),
                                        "fallback_rate_worst": sj.get("worst_fallback_rate")}
        if not args.no_extras and world == 1:
            kp_timed = mine.numpy().copy()
            del pipe
            torch.cuda.empty_cache()
            line["extras"] = extras(sds, [fr[i] for i in ids] + [f for j, f in enumerate(fr) if j not in set(ids)], H, W, args.dtype, B, kp_timed)
        line["ok"] = bool(ok)
        print(json.dumps(line), file=json_out, flush=True)
    if use_dist:
        flag = torch.tensor([1 if ok else 0], device="cuda")
        dist.broadcast(flag, src=0)
        ok = bool(flag.item())
        dist.destroy_process_group()
    sys.exit(0 if ok else 1)   # a regressed path must not leave a valid-looking fps line with rc 0


if __name__ == "__main__":
    main()
