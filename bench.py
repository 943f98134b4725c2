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

Usage:  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}  # dense peaks, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames-per-step", type=int, default=72, help="frames per GPU per step")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--overlap", action="store_true", help="FCN of batch s+1 and voting of batch s on two HIP streams")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU vote sample")
    return ap.parse_args()


def cpu_baseline(frame, rmax, budget_s):
    """The oracle (kind "port") timed on this box's host cores on a bounded sample of the same workload:
    one torch-CPU fp32 FCN forward (all cores) and the brute-force N x D^3 vote on a subset of the frame's
    points (cost is exactly proportional to the number of points, so it is scaled to the full cloud)."""
    import numpy as np
    import torch

    from oracle import cpu_oracle, fcn_torch
    from rcvpose_amd import synthetic_weights

    shapes = json.load(open(os.path.join(ROOT, "tests", "golden", "fcn_state_dict_keys.json")))
    sd = {k: torch.from_numpy(v) for k, v in synthetic_weights.make_state_dict_numpy(shapes, seed=0).items()}
    x = fcn_torch.preprocess(frame["rgb"])
    fcn_torch.forward(sd, x)  # warm-up
    ts = []
    for _ in range(2):
        t0 = time.perf_counter()
        fcn_torch.forward(sd, x)
        ts.append(time.perf_counter() - t0)
    t_fcn = min(ts)
    xyz, rl = cpu_oracle.seam(frame["seg"], frame["radial"][0], frame["depth"], rmax[0])
    pts, rv, info = cpu_oracle.prepare(xyz, rl)
    n = len(pts)
    probe = 16
    t0 = time.perf_counter()
    cpu_oracle.fast_for(pts[:probe], rv[:probe], info.dim)
    per_point = (time.perf_counter() - t0) / probe
    sample = int(max(probe, min(n, budget_s / max(per_point, 1e-9))))
    idx = np.linspace(0, n - 1, sample).astype(np.int64)
    t0 = time.perf_counter()
    cpu_oracle.fast_for(pts[idx], rv[idx], info.dim)
    t_vote_sample = time.perf_counter() - t0
    t_vote = t_vote_sample * n / sample
    fps = 1.0 / (3.0 * t_fcn + 3.0 * t_vote)
    return {
        "value": fps, "unit": "frames/s", "cores": int(max(cpu_oracle.num_threads(), torch.get_num_threads())),
        "kind": "port",
        "sample": "1 torch-CPU fp32 FCN forward (%.2f s) x3 + brute-force N*D^3 vote on %d of %d points, D=%d "
                  "(%.1f s, scaled by N to %.0f s) x3 keypoints" % (t_fcn, sample, n, info.dim, t_vote_sample, t_vote),
    }


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    assert args.gpus == world, "--gpus must equal WORLD_SIZE (use torch.distributed.run for N > 1)"

    from rcvpose_amd import synthetic, synthetic_weights
    from rcvpose_amd.pipeline import FramePipeline

    B, G, H, W = args.frames_per_step, 3, 480, 640
    shapes = json.load(open(os.path.join(ROOT, "tests", "golden", "fcn_state_dict_keys.json")))
    sds = [{k: torch.from_numpy(v) for k, v in synthetic_weights.make_state_dict_numpy(shapes, seed=s).items()}
           for s in range(G)]
    pipe = FramePipeline(sds, H, W, dtype=args.dtype, frames_per_step=B, overlap=args.overlap)
    del sds

    # frames are sharded over ranks: global frame f -> rank f % world; a pool of distinct
    # frames per rank is cycled, all resident in HBM before the timed region
    pool = max(B, 4)
    fr = [synthetic.synthetic_frame(rank + world * i) for i in range(pool)]
    rmax = np.max([f["max_radii_dm"] for f in fr], axis=0) + 0.05
    rgb_pool = torch.from_numpy(np.stack([f["rgb"] for f in fr])).cuda()
    depth_pool = torch.from_numpy(np.stack([f["depth"] for f in fr]).view(np.int16)).cuda()
    seg_pool = torch.from_numpy(np.stack([f["seg"] for f in fr])).cuda()            # [P,H,W]
    rad_pool = torch.from_numpy(np.stack([f["radial"] for f in fr])).cuda()         # [P,3,H,W]

    def batch(step):
        ids = [(step * B + i) % pool for i in range(B)]
        it = torch.tensor(ids, device="cuda")
        seg = seg_pool[it][None].expand(G, B, H, W)[:, :, None]
        rad = rad_pool[it].permute(1, 0, 2, 3)[:, :, None]
        return ids, rgb_pool[it], depth_pool[it], (seg, rad)

    from rcvpose_amd import sharding

    # results of every step land in pinned host memory through an async copy on the voting stream
    host_out = [torch.empty((B * world, G, 3), dtype=torch.float64).pin_memory() for _ in range(args.steps + args.warmup)]
    rmax_dev = torch.tensor(rmax, dtype=torch.float64, device="cuda")

    def step_fn(step):
        """One batch: FCN on one HIP stream; seam + voting + the single collective of the path (one RCCL all_gather of
        72 B per frame over xGMI, frames back in global order f = i*world + rank) + result copy on another.  Batch s+1's
        forward overlaps batch s's voting; nothing waits on the host inside the timed region."""
        ids, rgb, depth, inject = batch(step)

        def collect(k):
            out = sharding.gather_keypoints(k, B * world)
            host_out[step].copy_(out, non_blocking=True)

        pipe.enqueue_step(rgb, depth, rmax_dev, inject=inject, collect=collect)
        return ids

    def fence():
        pipe.finish()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for s in range(args.warmup):
        step_fn(s)
    fence()
    pipe.engine.set_profiling(True)   # HIP events around every FCN op, on the launch stream
    fence()
    t0 = time.perf_counter()
    last_ids = None
    for s in range(args.steps):
        last_ids = step_fn(args.warmup + s)
    fence()
    dt = time.perf_counter() - t0
    prof = pipe.engine.get_profile()
    pipe.engine.set_profiling(False)
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    frames_total = args.steps * B * world
    fps = frames_total / dt
    last = (last_ids, host_out[args.warmup + args.steps - 1])

    # sanity of the last step on rank 0: recovered keypoints within one voxel of the truth
    ids, out = last
    mine = out[rank::world]
    truth_vox = 0.0
    for i, fid in enumerate(ids):
        truth_vox = max(truth_vox, float(np.abs(mine[i].numpy() - fr[fid]["keypoints_m"] * 1000).max() / 5.0))

    if rank == 0:
        conv_flop_step = prof["conv_flop_per_image_group"] * B * G
        conv_s = prof["conv_ms"] * 1e-3
        achieved = conv_flop_step / conv_s / 1e12
        peak = MFMA_PEAK_TFLOPS[args.dtype]
        # HBM bytes of the conv family from the committed PMC passes (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
        # runs, gfx950 correction applied: profiles/r01n_conv_hbm_traffic.json); only valid for the profiled workload
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01n_conv_hbm_traffic.json")
        if os.path.exists(tpath) and args.dtype == "bf16":
            tj = json.load(open(tpath))["conv_family"]
            traffic = tj["hbm_bytes_per_network_forward"] * B * G  # bytes per step on this GPU
        line = {
            "metric": "end-to-end frames/sec (FCN+3-kpt voting), 640x480 RGB-D",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "1-GPU full pipeline: 3x DenseFCNResNet152 forward + seam + LDS radial voting + arg-max, "
                                   "3 keypoints, ~15.5k back-projected pts, 5 mm voxel grid (D~228), 640x480",
                       "frames_per_step_per_gpu": B, "weights": "random (seeded He), ideal maps injected after the forward",
                       "overlap": bool(args.overlap), "parallelism": "frames sharded f%%%d, 1 RCCL all_gather/step" % world},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "traffic": traffic, "traffic_unit": "HBM bytes per step (all conv launches, PMC passes)",
                         "algorithmic_flop_per_step": conv_flop_step,
                         "kernel": "implicit-GEMM convolution family (k_conv_block, k_conv_dma, k_conv_halo, k_conv<%s,..>, k_tail), %d convolutions per "
                                   "forward, algorithmic 265.568 GFLOP per network forward" %
                                   (args.dtype, sum(1 for o in pipe.engine.ops if o["kind"] == 3)),
                         "fcn_conv_ms_per_step": prof["conv_ms"], "fcn_other_ms_per_step": prof["other_ms"]},
            "sanity": {"max_distance_to_true_keypoint_voxels": truth_vox},
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(fr[0], rmax, args.cpu_seconds)
            line["speedup_vs_cpu_baseline"] = fps / line["cpu_baseline"]["value"]
            # parity of the timed path against the oracle (the reference's algorithm) on the last step's first frame
            from oracle import cpu_oracle

            fid, worst = ids[0], 0.0
            for k in range(G):
                xyz, rl = cpu_oracle.seam(fr[fid]["seg"], fr[fid]["radial"][k], fr[fid]["depth"], rmax[k])
                ref = cpu_oracle.accumulator_3d(xyz, rl, sparse=True)[0]
                worst = max(worst, float(np.abs(mine[0, k].numpy() - ref).max() / 5.0))
            line["parity"] = {"keypoints_vs_oracle_max_voxels": worst, "gate_voxels": 1.0, "frames_checked": 1}
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
