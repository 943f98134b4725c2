#!/usr/bin/env python3
"""Data dependence of the voting stage: seam + Accumulator_3D per 216 (frame, keypoint) items as bench.py
runs them (one batched launch sequence, exact branch-and-bound peak search with the LDS rasteriser as fallback), over

    radial noise sigma {0.02, 0.05, 0.1, 0.2} dm  x  outlier fraction {0, 5, 20} %  x  seg false positives {0, 10} %

of synthetic LINEMOD-shaped frames (rcvpose_amd/synthetic.py).  Per case: ms per 216 items (HIP events on the launch stream),
how many items the search handed to the rasteriser (AccHeader.bb_done == 0), keypoint error against the synthetic truth, and --
for two items per case -- equality with the CPU oracle.  Writes one JSON (-> profiles/r03_vote_sweep.json).

    python tools/vote_sweep.py [--frames 72] [--distinct 6] [--out gpurun_out/vote_sweep.json]
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from rcvpose_amd import _hip, synthetic, synthetic_weights
from rcvpose_amd.pipeline import FramePipeline

# AccHeader (rcvpose_amd/csrc/accumulator.hip): int32 fields after the six doubles
OFF_N, OFF_DIM, OFF_STATUS, OFF_BB_ENABLED, OFF_BB_DONE, OFF_BB_OVERFLOW = 48, 60, 104, 112, 116, 124


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=72)
    ap.add_argument("--distinct", type=int, default=6, help="distinct synthetic frames per case (cycled to --frames)")
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--oracle-items", type=int, default=2)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "vote_sweep.json"))
    ap.add_argument("--raster-too", action="store_true", help="also time every case with the rasteriser as the only engine (engine 'full')")
    args = ap.parse_args()
    B, G = args.frames, 3
    shapes = json.load(open(os.path.join(ROOT, "tests", "golden", "fcn_state_dict_keys.json")))
    sds = [{k: torch.from_numpy(v) for k, v in synthetic_weights.make_state_dict_numpy(shapes, seed=s).items()} for s in range(G)]
    pipe = FramePipeline(sds, 480, 640, dtype="bf16", frames_per_step=B)  # (only its voting half is used)
    engine_auto, engine_full = _hip.VOTE_ENGINES["auto"], _hip.VOTE_ENGINES["full"]
    from oracle import cpu_oracle

    cases = []
    for sigma in (0.02, 0.05, 0.1, 0.2):
        for outl in (0.0, 0.05, 0.2):
            for fp in (0.0, 0.1):
                cases.append((sigma, outl, fp))
    rows = []
    for sigma, outl, fp in cases:
        fr = [synthetic.synthetic_frame(i, noise_dm=sigma, outlier_frac=outl, seg_false_pos=fp) for i in range(args.distinct)]
        fr = [fr[i % len(fr)] for i in range(B)]
        rmax = torch.tensor(np.stack([f["max_radii_dm"] for f in fr]), dtype=torch.float64, device="cuda")
        pipe.seg.copy_(torch.from_numpy(np.stack([np.stack([f["seg"] for f in fr])] * G))[:, :, None].cuda())
        pipe.radial.copy_(torch.from_numpy(np.stack([np.stack([f["radial"][k] for f in fr]) for k in range(G)]))[:, :, None].cuda())
        depth = torch.from_numpy(np.stack([f["depth"] for f in fr]).view(np.int16)).cuda()
        pipe.vote(depth, rmax)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            pipe.vote(depth, rmax)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.iters
        kp, res = pipe.read_results()
        hdr = pipe.ws[:, :512].cpu().numpy().view(np.int32)
        assert all(int(hdr[i, OFF_N // 4]) == res[i].n_points and int(hdr[i, OFF_DIM // 4]) == res[i].dim for i in range(B * G)), "AccHeader layout"
        fallback = int(sum(1 for i in range(B * G) if hdr[i, OFF_BB_DONE // 4] == 0))
        overflow = int(sum(1 for i in range(B * G) if hdr[i, OFF_BB_OVERFLOW // 4] != 0))
        err = max(float(np.abs(kp[i] - fr[i]["keypoints_m"] * 1000).max() / 5.0) for i in range(args.distinct))
        same = True
        for it in range(args.oracle_items):
            f, k = fr[it], it % G
            xyz, rl = cpu_oracle.seam(f["seg"], f["radial"][k], f["depth"], f["max_radii_dm"][k])
            ref = cpu_oracle.accumulator_3d(xyz, rl, sparse=True)
            same = same and bool(np.array_equal(kp[it, k], ref[0]))
        hand = [int(hdr[i, 324 // 4]) for i in range(B * G)]            # AccHeader::bb_handover
        gate_f = [hdr[i, 268 // 4] / max(1, res[i].n_points) for i in range(B * G)]   # AccHeader::gate_best / n
        rmx = pipe.ws[:, 24:32].cpu().numpy().view(np.float64)[:, 0]      # AccHeader::rmax
        ms_r = None
        if args.raster_too:
            pipe.vote_engine = engine_full
            pipe.vote(depth, rmax)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(args.iters):
                pipe.vote(depth, rmax)
            e1.record()
            torch.cuda.synchronize()
            ms_r = e0.elapsed_time(e1) / args.iters
            kp_r, _ = pipe.read_results()
            same = same and bool(np.array_equal(kp_r, kp))
            pipe.vote_engine = engine_auto
        row = {"sigma_dm": sigma, "outlier_frac": outl, "seg_false_pos": fp, "ms_per_%d_items" % (B * G): ms,
               "ms_rasteriser_only": ms_r, "handed_over_at_gate": hand.count(2), "handed_over_at_restart": hand.count(1),
               "lower_bound_over_points_at_gate": float(np.mean(gate_f)), "f_times_rmax_at_gate": float(np.mean(np.array(gate_f) * rmx)),
               "items_decided_by_rasteriser": fallback, "items_with_list_overflow": overflow,
               "mean_points": float(np.mean([r.n_points for r in res])), "mean_dim": float(np.mean([r.dim for r in res])),
               "peak_votes_over_points": float(np.mean([r.peak_votes / max(1, r.n_points) for r in res])),
               "max_keypoint_error_voxels_vs_truth": err, "equal_to_oracle_on_checked_items": same}
        rows.append(row)
        print(json.dumps(row), flush=True)
    benign = rows[0]["ms_per_%d_items" % (B * G)]
    worst = max(rows, key=lambda r: r["ms_per_%d_items" % (B * G)])
    out = {"command": "python tools/vote_sweep.py --frames %d --distinct %d" % (B, args.distinct), "items": B * G,
           "benign_ms": benign, "worst_ms": worst["ms_per_%d_items" % (B * G)],
           "worst_case": {k: worst[k] for k in ("sigma_dm", "outlier_frac", "seg_false_pos")},
           "worst_over_benign": worst["ms_per_%d_items" % (B * G)] / benign,
           "worst_fallback_rate": max(r["items_decided_by_rasteriser"] for r in rows) / float(B * G), "cases": rows}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)
    print("benign %.2f ms, worst %.2f ms (%s)" % (benign, out["worst_ms"], out["worst_case"]))


if __name__ == "__main__":
    main()
