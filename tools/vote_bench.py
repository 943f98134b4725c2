#!/usr/bin/env python3
"""Micro-benchmark of the voting path alone (seam -> prepare -> vote -> arg-max) on one synthetic frame.

    python tools/vote_bench.py [--iters 20] [--kp 0]
Prints per-item time measured with HIP events on the launch stream.  Used with rocprofv3 for PMC runs.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from rcvpose_amd import AccumulatorSpace as A
from rcvpose_amd import synthetic


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--kp", type=int, default=0)
    ap.add_argument("--shape", default="lm")
    ap.add_argument("--unit", type=float, default=5.0)
    ap.add_argument("--mode", default="auto", choices=["auto", "full"])
    args = ap.parse_args()
    f = synthetic.synthetic_frame(0, shape=args.shape)
    seg = torch.from_numpy(f["seg"]).cuda()
    rad = torch.from_numpy(f["radial"][args.kp]).cuda()
    depth = torch.from_numpy(f["depth"].view(np.int16)).cuda()
    K = np.array([[f["K"][0], 0, f["K"][2]], [0, f["K"][1], f["K"][3]], [0, 0, 1]])
    ws = A.AccWorkspace(480 * 640)
    rmax = float(f["max_radii_dm"][args.kp])

    def item():
        A.seam_compact(seg, rad, depth, rmax, K=K, workspace=ws)
        A.accumulate_on_device(ws.xyz, ws.radial, ws.counts, ws.max_points, ws.max_points, ws, acc_unit=args.unit, engine=args.mode)

    for _ in range(3):
        item()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        item()
    e1.record()
    torch.cuda.synchronize()
    res = ws.read_result()
    print("n_points %d dim %d peak %d status %d : %.3f ms per (frame, keypoint) item" %
          (res.n_points, res.dim, res.peak_votes, res.status, e0.elapsed_time(e1) / args.iters))


if __name__ == "__main__":
    main()
