#!/usr/bin/env python3
"""The voting stage alone on a batch of synthetic frames (seam + Accumulator_3D for frames x 3 keypoints in one batched launch
sequence), as bench.py runs it -- used under rocprofv3 for the PMC passes of the k_bb_* kernels.
    python tools/vote_batch_bench.py [--frames 72] [--iters 3] [--engine auto|full]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from rcvpose_amd import synthetic, synthetic_weights
from rcvpose_amd.pipeline import FramePipeline

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=72)
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--engine", default="auto", help="'full': the LDS rasteriser decides every item")
args = ap.parse_args()
B = args.frames
shapes = json.load(open(os.path.join(ROOT, "tests", "golden", "fcn_state_dict_keys.json")))
sds = [{k: torch.from_numpy(v) for k, v in synthetic_weights.make_state_dict_numpy(shapes, seed=s).items()} for s in range(3)]
pipe = FramePipeline(sds, 480, 640, dtype="bf16", frames_per_step=B, engine=args.engine)
fr = [synthetic.synthetic_frame(i) for i in range(min(B, 12))]
fr = [fr[i % len(fr)] for i in range(B)]
rmax = torch.tensor(np.stack([f["max_radii_dm"] for f in fr]), dtype=torch.float64, device="cuda")
pipe.seg.copy_(torch.from_numpy(np.stack([np.stack([f["seg"] for f in fr])] * 3))[:, :, None].cuda())
pipe.radial.copy_(torch.from_numpy(np.stack([np.stack([f["radial"][k] for f in fr]) for k in range(3)]))[:, :, None].cuda())
depth = torch.from_numpy(np.stack([f["depth"] for f in fr]).view(np.int16)).cuda()
pipe.vote(depth, rmax)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(args.iters):
    pipe.vote(depth, rmax)
e1.record()
torch.cuda.synchronize()
kp, res = pipe.read_results()
assert all(r.status == 0 for r in res)
print("seam + voting: %.3f ms per batch of %d frames x 3 keypoints (%.1f us per item)" % (
    e0.elapsed_time(e1) / args.iters, B, e0.elapsed_time(e1) / args.iters / (3 * B) * 1e3))
