#!/usr/bin/env python3
"""Where does the forward lose time when the voting of the previous batch runs beside it?  Per-op HIP-event times of the forward with
and without the voting stream (FramePipeline(overlap=...)): the voting kernels are VALU-bound, the forward alternates between
MFMA-bound kernels that fill a CU (k_conv_halo: 160 KB of LDS, 8 x 256 registers) and HBM-bound ones that do not (chains of layers
1-2, stem, pool, upsampling).
    python tools/overlap_ops.py [--frames 53] [--steps 4]"""
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
ap.add_argument("--frames", type=int, default=53)
ap.add_argument("--steps", type=int, default=4)
args = ap.parse_args()
H, W, B, G = 480, 640, args.frames, 3
shapes = json.load(open(os.path.join(ROOT, "tests", "golden", "fcn_state_dict_keys.json")))
sds = [{k: torch.from_numpy(v) for k, v in synthetic_weights.make_state_dict_numpy(shapes, seed=s).items()} for s in range(3)]
fr = [synthetic.synthetic_frame(i) for i in range(B)]
rgb = torch.from_numpy(np.stack([f["rgb"] for f in fr])).cuda()
depth = torch.from_numpy(np.stack([f["depth"] for f in fr]).view(np.int16)).cuda()
seg = torch.from_numpy(np.stack([f["seg"] for f in fr])).cuda()[None].expand(G, B, H, W)[:, :, None].contiguous()
rad = torch.from_numpy(np.stack([f["radial"] for f in fr])).cuda().permute(1, 0, 2, 3)[:, :, None].contiguous()
rmax = torch.tensor(np.stack([f["max_radii_dm"] for f in fr]), dtype=torch.float64, device="cuda")
res = {}
for overlap in (False, True):
    pipe = FramePipeline(sds, H, W, frames_per_step=B, overlap=overlap)
    for _ in range(2):
        pipe.enqueue_step(rgb, depth, rmax, inject=(seg, rad))
    pipe.finish()
    pipe.engine.set_profiling(True)
    pipe.set_stage_timing(True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.steps):
        pipe.enqueue_step(rgb, depth, rmax, inject=(seg, rad))
    pipe.finish()
    e1.record()
    torch.cuda.synchronize()
    wall = e0.elapsed_time(e1) / args.steps
    ops, t = pipe.engine.ops, pipe.engine.get_op_times()
    st = pipe.stage_times()
    res[overlap] = (wall, ops, t, {k: v / args.steps for k, v in st.items()})
    del pipe
    torch.cuda.empty_cache()
KIND = {0: "stem", 1: "maxpool", 2: "upsample", 3: "conv", 4: "head"}
print("ms per step: no overlap %.2f, overlap %.2f; stages %s | %s" % (res[False][0], res[True][0], res[False][3], res[True][3]))
rows = {}
for (o, a, b) in zip(res[False][1], res[False][2], res[True][2]):
    key = (KIND.get(o["kind"], str(o["kind"])), o["hin"], o["c0"] + max(o["c1"], 0), o["cout"], o["ksize"], o["stride"])
    r = rows.setdefault(key, [0, 0.0, 0.0])
    r[0] += 1
    r[1] += a
    r[2] += b
print("%-9s %5s %6s %6s k s   n   alone   beside-voting   (ms of the LAST forward)" % ("kind", "hin", "cin", "cout"))
for k, (n, a, b) in rows.items():
    if max(a, b) > 0.05:
        print("%-9s %5d %6d %6d %d %d %3d %8.3f %8.3f  %+6.2f" % (k[0], k[1], k[2], k[3], k[4], k[5], n, a, b, b - a))
print("sum %.2f -> %.2f" % (sum(res[False][2]), sum(res[True][2])))
