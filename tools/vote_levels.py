#!/usr/bin/env python3
"""Per-level candidate counts of the branch-and-bound peak search (AccHeader::bb_count / bs_count) and the decision fields, for a few
workloads: the bench cloud at several noise levels and BASELINE configs[4] (YCB-shaped, 1 mm voxels, float16 radial maps).
    python tools/vote_levels.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from rcvpose_amd import synthetic, synthetic_weights
from rcvpose_amd.pipeline import FramePipeline

shapes = json.load(open(os.path.join(ROOT, "tests", "golden", "fcn_state_dict_keys.json")))
sds = [{k: torch.from_numpy(v) for k, v in synthetic_weights.make_state_dict_numpy(shapes, seed=s).items()} for s in range(3)]
B, G, H, W = 4, 3, 480, 640


def run(name, fr, **kw):
    stress = "acc_unit" in kw
    pipe = FramePipeline(sds, H, W, dtype="bf16", frames_per_step=B, **kw)
    rad = np.stack([np.stack([f["radial"][k] for f in fr]) for k in range(G)])
    if stress:
        rad = rad.astype(np.float16)
        rmax = torch.tensor(rad.astype(np.float32).max(axis=(2, 3)).T.astype(np.float64), device="cuda")
    else:
        rmax = torch.tensor(np.stack([f["max_radii_dm"] for f in fr]), dtype=torch.float64, device="cuda")
    pipe.seg.copy_(torch.from_numpy(np.stack([np.stack([f["seg"] for f in fr])] * G))[:, :, None].cuda())
    pipe.radial.copy_(torch.from_numpy(rad)[:, :, None].cuda())
    depth = torch.from_numpy(np.stack([f["depth"] for f in fr]).view(np.int16)).cuda()
    pipe.vote(depth, rmax)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        pipe.vote(depth, rmax)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    kp, res = pipe.read_results()
    hdr = pipe.ws[:, :512].cpu().numpy()
    print("== %s: %.2f ms per %d items" % (name, ms, B * G))
    for i in range(0, B * G, 4):
        h32 = hdr[i].view(np.int32)
        rmx = hdr[i][24:32].view(np.float64)[0]
        n, dim, done, best, ovf, hand, gate = (int(h32[q // 4]) for q in (48, 60, 116, 120, 124, 324, 268))
        pairs = [int(v) for v in h32[128 // 4:128 // 4 + 7]]
        singles = [int(v) for v in h32[160 // 4:160 // 4 + 7]]
        tests = sum((p + s) for p, s in zip(pairs[:6], singles[:6])) * 8.0 * n
        print("  item %2d: n %d dim %d rmax %.0f | peak %d (f %.3f, f*rmax %.1f; at gate %.3f) done %d overflow %d handover %d | pair entries %s | singles %s | ~%.2f G block-point tests"
              % (i, n, dim, rmx, res[i].peak_votes, res[i].peak_votes / max(n, 1), res[i].peak_votes / max(n, 1) * rmx, gate / max(n, 1), done, ovf, hand, pairs, singles, tests / 1e9))
    del pipe
    torch.cuda.empty_cache()


for sigma in (0.02, 0.05, 0.1):
    run("bench cloud, sigma %.2f dm" % sigma, [synthetic.synthetic_frame(i, noise_dm=sigma) for i in range(B)])
ycb = [synthetic.synthetic_frame(300 + i, shape="ycb") for i in range(B)]
fx, fy, cx, cy = ycb[0]["K"]
run("configs[4]: ycb, 1 mm, fp16 radial", ycb, K=np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]]), acc_unit=1.0, radial_f16=True)
run("ycb at 5 mm", ycb, K=np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]]))
