#!/usr/bin/env python3
"""Do the three keypoint networks run faster as three phase-shifted streams than as the `groups` of one launch sequence?

Why it could: the chip lowers its clock under MFMA load (MI355X_MICROARCH.md, DVFS give-back), a forward alternates between
matrix-bound phases (the decoder's 3x3 convolutions, ~40 ms of continuous k_conv_halo) and HBM- / LDS-bound ones (layers 1-2, the
1x1 chains), and power is convex in the clock -- three networks a third of a forward apart would draw an even power and could hold
a higher clock in their matrix-bound launches.  Why it could not: each launch then has a third of the workgroups (tails, fewer
rounds), and kernels of different streams share CUs only where registers and LDS allow (k_conv_halo fills a CU).

    python tools/probe/stream_split_probe.py [--batch 53] [--iters 6]
prints ms per (3 networks x batch) forward for: one engine with groups = 3; three engines on three streams started together;
three engines on three streams a third of a forward apart."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from rcvpose_amd import synthetic_weights
from rcvpose_amd.fcn_engine import FcnEngine

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=53)
ap.add_argument("--iters", type=int, default=6)
ap.add_argument("--dtype", default="f16x3")
args = ap.parse_args()
H, W, B = 480, 640, args.batch
shapes = json.load(open(os.path.join(ROOT, "tests", "golden", "fcn_state_dict_keys.json")))
sds = [{k: torch.from_numpy(v) for k, v in synthetic_weights.make_state_dict_numpy(shapes, seed=s).items()} for s in range(3)]
x = torch.from_numpy(np.random.default_rng(0).standard_normal((B, 3, H, W)).astype(np.float32)).cuda()


def timed(fn, streams):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for s in streams:
        s.wait_event(e0)
    fn()
    cur = torch.cuda.current_stream()
    for s in streams:
        cur.wait_stream(s)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


one = FcnEngine(sds, H, W, dtype=args.dtype)
outs = (torch.empty((3, B, 1, H, W), device="cuda"), torch.empty((3, B, 1, H, W), device="cuda"))
for _ in range(2):
    one.forward(x, out=outs)
t_one = timed(lambda: [one.forward(x, out=outs) for _ in range(args.iters)], []) / args.iters
print("one engine, groups = 3: %.2f ms per forward of %d frames x 3 networks" % (t_one, B))
ref = [o.clone() for o in outs]
del one
torch.cuda.empty_cache()

engs = [FcnEngine([sd], H, W, dtype=args.dtype) for sd in sds]
streams = [torch.cuda.Stream() for _ in range(3)]
o3 = [(torch.empty((1, B, 1, H, W), device="cuda"), torch.empty((1, B, 1, H, W), device="cuda")) for _ in range(3)]
xs = {n: x[:n].contiguous() for n in (max(1, B // 3), max(1, 2 * B // 3))}
od = {n: (torch.empty((1, n, 1, H, W), device="cuda"), torch.empty((1, n, 1, H, W), device="cuda")) for n in xs}
for k in range(3):
    with torch.cuda.stream(streams[k]):
        engs[k].forward(x, out=o3[k])
        for n in xs:
            engs[k].forward(xs[n], out=od[n])
        engs[k].forward(x, out=o3[k])
torch.cuda.synchronize()
for k in range(3):
    for a, b in zip(o3[k], ref):
        assert torch.equal(a[0], b[k]), "a network's maps depend on the group slot it travels in?"


def run(shift):
    def go():
        for k in range(3):
            with torch.cuda.stream(streams[k]):
                if shift and k:  # a shorter forward first: stream k then runs ~k/3 of a forward behind stream 0
                    n = sorted(xs)[k - 1]
                    engs[k].forward(xs[n], out=od[n])
        for _ in range(args.iters):
            for k in range(3):
                with torch.cuda.stream(streams[k]):
                    engs[k].forward(x, out=o3[k])
    return go


t0 = timed(run(False), streams) / args.iters
print("three engines, three streams, started together: %.2f ms per 3 forwards" % t0)
t_shift = timed(run(True), streams)
extra = (1.0 / 3 + 2.0 / 3) * t_one / 3  # the two delay forwards: B/3 and 2B/3 frames of ONE network
print("three engines, three streams, a third of a forward apart: %.2f ms per 3 forwards (total %.1f ms incl. the two delay forwards "
      "of about %.1f ms)" % ((t_shift - extra) / args.iters, t_shift, extra))
# sequential single-network forwards on one stream (what a third of the workgroups per launch costs by itself)
t_seq = timed(lambda: [engs[k].forward(x, out=o3[k]) for _ in range(args.iters) for k in range(3)], []) / args.iters
print("three engines, one stream, one after the other: %.2f ms per 3 forwards" % t_seq)
