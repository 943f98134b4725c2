#!/usr/bin/env python3
"""Per-layer timing of the FCN engine (HIP events around every op, on the launch stream).

    python tools/fcn_bench.py [--batch 4] [--groups 3] [--dtype bf16] [--iters 5]
"""
import argparse
import collections
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from rcvpose_amd import synthetic_weights
from rcvpose_amd.fcn_engine import FcnEngine

KIND = {1: "stem", 2: "maxpool", 3: "conv", 4: "upsample", 5: "head"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--groups", type=int, default=3)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--no-dma", action="store_true", help="disable the LDS-DMA conv kernel")
    ap.add_argument("--off", type=int, action="append", default=[], help="engine option key to switch off (repeatable)")
    ap.add_argument("--opt", action="append", default=[], help="engine option as key=value (repeatable)")
    ap.add_argument("--q6", action="store_true", help="f16x3: build the engine with fp6 cross terms in the stride-1 3x3 convolutions of layers 2-4 (A-B: --opt 11=0)")
    ap.add_argument("--graph", action="store_true", help="also time the forward replayed as one HIP graph")
    ap.add_argument("--all", action="store_true", help="print every op, not the per-shape summary")
    ap.add_argument("--ignore-range", action="store_true", help="clear the engine's range flag after every forward (diagnostic builds that compute garbage on purpose)")
    args = ap.parse_args()
    shapes = json.load(open(os.path.join(ROOT, "tests", "golden", "fcn_state_dict_keys.json")))
    sds = [{k: torch.from_numpy(v) for k, v in synthetic_weights.make_state_dict_numpy(shapes, seed=s).items()}
           for s in range(args.groups)]
    eng = FcnEngine(sds, 480, 640, dtype=args.dtype, q6=args.q6)
    if args.no_dma:
        eng.set_option(1, 0)
    for key in args.off:
        eng.set_option(key, 0)
    for kv in args.opt:
        eng.set_option(int(kv.split("=")[0]), int(kv.split("=")[1]))
    x = torch.randn(args.batch, 3, 480, 640, device="cuda")
    for _ in range(2):
        eng.forward(x)
        if args.ignore_range:
            eng.range_status(clear=True)
    eng.set_profiling(True)
    acc = np.zeros(len(eng.ops))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(args.iters):
        eng.forward(x)
        acc += np.array(eng.get_op_times())
        if args.ignore_range:
            eng.range_status(clear=True)
    e1.record()
    torch.cuda.synchronize()
    acc /= args.iters
    BG = args.batch * args.groups
    by = collections.OrderedDict()
    tot_flop = 0.0
    for o, ms in zip(eng.ops, acc):
        cin = o["c0"] + (o["c1"] if o["src1"] >= 0 else 0)
        fl = 2.0 * o["hout"] * o["wout"] * o["cout"] * o["ksize"] ** 2 * cin * BG if o["kind"] in (1, 3, 5) else 0.0
        key = (KIND[o["kind"]], o["hout"], o["wout"], cin, o["cout"], o["ksize"], o["stride"])
        d = by.setdefault(key, [0, 0.0, 0.0])
        d[0] += 1
        d[1] += ms
        d[2] += fl
        tot_flop += fl if o["kind"] == 3 else 0
        if args.all:
            print("%-8s %4dx%-4d cin %5d cout %5d k%d s%d  %8.3f ms  %7.1f TF" % (*key, ms, fl / ms / 1e9 if ms else 0))
    print("%-8s %9s %6s %6s k s   n    ms_total   TFLOP/s   M=B*H*W" % ("kind", "HxW", "cin", "cout"))
    for key, (n, ms, fl) in by.items():
        print("%-8s %4dx%-4d %6d %6d %d %d %3d %10.3f %9.1f   %d" % (key[0], key[1], key[2], key[3], key[4], key[5], key[6], n, ms,
                                                                 fl / ms / 1e9 if ms else 0, args.batch * key[1] * key[2]))
    conv_ms = sum(ms for o, ms in zip(eng.ops, acc) if o["kind"] == 3)
    print("conv total %.3f ms  -> %.1f TFLOP/s ; all ops %.3f ms ; wall per forward %.3f ms (B=%d, G=%d, %s)" %
          (conv_ms, tot_flop / conv_ms / 1e9, acc.sum(), e0.elapsed_time(e1) / args.iters, args.batch, args.groups, args.dtype))
    if args.graph:
        eng.set_profiling(False)
        seg_e, rad_e = [t.clone() for t in eng.forward(x)]
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0.record()
        for _ in range(args.iters):
            eng.forward(x)
        t1.record()
        torch.cuda.synchronize()
        eager = t0.elapsed_time(t1) / args.iters
        graph, seg_g, rad_g = eng.capture(x)
        torch.cuda.synchronize()
        t0.record()
        for _ in range(args.iters):
            graph.replay()
        t1.record()
        torch.cuda.synchronize()
        print("wall per forward without per-op events: %.3f ms launched op by op, %.3f ms replayed as one HIP graph; identical outputs: %s" %
              (eager, t0.elapsed_time(t1) / args.iters, bool(torch.equal(seg_e, seg_g) and torch.equal(rad_e, rad_g))))


if __name__ == "__main__":
    main()
