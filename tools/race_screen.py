#!/usr/bin/env python3
"""Race screen for the fused / staggered convolution kernels: repeat the same forward many times and demand bit-identical
outputs every time, and equality with the unfused kernels (options 1, 3, 4, 5 off).   python tools/race_screen.py [--reps 30]
--dtype f16x3: the split-precision engine's chain kernels (k_chain2_x3, counted vmcnt waits) against k_chain_x3 (option 10 = 0), exact tail
(option 13 = 0: the bit-identical family), at a batch that fills the chip:   python tools/race_screen.py --dtype f16x3 --batch 53 --reps 20"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rcvpose_amd import synthetic_weights
from rcvpose_amd.fcn_engine import FcnEngine

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--batch", type=int, default=9)
ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16x3"])
args = ap.parse_args()
shapes = json.load(open(os.path.join(ROOT, "tests", "golden", "fcn_state_dict_keys.json")))
sds = [{k: torch.from_numpy(v) for k, v in synthetic_weights.make_state_dict_numpy(shapes, seed=s).items()} for s in range(3)]
eng = FcnEngine(sds, 480, 640, dtype=args.dtype)
g = torch.Generator().manual_seed(5)
x = torch.randn((args.batch, 3, 480, 640), generator=g).cuda()
OFF_ON = [(k, 0, 1) for k in (1, 3, 4, 5, 6, 7)] if args.dtype == "bf16" else [(10, 0, 1), (13, 0, 0)]
for key, off, _on in OFF_ON:
    eng.set_option(key, off)
ref = [t.clone() for t in eng.forward(x)]
for key, _off, on in OFF_ON:
    eng.set_option(key, on)
bad = 0
for r in range(args.reps):
    seg, rad = eng.forward(x)
    if not (torch.equal(seg, ref[0]) and torch.equal(rad, ref[1])):
        bad += 1
        print("rep %d differs: max |d radial| %.3e" % (r, (rad - ref[1]).abs().max().item()))
print("%d / %d repetitions identical to %s" % (args.reps - bad, args.reps, "the unfused register-staged kernels" if args.dtype == "bf16" else "the forward with k_chain_x3 (option 10 = 0)"))
sys.exit(1 if bad else 0)
