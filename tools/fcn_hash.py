#!/usr/bin/env python3
"""Print a checksum of the FCN outputs for fixed synthetic weights/inputs (A/B check of kernel changes: equal checksums
before and after mean bit-identical seg / radial maps).   python tools/fcn_hash.py [--dtype bf16|fp32]"""
import argparse, hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rcvpose_amd import synthetic_weights
from rcvpose_amd.fcn_engine import FcnEngine

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--opt", action="append", default=[], help="engine option as key=value (repeatable)")
ap.add_argument("--batch", type=int, default=0, help="frames of the 480x640 case (default 2)")
args = ap.parse_args()
shapes = json.load(open(os.path.join(ROOT, "tests", "golden", "fcn_state_dict_keys.json")))
sds = [{k: torch.from_numpy(v) for k, v in synthetic_weights.make_state_dict_numpy(shapes, seed=s).items()} for s in range(2)]
for (H, W, B) in ((480, 640, args.batch or 2), (96, 160, 3)):
    eng = FcnEngine(sds, H, W, dtype=args.dtype)
    for kv in args.opt:
        eng.set_option(int(kv.split("=")[0]), int(kv.split("=")[1]))
    g = torch.Generator().manual_seed(1)
    x = torch.randn((B, 3, H, W), generator=g).cuda()
    seg, rad = eng.forward(x)
    h = hashlib.sha256(seg.cpu().numpy().tobytes() + rad.cpu().numpy().tobytes()).hexdigest()[:16]
    print("%s %dx%d B=%d sha256 %s" % (args.dtype, H, W, B, h))
