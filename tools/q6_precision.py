#!/usr/bin/env python3
"""Map error of the split engine against the fp32 engine for several choices of the layers that run their cross terms in fp6
(engine option 11; FcnEngine(q6=<regex over conv names>)), on frames of the bench and the bench's three networks.
    python tools/q6_precision.py [--frames 4] [pattern ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from rcvpose_amd import synthetic, synthetic_weights
from rcvpose_amd.fcn_engine import FcnEngine, preprocess_rgb

args = sys.argv[1:]
nf = 4
if "--frames" in args:
    i = args.index("--frames")
    nf = int(args[i + 1])
    del args[i:i + 2]
pats = args or ["", r"block3\.\d+\.conv2", r"block3\.(\d|1[0-7])\.conv2", r"block3\.(1[8-9]|2\d|3\d)\.conv2", r"block[234]\.\d+\.conv2", r"block[24]\.\d+\.conv2"]
shapes = json.load(open(os.path.join(ROOT, "tests", "golden", "fcn_state_dict_keys.json")))
sds = [{k: torch.from_numpy(v) for k, v in synthetic_weights.make_state_dict_numpy(shapes, seed=s).items()} for s in range(3)]
frames = [synthetic.synthetic_frame(i) for i in range(nf)]
x = preprocess_rgb(torch.from_numpy(np.stack([f["rgb"] for f in frames])).cuda())
ref = FcnEngine(sds, 480, 640, dtype="fp32")
seg_r, rad_r = [t.clone() for t in ref.forward(x)]
del ref
torch.cuda.empty_cache()
for pat in pats:
    eng = FcnEngine(sds, 480, 640, dtype="f16x3", q6=pat if pat else False)
    seg, rad = eng.forward(x)
    n = sum(1 for o in eng.ops if o["wq_off"])
    er = [[float((rad[g, f] - rad_r[g, f]).abs().max() / rad_r[g, f].abs().max()) for f in range(nf)] for g in range(3)]
    es = [[float((seg[g, f] - seg_r[g, f]).abs().max() / seg_r[g, f].abs().max()) for f in range(nf)] for g in range(3)]
    print("%-34s %2d ops: radial worst %.2e (per network %s)  seg worst %.2e (per network %s)" % (
        pat or "(none)", n, max(map(max, er)), " ".join("%.1e" % max(e) for e in er), max(map(max, es)), " ".join("%.1e" % max(e) for e in es)), flush=True)
    del eng
    torch.cuda.empty_cache()
