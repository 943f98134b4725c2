#!/usr/bin/env python3
"""Diagnostic: where one chunk of k_conv_block<256, chain, staggered> spends its cycles.

    python tools/blk_stamps.py --build      (CPU container: compiles fcn.hip with -DRCV_BLK_STAMPS into rcvpose_amd/build/)
    python tools/blk_stamps.py [--batch 16] (GPU box: runs a forward on the stamped library and prints the segment table)

The stamped library is a separate build; the product library never executes a stamp.  Read shares, not totals."""
import argparse, ctypes, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rcvpose_amd import build as B

LIB = os.path.join(B.HERE, "build", "librcvpose_stamps.so")
ap = argparse.ArgumentParser()
ap.add_argument("--build", action="store_true")
ap.add_argument("--define", action="append", default=[], help="extra -D for the diagnostic build (timing experiments)")
ap.add_argument("--batch", type=int, default=16)
args = ap.parse_args()
if args.build:
    B.build_library()
    obj = os.path.join(B.HERE, "build", "fcn_stamps.o")
    subprocess.check_call([B._hipcc()] + B.COMMON + ["-DRCV_BLK_STAMPS"] + ["-D" + d for d in args.define] + ["-c", os.path.join(B.CSRC, "fcn.hip"), "-o", obj])
    objs = [os.path.join(B.HERE, "build", s.replace(".hip", ".o")) for s in B.SOURCES if s != "fcn.hip"] + [obj]
    subprocess.check_call([B._hipcc(), "--offload-arch=" + B.ARCH, "-shared", "-fPIC", "-o", LIB] + objs)
    print("built", LIB)
    sys.exit(0)

import numpy as np
import torch
from rcvpose_amd import _hip
_hip.LIB_PATH = LIB
from rcvpose_amd import synthetic_weights
from rcvpose_amd.fcn_engine import FcnEngine

shapes = json.load(open(os.path.join(ROOT, "tests", "golden", "fcn_state_dict_keys.json")))
sds = [{k: torch.from_numpy(v) for k, v in synthetic_weights.make_state_dict_numpy(shapes, seed=s).items()} for s in range(3)]
eng = FcnEngine(sds, 480, 640, dtype="bf16")
x = torch.randn(args.batch, 3, 480, 640, device="cuda")
for _ in range(3):
    eng.forward(x)
torch.cuda.synchronize()
out = (ctypes.c_uint * 160)()
L = _hip.lib()
L.rcv_debug_blk_stamps.restype = ctypes.c_int
assert L.rcv_debug_blk_stamps(out) == 0
st = np.array(out[:], dtype=np.int64).reshape(2, 80)
names = []
for ks in range(4):
    names += ["Le%d" % ks, "Me%d" % ks]
names.append("P")
for t in range(4):
    names += ["Lr%d" % t, "Mr%d" % t]
print("one chunk (nc = 3) of the last 256-wide chained block launched; cycles of the s_memtime clock; a stamp costs about 40")
for w in range(2):
    s = st[w]
    t0 = s[0]
    print("wave %d (group %d): chunk = %d cycles; starts %d cycles after wave 0's chunk" % (4 * w, w, s[34] - s[0], s[0] - st[0][0]))
    print("  %-4s %8s %8s %8s %8s" % ("seg", "start", "work", "(pre-wait)", "barrier"))
    starts = [s[4 * ks + 2 * h] for ks in range(4) for h in range(2)] + [s[16]] + [s[18 + 4 * t + 2 * h] for t in range(4) for h in range(2)]
    ends = [s[4 * ks + 2 * h + 1] for ks in range(4) for h in range(2)] + [s[17]] + [s[18 + 4 * t + 2 * h + 1] for t in range(4) for h in range(2)]
    nxt = starts[1:] + [s[34]]
    for k, nm in enumerate(names):
        pre = ""
        if nm.startswith("Le"):
            pre = str(s[40 + int(nm[2])] - starts[k])
        print("  %-4s %8d %8d %8s %8d" % (nm, starts[k] - t0, ends[k] - starts[k], pre, nxt[k] - ends[k]))
    print("  inside Le0: W pieces issued +%d, W3 pieces +%d, fragment reads back +%d" % (s[44] - s[0], s[45] - s[44], s[40] - s[45]))
    print("  inside Le1: W pieces issued +%d, W3 pieces +%d, fragment reads back +%d" % (s[46] - s[4], s[47] - s[46], s[41] - s[47]))
