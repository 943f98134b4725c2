#!/usr/bin/env python3
"""Diagnostic: where one K step of the split-precision k_conv_halo<128,4> spends its cycles (s_memtime stamps of waves 0 and 4 of
one workgroup; a separate build: `tools/ab_build.sh -DRCV_HALO_STAMPS`, then
    RCV_LIB_PATH=$PWD/tools/probe/ab/librcvpose_hip_a.so python tools/halo_stamps.py [--batch 36]
The stamps of the LAST <128,4> launch of the forward survive (conv_up2, 512 -> 128 at 120x160).  Read shares, not totals."""
import argparse, ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from rcvpose_amd import _hip, synthetic_weights
from rcvpose_amd.fcn_engine import FcnEngine

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=36)
ap.add_argument("--tail", action="store_true", help="the build was made with -DRCV_HALO_STAMPS -DRCV_TAIL_STAMPS: phases of one tile of k_tail_x3")
ap.add_argument("--chain2", action="store_true", help="the build was made with -DRCV_HALO_STAMPS -DRCV_CHAIN2_STAMPS: stamps of k_chain2_x3<256> (half chunks 6 and 7)")
ap.add_argument("--chain", action="store_true", help="the build was made with -DRCV_HALO_STAMPS -DRCV_CHAIN_STAMPS: stamps of k_chain_x3<256> (chunk 3: expand step 3, reduce step 1)")
ap.add_argument("--ignore-range", action="store_true", help="diagnostic builds that compute garbage on purpose")
args = ap.parse_args()
shapes = json.load(open(os.path.join(ROOT, "tests", "golden", "fcn_state_dict_keys.json")))
sds = [{k: torch.from_numpy(v) for k, v in synthetic_weights.make_state_dict_numpy(shapes, seed=s).items()} for s in range(3)]
eng = FcnEngine(sds, 480, 640, dtype="f16x3")
x = torch.randn(args.batch, 3, 480, 640, device="cuda")
for _ in range(3):
    try:
        eng.forward(x)
    except _hip.HalfRangeError:
        if not args.ignore_range:
            raise
        eng.range_status(clear=True)
torch.cuda.synchronize()
L = ctypes.CDLL(_hip.LIB_PATH)
if args.tail:
    out = (ctypes.c_uint * 128)()
    assert L.rcv_debug_chain_stamps(out) == 0
    st = np.array(out[:], dtype=np.int64).reshape(2, 64)
    names = ["wait for the source patch (vmcnt)", "barrier", "build U (upsample: join, blend, split)", "barrier", "conv7 MFMAs (+ next patch's DMA issue)",
             "barrier", "conv7 epilogue", "barrier", "head + stores"]
    for w in range(2):
        s = st[w]
        print("wave %d: tile = %d cycles" % (0 if w == 0 else 3, s[9] - s[0]))
        for k, nm in enumerate(names):
            print("   %-46s %6d" % (nm, s[k + 1] - s[k]))
    sys.exit(0)
if args.chain2:
    out = (ctypes.c_uint * 128)()
    assert L.rcv_debug_chain_stamps(out) == 0
    st = np.array(out[:], dtype=np.int64).reshape(2, 64)
    for w in range(2):
        s = st[w]
        print("wave %d: half chunks 6 + 7 = %d cycles; per step (E = expand step of two K slices, R = reduce step): counted vmcnt wait / barrier / issue / compute" % (4 * w, s[48] - s[0]))
        for q in range(12):
            print("   %s%d  %5d %5d %5d %5d   = %5d" % ("E" if q % 6 < 4 else "R", q % 6 if q % 6 < 4 else q % 6 - 4, s[4 * q + 1] - s[4 * q], s[4 * q + 2] - s[4 * q + 1],
                                                   s[4 * q + 3] - s[4 * q + 2], s[4 * q + 4] - s[4 * q + 3], s[4 * q + 4] - s[4 * q]))
    sys.exit(0)
if args.chain:
    out = (ctypes.c_uint * 128)()
    assert L.rcv_debug_chain_stamps(out) == 0
    st = np.array(out[:], dtype=np.int64).reshape(2, 64)
    for w in range(2):
        s = st[w]
        print("wave %d: chunk 3 = %d cycles; per step (E = expand slice, R = reduce slice): counted vmcnt wait / barrier / DMA issue / compute" % (4 * w, s[48] - s[0]))
        for q in range(12):
            e = s[4 * q + 4] if q < 11 else s[48]
            print("   %s%d  %5d %5d %5d %5d   = %5d" % ("E" if q < 8 else "R", q if q < 8 else q - 8, s[4 * q + 1] - s[4 * q], s[4 * q + 2] - s[4 * q + 1],
                                                   s[4 * q + 3] - s[4 * q + 2], e - s[4 * q + 3], e - s[4 * q]))
    sys.exit(0)
out = (ctypes.c_uint * 32)()
assert L.rcv_debug_halo_stamps(out) == 0
st = np.array(out[:], dtype=np.int64).reshape(2, 16)
names = ["L1 (10 reads + 1 DMA + lgkm wait)", "barrier 1", "M1 (24 MFMAs issued)", "barrier 2", "L2 (6 reads + DMAs + vmcnt/lgkm wait)", "barrier 3",
         "M2 (24 MFMAs issued)", "barrier 4"]
for w in range(2):
    s = st[w]
    print("wave %d (group %d): step = %d cycles of the s_memtime clock (100 MHz x ... see guide; shares matter); starts %d after wave 0's" %
          (4 * w, w, s[8] - s[0], s[0] - st[0][0]))
    for k, nm in enumerate(names):
        print("  %-42s %6d" % (nm, s[k + 1] - s[k]))
