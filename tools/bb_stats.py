#!/usr/bin/env python3
"""Candidate-list sizes of the branch-and-bound peak search for the bench workload (debug/tuning aid).
    python tools/bb_stats.py [sigma_dm] [outlier_frac] [seg_false_pos]"""
import json, os, sys
ARGS = [float(a) for a in sys.argv[1:4]] + [0.02, 0.0, 0.0][len(sys.argv[1:4]):]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rcvpose_amd import synthetic, synthetic_weights
from rcvpose_amd.pipeline import FramePipeline

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BOX_OFF = 352  # offsetof(AccHeader, box)
B = 2
shapes = json.load(open(os.path.join(ROOT, "tests", "golden", "fcn_state_dict_keys.json")))
sds = [{k: torch.from_numpy(v) for k, v in synthetic_weights.make_state_dict_numpy(shapes, seed=s).items()} for s in range(3)]
pipe = FramePipeline(sds, 480, 640, dtype="bf16", frames_per_step=B)
fr = [synthetic.synthetic_frame(i, noise_dm=ARGS[0], outlier_frac=ARGS[1], seg_false_pos=ARGS[2]) for i in range(B)]
rmax = np.max([f["max_radii_dm"] for f in fr], axis=0) + 0.05
pipe.seg.copy_(torch.from_numpy(np.stack([np.stack([f["seg"] for f in fr])] * 3))[:, :, None].cuda())
pipe.radial.copy_(torch.from_numpy(np.stack([np.stack([f["radial"][k] for f in fr]) for k in range(3)]))[:, :, None].cuda())
depth = torch.from_numpy(np.stack([f["depth"] for f in fr]).view(np.int16)).cuda()
pipe.vote(depth, rmax)
torch.cuda.synchronize()
kp, res = pipe.read_results()
for i in range(B * 3):
    ints = pipe.ws[i, 48:288].cpu().numpy().view(np.int32)
    # AccHeader after the six doubles: n, m, zb, dim, tile geometry (8), wide, nparts, status, tile_counter, bb_enabled, bb_done,
    # bb_best, bb_overflow, bb_count[8] (pair regions), bs_count[8] (single regions), pb_count[8], sel_count[8]
    hb = pipe.ws[i, :512].cpu().numpy()
    import struct
    # AccHeader::box sits at the end of the header: find it by layout (after `restart`): offsets computed from the struct in accumulator.hip
    box = np.frombuffer(hb[BOX_OFF:BOX_OFF + 48].tobytes(), dtype=np.int32).reshape(2, 6)
    dims = [(int(b[1] - b[0] + 1), int(b[3] - b[2] + 1), int(b[5] - b[4] + 1)) for b in box]
    print("   candidate boxes (i, j, k extent): edge 8 %s, edge 4 %s" % (dims[0], dims[1]))
    print("item %2d (frame %d kp %d) n %d dim %d peak %d | bb done %d best %d | pair entries %s | single entries %s | probe %s" %
          (i, i // 3, i % 3, ints[0], ints[3], res[i].peak_votes, ints[17], ints[18], [int(v) for v in ints[20:27]],
           [int(v) for v in ints[28:35]], [int(v) for v in ints[36:42]]))
