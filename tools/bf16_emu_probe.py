#!/usr/bin/env python3
"""Per-stage difference of the bf16 engine and the bf16-emulating oracle (diagnostic for tests/test_gpu_fcn.py)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from oracle import fcn_torch
from rcvpose_amd import synthetic_weights
from rcvpose_amd.fcn_engine import FcnEngine

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (480, 640)
shapes = json.load(open(os.path.join(ROOT, "tests", "golden", "fcn_state_dict_keys.json")))
sd = {k: torch.from_numpy(v) for k, v in synthetic_weights.make_state_dict_numpy(shapes, seed=0).items()}
x = torch.from_numpy(synthetic_weights.make_input(11, (1, 3, H, W)))
s_ref, r_ref, st = fcn_torch.forward_bf16(sd, x, return_stages=True)
s32, r32, st32 = fcn_torch.forward(sd, x, return_stages=True)
eng = FcnEngine([sd], H, W, dtype="bf16", reuse_slots=False)
eng.set_option(3, 0)
for opts in ((), ((4, 0),), ((4, 0), (1, 0), (7, 0))):
    for k, v in opts:
        eng.set_option(k, v)
    seg, rad = eng.forward(x.cuda())
    print("options", opts)
    for name in eng.stages:
        got = eng.debug_stage(1, 0, name).cpu()
        ref = st[name]
        print("  %-9s vs emu: max rel %.3e mismatch %.4f | emu vs fp32 %.3e | engine vs fp32 %.3e" % (
            name, float((got - ref).abs().max() / ref.abs().max()), float((got != ref).float().mean()),
            float((ref - st32[name]).abs().max() / st32[name].abs().max()), float((got - st32[name]).abs().max() / st32[name].abs().max())))
    print("  radial vs emu %.3e  | emu vs fp32 %.3e | engine vs fp32 %.3e" % (
        float((rad[0].cpu() - r_ref).abs().max() / r_ref.abs().max()), float((r_ref - r32).abs().max() / r32.abs().max()),
        float((rad[0].cpu() - r32).abs().max() / r32.abs().max())))
