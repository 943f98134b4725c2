#!/usr/bin/env python3
"""Per-op A-B of engine options on one box: runs tools/fcn_bench.py --all under several option sets and lists the ops whose time
differs by more than 5 % from the default.   python tools/op_ab.py [--batch 53] [--dtype f16x3] "1=0" "7=2" ..."""
import re
import subprocess
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
batch, dtype, sets = "53", "f16x3", []
while args:
    a = args.pop(0)
    if a == "--batch":
        batch = args.pop(0)
    elif a == "--dtype":
        dtype = args.pop(0)
    else:
        sets.append(a)


def run(opts):
    cmd = [sys.executable, os.path.join(ROOT, "tools", "fcn_bench.py"), "--dtype", dtype, "--batch", batch, "--iters", "3", "--all"]
    for o in opts.split(",") if opts else []:
        cmd += ["--opt", o]
    out = subprocess.run(cmd, capture_output=True, text=True).stdout
    rows = []
    for line in out.splitlines():
        m = re.match(r"(\w+)\s+(\d+x\d+)\s+cin\s+(\d+) cout\s+(\d+) k(\d) s(\d)\s+([\d.]+) ms", line)
        if m:
            rows.append((m.group(1), m.group(2), int(m.group(3)), int(m.group(4)), int(m.group(5)), int(m.group(6)), float(m.group(7))))
    tot = re.search(r"conv total ([\d.]+) ms", out)
    return rows, float(tot.group(1)) if tot else None


base, tb = run("")
print("default: conv total %.3f ms" % tb)
for s in sets:
    rows, t = run(s)
    print("== options %s: conv total %.3f ms" % (s, t))
    for b, r in zip(base, rows):
        if b[6] > 0.05 and abs(r[6] - b[6]) / b[6] > 0.05:
            print("   %-8s %-8s cin %5d cout %5d k%d s%d   %.3f -> %.3f ms" % (b[:6] + (b[6], r[6])))
