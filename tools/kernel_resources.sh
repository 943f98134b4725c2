#!/bin/bash
# Register / spill / LDS figures of every kernel of one translation unit (default: the split-precision FCN unit):
#   tools/kernel_resources.sh [rcvpose_amd/csrc/fcn_x3.hip] [extra -D flags]
R=$(cd "$(dirname "$0")/.." && pwd)
SRC=${1:-$R/rcvpose_amd/csrc/fcn_x3.hip}
shift
EXTRA=""
case "$SRC" in *accumulator.hip|*pose.hip|*icp.hip) EXTRA="-ffp-contract=off";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -I$R/rcvpose_amd/csrc $EXTRA "$@" -c $SRC -o /tmp/kres_$$.o \
    -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re, sys
cur = None
rows = {}
for line in sys.stdin:
    m = re.search(r"remark: [^ ]+ (?:Function )?Name: (\S+)", line) or re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r"remark: .*?\s+([A-Za-z ]+): (\d+)", line)
    if m and cur: rows[cur][m.group(1).strip()] = int(m.group(2))
import subprocess
for k, v in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().replace("rcv::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    print("%-46s VGPR %3d AGPR %3d spill %3d  scratch %5d  occ %d  LDS %6d" % (name[:46], v.get("VGPRs", -1), v.get("AGPRs", -1), v.get("VGPRs Spill", -1), v.get("ScratchSize", -1), v.get("Occupancy", -1), v.get("LDS Size", -1)))
'
rm -f /tmp/kres_$$.o
