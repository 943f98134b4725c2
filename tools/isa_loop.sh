#!/bin/bash
# usage: tools/isa_loop.sh <source.hip> <kernel-name-substring> [loop depth]   -- VALU histogram of a kernel's inner loop
R=$(cd "$(dirname "$0")/.." && pwd)
SRC=$R/rcvpose_amd/csrc/$1
EXTRA=""; [ "$1" != "fcn.hip" ] && EXTRA="-ffp-contract=off"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $EXTRA -I$R/include -I$R/rcvpose_amd/csrc -S --cuda-device-only $SRC -o /tmp/isa_$$.s || exit 1
L=$(grep -n "^_Z.*$2.*:" /tmp/isa_$$.s | head -1 | cut -d: -f1)
awk -v L=$L 'NR>=L{print} NR>L && /s_endpgm/{exit}' /tmp/isa_$$.s > /tmp/kernel.s
D=${3:-3}
awk -v d="Inner Loop Header: Depth=$D" 'index($0,d){f=1} f{print} f&&/s_cbranch/{exit}' /tmp/kernel.s > /tmp/loop.s
grep "^\s[vd]_\|^\ss_wait\|^\sbuffer\|^\sglobal" /tmp/loop.s | awk '{print $1}' | sort | uniq -c | sort -rn
grep -E "^\s+\.(vgpr_count|sgpr_count|agpr_count)|; (NumVgprs|Occupancy|LDSByteSize)" /tmp/kernel.s | head -6
rm -f /tmp/isa_$$.s
