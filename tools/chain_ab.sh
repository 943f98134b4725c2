#!/bin/bash
# On the GPU box: the chain kernels' launches (1x1 expand with residual at 30x40 / 60x80 / 120x160) under every A-B library in
# tools/probe/ab/ and under the in-tree one, same box, same batch.   tools/chain_ab.sh [batch] [iters]
R=$(cd "$(dirname "$0")/.." && pwd)
B=${1:-53}; I=${2:-3}
run() { echo "=== $1"; RCV_LIB_PATH=$2 python $R/tools/fcn_bench.py --dtype f16x3 --batch $B --iters $I --ignore-range 2>&1 | grep -E "30x40 +256 +1024 1 1|60x80 +128 +512 1 1|120x160 +64 +256 1 1|conv total"; }
run in-tree $R/rcvpose_amd/librcvpose_hip.so
for f in $R/tools/probe/ab/lib_*.so; do case $f in *st_*) continue;; esac; run $(basename $f) $f; done
run in-tree-again $R/rcvpose_amd/librcvpose_hip.so
