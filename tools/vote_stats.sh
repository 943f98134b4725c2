#!/bin/bash
# Runs ON THE GPU BOX: per-kernel times of the voting stage at the bench batch (rocprofv3 --stats of tools/vote_batch_bench.py)
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/${1:-vstats}; mkdir -p $O
python3 $R/tools/vote_batch_bench.py --frames 53 --iters 5 2>&1 | grep -v amdgpu.ids > $O/plain.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/vote_batch_bench.py --frames 53 --iters 3 > $O/stats.log 2>&1
find $O -type f ! -name "*.csv" ! -name "*.log" ! -name "*.txt" -delete
cat $O/plain.txt
