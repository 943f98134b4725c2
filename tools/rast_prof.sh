#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/${1:-rast1}; mkdir -p $O
python3 $R/tools/vote_batch_bench.py --frames 24 --iters 3 --engine full 2>&1 | grep -v amdgpu.ids > $O/plain.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/vote_batch_bench.py --frames 24 --iters 3 --engine full > $O/stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/valu -- python3 $R/tools/vote_batch_bench.py --frames 24 --iters 2 --engine full > $O/valu.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_WAIT_ANY --output-format csv -d $O/lds -- python3 $R/tools/vote_batch_bench.py --frames 24 --iters 2 --engine full > $O/lds.log 2>&1
find $O -type f ! -name "*.csv" ! -name "*.log" ! -name "*.txt" -delete
cat $O/plain.txt; for f in $O/*.log; do tail -n 2 $f; done
