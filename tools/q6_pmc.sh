#!/bin/bash
# Runs ON THE GPU BOX: matrix-core / VALU / LDS counters of the FCN kernels with fp6 cross terms on (engine option 11), at the bench
# batch.  Two PMC passes with --kernel-trace only; the program follows `--` directly.   bash tools/q6_pmc.sh [tag]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/${1:-r05q6pmc}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -- python3 $R/tools/fcn_bench.py --dtype f16x3 --batch 53 --iters 1 --q6 > $O/mfma.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/valu -- python3 $R/tools/fcn_bench.py --dtype f16x3 --batch 53 --iters 1 --q6 > $O/valu.log 2>&1
find $O -type f ! -name "*.csv" ! -name "*.log" -delete
ls -R $O | head
