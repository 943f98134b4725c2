#!/bin/bash
# Runs ON THE GPU BOX: LDS conflict counters of the FCN kernels at the bench batch (own PMC pass, kernel trace only).
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/${1:-lds1}; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/lds -- python3 $R/tools/fcn_bench.py --dtype f16x3 --batch 53 --iters 1 > $O/lds.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE --output-format csv -d $O/lds2 -- python3 $R/tools/fcn_bench.py --dtype f16x3 --batch 53 --iters 1 > $O/lds2.log 2>&1
find $O -type f ! -name "*.csv" ! -name "*.log" -delete
for f in $O/*.log; do tail -n 2 $f; done
