#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the rocprofv3 passes behind profiles/r03* (r02: the same passes with the bf16 engine at 72 frames).  Kernel trace / stats and each PMC
# counter set are separate runs (never --pmc together with the sys/hip trace domains); the program follows `--` directly.
#   tools/collect_profiles.sh <tag>          -> gpurun_out/<tag>/...
TAG=${1:-r04}
DT=${2:-f16x3}     # FCN arithmetic of the traffic / matrix-core passes (bench.py's default)
FB=${3:-53}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; echo "== $name: rocprofv3 $*"; timeout 600 rocprofv3 "$@" > $OUT/$name.log 2>&1; echo "rc $?"; }
# 1. kernel stats of the bench command itself
run stats --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras
# 2. HBM traffic of the FCN kernels at three batch sizes (FETCH_SIZE and WRITE_SIZE cannot share a pass)
for B in 12 $FB; do
  run fetch_b$B --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_b$B -- python3 $ROOT/tools/fcn_bench.py --dtype $DT --batch $B --iters 1
  run write_b$B --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write_b$B -- python3 $ROOT/tools/fcn_bench.py --dtype $DT --batch $B --iters 1
done
# 3. matrix-core busy / wave cycles of the convolution kernels at the bench batch
run mfma --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma -- python3 $ROOT/tools/fcn_bench.py --dtype $DT --batch $FB --iters 1
run mfma2 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/mfma2 -- python3 $ROOT/tools/fcn_bench.py --dtype $DT --batch $FB --iters 1
# 4. VALU occupancy of the voting kernels at the bench batch
run valu --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/valu -- python3 $ROOT/tools/vote_batch_bench.py --frames $FB --iters 2
run vote_fetch --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/vote_fetch -- python3 $ROOT/tools/vote_batch_bench.py --frames $FB --iters 2
run vote_write --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/vote_write -- python3 $ROOT/tools/vote_batch_bench.py --frames $FB --iters 2
# keep only the CSVs (the merged-back directory is capped at 64 MiB)
find $OUT -name "*.csv" -size +20M -delete
find $OUT -type f ! -name "*.csv" ! -name "*.log" -delete
du -sh $OUT; ls $OUT/*/ 2>/dev/null | head -40; tail -3 $OUT/*.log
