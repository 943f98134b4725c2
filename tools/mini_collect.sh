ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/r04p6; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $OUT/stats.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/valu -- python3 $ROOT/tools/vote_batch_bench.py --frames 53 --iters 2 > $OUT/valu.log 2>&1
find $OUT -name "*.csv" -size +20M -delete
find $OUT -type f ! -name "*.csv" ! -name "*.log" -delete
du -sh $OUT
