#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the whole GPU suite and the bench line of the current tree.
#   tools/final_check.sh <tag> [profiles]     -> gpurun_out/<tag>_tests.log, <tag>_bench.json (+ the rocprofv3 passes and the driver bench)
TAG=${1:-final}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_tests.log 2>&1; tail -n 3 gpurun_out/${TAG}_tests.log
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -c 300 gpurun_out/${TAG}_bench.json
if [ "$2" = "profiles" ]; then
  timeout 1200 bash tools/collect_profiles.sh ${TAG}_prof f16x3 53 > gpurun_out/${TAG}_collect.log 2>&1; tail -n 2 gpurun_out/${TAG}_collect.log
  timeout 600 python tools/driver_bench.py --frames 128 --frames_per_step 53 --out gpurun_out/${TAG}_driver_bench.json > gpurun_out/${TAG}_driver.log 2>&1; tail -n 2 gpurun_out/${TAG}_driver.log
fi
