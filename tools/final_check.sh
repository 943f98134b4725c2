mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04f_tests.log 2>&1; tail -3 gpurun_out/r04f_tests.log
timeout 900 python bench.py > gpurun_out/r04f_bench.json 2> gpurun_out/r04f_bench.err; tail -c 600 gpurun_out/r04f_bench.json
timeout 1200 bash tools/collect_profiles.sh r04p5 f16x3 53 > gpurun_out/r04p5_collect.log 2>&1; tail -2 gpurun_out/r04p5_collect.log
timeout 600 python tools/driver_bench.py --frames 128 --frames_per_step 53 --out gpurun_out/r04f_driver_bench.json > gpurun_out/r04f_driver.log 2>&1; tail -5 gpurun_out/r04f_driver.log
