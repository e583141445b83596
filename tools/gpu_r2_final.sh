#!/bin/bash
# round 2 evidence run: GPU suite, then bench line + rocprofv3 kernel stats + PMC passes for cfg3, cfg2, cfg4, cfg5, wrappers
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; rc=$?; tail -5 gpurun_out/gpu_tests.log; [ $rc -eq 0 ] || exit 1
bash tools/gpu_prof_all.sh || exit 1
bash tools/gpu_prof_cfg5.sh || exit 1
timeout -k 10 600 python tools/bench_wrappers.py --steps 640 > gpurun_out/wrappers_bench.json 2> gpurun_out/wrappers_bench.err || { tail -5 gpurun_out/wrappers_bench.err; exit 1; }
cat gpurun_out/wrappers_bench.json
timeout -k 10 300 python bench.py --steps-per-launch 0 --no-cpu-baseline > gpurun_out/bench_cfg3_single.json 2>/dev/null || exit 1
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_cfg3_driver_style.json 2>/dev/null || exit 1
cut -c1-400 gpurun_out/bench_cfg3_driver_style.json
