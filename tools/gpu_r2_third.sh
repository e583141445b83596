#!/bin/bash
# round 2: GPU suite, then bench lines for cfg3 (default), single-step entry, cfg2, cfg4
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; rc=$?; tail -5 gpurun_out/gpu_tests.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py > gpurun_out/bench_cfg3.json 2> gpurun_out/bench_cfg3.err || { tail -5 gpurun_out/bench_cfg3.err; exit 1; }
cut -c1-1500 gpurun_out/bench_cfg3.json
timeout -k 10 300 python bench.py --steps-per-launch 0 --no-cpu-baseline > gpurun_out/bench_cfg3_single.json 2>/dev/null || exit 1
cut -c1-300 gpurun_out/bench_cfg3_single.json
timeout -k 10 300 python bench.py --workload cfg4 --no-cpu-baseline > gpurun_out/bench_cfg4.json 2>/dev/null || exit 1
cut -c1-300 gpurun_out/bench_cfg4.json; python -c "import json;d=json.load(open('gpurun_out/bench_cfg4.json'));print(d['roofline']['kernels_us'], d['roofline']['step_frac'])"
timeout -k 10 300 python bench.py --workload cfg2 --no-cpu-baseline > gpurun_out/bench_cfg2.json 2>/dev/null || exit 1
cut -c1-300 gpurun_out/bench_cfg2.json
