#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; rc=$?; tail -5 gpurun_out/gpu_tests.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 600 python tools/bench_wrappers.py --steps 640 > gpurun_out/wrappers_bench.json 2> gpurun_out/wrappers_bench.err || { tail -5 gpurun_out/wrappers_bench.err; exit 1; }
cat gpurun_out/wrappers_bench.json
timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg3', round(d['ms_per_step']*1e3,2), 'us/step', d['roofline']['kernels_us'])"
