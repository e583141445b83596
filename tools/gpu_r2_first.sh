#!/bin/bash
# round 2, first GPU pass: step_multi parity, whole GPU suite, K-step A/B on cfg3
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_step_multi.py -x -q > gpurun_out/t_multi.log 2>&1; rc=$?; tail -5 gpurun_out/t_multi.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 600 python tools/bench_multi.py --steps 1024 --k 0 1 8 32 64 128 > gpurun_out/bench_multi_cfg3.log 2>&1; rc=$?; cat gpurun_out/bench_multi_cfg3.log | cut -c1-200; [ $rc -eq 0 ] || exit 1
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; rc=$?; tail -5 gpurun_out/gpu_tests.log; [ $rc -eq 0 ] || exit 1
