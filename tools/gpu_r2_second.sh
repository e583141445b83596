#!/bin/bash
# round 2: GPU suite, K-step A/B on cfg3, phase clock (instrumented build), LDS conflict ablations
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; rc=$?; tail -5 gpurun_out/gpu_tests.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 600 python tools/bench_multi.py --steps 1024 --k 0 32 128 > gpurun_out/bench_multi_cfg3.log 2>&1; rc=$?; cat gpurun_out/bench_multi_cfg3.log | cut -c1-160; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/phase_clock.py --envs 64 4096 --steps 12 --multi 32 > gpurun_out/phase_clock_multi32.txt 2>&1 || { tail -5 gpurun_out/phase_clock_multi32.txt; exit 1; }
cat gpurun_out/phase_clock_multi32.txt
bash tools/gpu_pmc_lds.sh gpurun_out/pmc_lds.txt
