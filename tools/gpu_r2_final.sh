#!/bin/bash
# round 2 evidence, part 1: GPU suite, then rocprofv3 kernel stats + PMC passes for cfg3, cfg2, cfg4, cfg5
# (summarise with tools/summarize_pmc.py, then run tools/gpu_r2_lines.sh for the bench lines that quote the PMC traffic)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; rc=$?; tail -5 gpurun_out/gpu_tests.log; [ $rc -eq 0 ] || exit 1
bash tools/gpu_prof_all.sh || exit 1
bash tools/gpu_prof_cfg5.sh || exit 1
