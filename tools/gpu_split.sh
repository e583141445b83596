#!/bin/bash
# dev build (cfg3 kernels only): step_multi parity on simple_layout, then split vs fused K-step launches
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_step_multi.py -x -q -k "simple_layout-r64-classes or stress or fused_terms or oracle or without_rollout or no_observation or wrap" > gpurun_out/t_multi.log 2>&1; rc=$?; tail -5 gpurun_out/t_multi.log; [ $rc -eq 0 ] || exit 1
for s in 1 0; do
  TC_MULTI_SPLIT=$s timeout -k 10 600 python tools/bench_multi.py --steps 1024 --k 8 32 128 2>/dev/null | cut -c1-200 | sed "s/^/split=$s /"
done
