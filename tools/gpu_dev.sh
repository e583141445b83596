#!/bin/bash
# dev build (cfg3 kernels only): step_multi parity on simple_layout, bench with pipelined K-step calls
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_step_multi.py -x -q -k "simple_layout-r64-classes or stress or fused_terms or against_oracle or without_rollout or no_observation or wrap or noise" > gpurun_out/t_dev.log 2>&1; rc=$?; tail -4 gpurun_out/t_dev.log; [ $rc -eq 0 ] || exit 1
for ch in 16 8 32 0; do
  TC_CHUNK=$ch timeout -k 10 300 python bench.py --steps-per-launch 128 --steps 1024 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('chunk=$ch K=128 cfg3', round(d['ms_per_step']*1e3,2), 'us/step', {k: round(v/128,2) for k,v in d['roofline']['kernels_us'].items()})"
done
TC_CHUNK=16 timeout -k 10 300 python bench.py --steps-per-launch 256 --steps 1024 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('chunk=16 K=256 cfg3', round(d['ms_per_step']*1e3,2), 'us/step', {k: round(v/256,2) for k,v in d['roofline']['kernels_us'].items()})"
TC_CHUNK=16 timeout -k 10 300 python bench.py --steps-per-launch 64 --steps 1024 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('chunk=16 K=64 cfg3', round(d['ms_per_step']*1e3,2), 'us/step', {k: round(v/64,2) for k,v in d['roofline']['kernels_us'].items()})"
