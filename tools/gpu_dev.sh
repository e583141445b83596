#!/bin/bash
# dev build (cfg3 kernels only): step_multi parity on simple_layout, bench, phase clock of the simulate kernel
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_step_multi.py -x -q -k "simple_layout-r64-classes or stress or fused_terms or against_oracle or without_rollout or no_observation or wrap" > gpurun_out/t_dev.log 2>&1; rc=$?; tail -4 gpurun_out/t_dev.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "simple_layout and classes and r64 and not rgb and not terms" > gpurun_out/t_dev2.log 2>&1; rc=$?; tail -4 gpurun_out/t_dev2.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg3', round(d['ms_per_step']*1e3,2), 'us/step', d['roofline']['kernels_us'])"
timeout -k 10 300 python tools/phase_clock.py --envs 4096 --steps 12 --multi 32 --no-obs 2>&1 | grep -v amdgpu.ids
