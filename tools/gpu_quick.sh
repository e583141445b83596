#!/bin/bash
# quick A/B: GPU parity tests, then track-only and full step time at 64 and 4096 envs (cfg3)
cd "$(dirname "$0")/.."
timeout -k 10 400 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; rc=$?; tail -3 gpurun_out/gpu_tests.log; [ $rc -eq 0 ] || exit 1
for n in 64 4096; do
  for f in "0xC00 trackonly" "0x400 A+B" "0 full"; do
    set -- $f
    TC_DEBUG_FLAGS=$1 timeout -k 10 120 python bench.py --envs $n --steps 600 --warmup 60 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('envs $n $2', round(d['ms_per_step']*1000,1), 'us step')" || exit 1
  done
done
