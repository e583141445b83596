#!/bin/bash
# one wave's critical path vs contention: kernel time of the tracking-only and the full step at tiny and full batches
cd "$(dirname "$0")/.."
for n in 1 64 256 1024 4096; do
  for f in "0xC00 trackonly" "0x400 A+B" "0 full"; do
    set -- $f
    TC_DEBUG_FLAGS=$1 timeout -k 10 120 python bench.py --envs $n --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('envs $n $2', round(d['roofline']['kernel_us'],1),'us kernel', round(d['ms_per_step']*1000,1), 'us step')" || { echo "FAILED $n $f"; exit 1; }
  done
done
