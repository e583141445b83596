#!/bin/bash
# cfg3 step time with parts of the fused kernel switched off (TC_DEBUG_FLAGS, profiling only): where the time goes
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for f in "0 full" "0x100 no-raster-drawing(stores_kept)" "0x200 no-obs-stores" "0x300 no-raster-at-all" "0x400 no-camera+raster" "0xC00 kinematics+tracking-only" "0x800 no-phaseB" "0x1000 no-outline-pixels" "0x2000 no-fill-rows" "0x4000 no-caps" "0xF0000 no-raster-setup"; do
  set -- $f
  TC_DEBUG_FLAGS=$1 timeout -k 10 120 python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', round(d['roofline']['kernel_us'],1),'us')" || { echo "FAILED $f"; exit 1; }
done
