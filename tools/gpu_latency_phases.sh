#!/bin/bash
# per-phase LATENCY view: 64 envs = one wave per CU, nothing to contend with; step time with parts switched off
cd "$(dirname "$0")/.."
for f in "0 full" "0x100 no-raster-drawing" "0x200 no-obs-stores" "0x300 no-raster-at-all" "0x400 no-camera+raster" "0xC00 track-only" "0x1000 no-outline" "0x2000 no-fill" "0x4000 no-caps" "0xF0000 no-raster-setup" "0x10000 no-fill-events" "0x20000 no-line-setup" "0x40000 no-slopes" "0x80000 no-quad"; do
  set -- $f
  TC_DEBUG_FLAGS=$1 timeout -k 10 120 python bench.py --envs 64 --steps 400 --warmup 40 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', round(d['ms_per_step']*1000,1), 'us step')" || { echo "FAILED $f"; exit 1; }
done
