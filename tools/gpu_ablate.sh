#!/bin/bash
# frame-kernel time with parts of the raster stage switched off (TC_DEBUG_FLAGS; results are wrong on purpose)
cd "$(dirname "$0")/.."
for f in 0 0x100 0x200 0x300 0x1000 0x2000 0x4000 0x7000 0x10000 0x20000 0x80000; do
  TC_DEBUG_FLAGS=$f timeout -k 10 120 python bench.py --no-cpu-baseline --steps 512 --preroll-ms 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_us']; print('flags %-8s step %.2f us  env %.1f  frame %.1f us/step' % ('$f', d['ms_per_step']*1e3, k['tc_env_kernel']/32, k['tc_frame_kernel']/32))"
done
