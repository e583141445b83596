#!/bin/bash
# frame-kernel time with parts of the camera / raster stage switched off (TC_DEBUG_FLAGS; results are wrong on purpose).
# TC_CHUNK=0: no pipelining, so the frame kernel runs alone behind the simulate kernel and its time is its own.
cd "$(dirname "$0")/.."
WL=${1:-cfg3}
for f in 0 0x100 0x200 0x300 0x100300 0x100000 0x1000 0x2000 0x4000 0x7000 0x10000 0x20000 0x40000 0x80000; do
  TC_CHUNK=0 TC_DEBUG_FLAGS=$f timeout -k 10 120 python bench.py --workload $WL --no-cpu-baseline --steps 256 --warmup 32 --steps-per-launch 32 --preroll-ms 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_us']; print('$WL flags %-9s step %.2f us  sim %.2f  frame %.2f us/step' % ('$f', d['ms_per_step']*1e3, list(k.values())[0]/32, list(k.values())[-1]/32))"
done
