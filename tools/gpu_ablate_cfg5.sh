#!/bin/bash
# cfg5 frame-kernel time with parts switched off (make dev-ablate DEVK=9 DEVFMT=TC_FMT_RGB build, TC_DEBUG_FLAGS)
cd "$(dirname "$0")/.."
for f in ${@:-0 0x200 0x100 0x300 0x1000 0x2000 0x4000 0x20000 0x40000 0x80000 0x10000}; do
  TC_DEBUG_FLAGS=$f timeout -k 10 200 python bench.py --workload cfg5 --no-cpu-baseline --no-single-step --steps 16 --warmup 4 --preroll-ms 50 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('flags %-9s' % '$f', round(d['ms_per_step']*1e3,1), 'us/step  frame', round(r['kernels_us']['tc_frame_kernel'],1), 'spd', r['steps_per_dispatch'], 'simulate', round([v for k,v in r['kernels_us'].items() if 'frame' not in k][0],1))"
done
