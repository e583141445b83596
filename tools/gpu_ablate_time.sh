#!/bin/bash
# frame-kernel time with parts of it switched off (make dev-ablate build, TC_DEBUG_FLAGS): what a part costs in TIME, to
# set against what it costs in instructions (tools/gpu_pmc_ablate.sh)
cd "$(dirname "$0")/.."
for f in ${@:-0 0x200 0x100 0x300 0x1000 0x2000 0x4000 0x10000 0x20000 0x40000 0x80000 0x100000}; do
  TC_DEBUG_FLAGS=$f timeout -k 10 100 python bench.py --no-cpu-baseline --no-single-step --steps 512 --warmup 1024 --preroll-ms 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('flags %-9s' % '$f', round(d['ms_per_step']*1e3,2), 'us/step  frame', round(r['kernels_us']['tc_frame_kernel'],1), 'simulate', round([v for k,v in r['kernels_us'].items() if 'frame' not in k][0],1))"
done
