#!/bin/bash
# frame-kernel time of a workload with parts switched off (a make dev-ablate build of the workload's variants must be the
# library in place, TC_DEBUG_FLAGS): tools/gpu_ablate_wl.sh "<bench args>" flags...
cd "$(dirname "$0")/.."
ARGS="$1"; shift
for f in ${@:-0 0x200 0x100 0x300 0x100000}; do
  TC_DEBUG_FLAGS=$f timeout -k 10 200 python bench.py $ARGS --no-cpu-baseline --no-single-step 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('flags %-9s' % '$f', round(d['ms_per_step']*1e3,1), 'us/step  frame', round(r['kernels_us']['tc_frame_kernel'],1), 'spd', r['steps_per_dispatch'], 'simulate', round([v for k,v in r['kernels_us'].items() if 'frame' not in k][0],1))"
done
