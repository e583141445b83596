#!/bin/bash
# Is the frame kernel bound by vector issue or by latency x occupancy?  The same bench command with unused LDS bytes added
# to every frame workgroup (make dev-exp build): 16 -> 14 -> 12 -> 10 workgroups (= wavefronts) per CU.
cd "$(dirname "$0")/.."
for pad in 0 1280 3072 6144; do
  TC_LDS_PAD=$pad timeout -k 10 200 python bench.py --no-cpu-baseline --no-single-step --steps 512 --warmup 128 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('pad $pad lds', d['config']['lds_bytes_per_env'], round(d['ms_per_step']*1e3,2), 'us/step', 'frame', round(r['kernels_us']['tc_frame_kernel'],1), 'simulate', round([v for k,v in r['kernels_us'].items() if 'frame' not in k][0],1))"
done
