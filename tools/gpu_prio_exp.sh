#!/bin/bash
# issue priority of the camera stage (latency chains) against the raster stage (dense vector work) inside tc_frame_kernel
cd "$(dirname "$0")/.."
for lib in "" explibs/lib_prio30.so explibs/lib_prio21.so explibs/lib_prio03.so; do
  TINYCARLO_HIP_LIB=${lib:+$PWD/$lib} timeout -k 10 100 python bench.py --no-cpu-baseline --no-single-step --steps 512 --warmup 1024 --preroll-ms 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('${lib:-shipped}', round(d['ms_per_step']*1e3,2), 'us/step  frame', round(r['kernels_us']['tc_frame_kernel'],1), 'simulate', round([v for k,v in r['kernels_us'].items() if 'frame' not in k][0],1))"
done
