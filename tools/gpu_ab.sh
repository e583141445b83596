#!/bin/bash
# same-box A/B of library builds: tools/gpu_ab.sh "<bench args>" libA.so libB.so ...  (each build twice, interleaved)
R=$GRAFT_REPO_ROOT
cd $R
ARGS="$1"; shift
for rep in 1 2; do
  for lib in "$@"; do
    TINYCARLO_HIP_LIB=$R/tinycarlo_amd/$lib timeout -k 10 300 python bench.py $ARGS --no-cpu-baseline --no-single-step 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-12s' % '$lib', round(d['value']/1e6,2), 'M/s', round(d['ms_per_step']*1e3,2), 'us/step', {k:round(v,1) for k,v in r['kernels_us'].items()}, 'spd', r['steps_per_dispatch'])"
  done
done
