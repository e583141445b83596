#!/bin/bash
# env-switch variants of the streamed call: bench default and the 20-step command each
R=$GRAFT_REPO_ROOT
cd $R && mkdir -p gpurun_out/stream && O=gpurun_out/stream
line() { f=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline --no-single-step "$@" > $O/$f.json 2> $O/$f.err || { tail -3 $O/$f.err; return 1; }
  python - $O/$f.json $f <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']
print(sys.argv[2], 'M/s', round(d['value']/1e6,2), 'us/step', round(d['ms_per_step']*1e3,2), 'frame_us', round(r['kernel_us'],1), 'spd', r['steps_per_dispatch'], 'sim_us', {k:round(v,1) for k,v in r['kernels_us'].items()})
PY
}
for v in "${@}"; do
  tag=${v//[^A-Za-z0-9]/_}
  env $v true
  ( export $v; line d_$tag && line s_$tag --steps 20 --warmup 5 ) || exit 1
done
