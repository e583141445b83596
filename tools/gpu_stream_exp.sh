#!/bin/bash
# streamed K-step calls against the chunked pipeline (TC_STREAM=0): parity first, then the bench lines
R=$GRAFT_REPO_ROOT
cd $R && mkdir -p gpurun_out/stream && O=gpurun_out/stream
timeout -k 10 600 python -m pytest tests/test_gpu_bench_shapes.py -x -q -k "${TC_EXP_TESTS:-cfg3 or every_switch or longer}" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
line() { f=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > $O/$f.json 2> $O/$f.err || { tail -3 $O/$f.err; return 1; }
  python - $O/$f.json $f <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']
print(sys.argv[2], 'M/s', round(d['value']/1e6,2), 'us/step', round(d['ms_per_step']*1e3,2), 'frame_us', round(r['kernel_us'],1), 'spd', r['steps_per_dispatch'], 'sim_us', {k:round(v,1) for k,v in r['kernels_us'].items()}, 'single', round((d.get('value_single_step') or {}).get('ms_per_step',0)*1e3,1))
PY
}
line s_default --no-single-step || exit 1
line s_driver --steps 20 --warmup 5 --no-single-step || exit 1
TC_STREAM=0 line c_default --no-single-step || exit 1
TC_STREAM=0 line c_driver --steps 20 --warmup 5 --no-single-step || exit 1
line s_default2 --no-single-step || exit 1
line s_driver2 --steps 20 --warmup 5 --no-single-step || exit 1
