#!/bin/bash
# dev build (make -C tinycarlo_amd/csrc dev: cfg3 kernels only): parity on simple_layout / stress map, then the frame kernel alone
# (TC_CHUNK=0), the default bench and the 20-step call
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_step_multi.py tests/test_gpu_candidate_grid.py -x -q -k "simple_layout-r64-classes or stress or against_oracle or wrap or noise or displaced and simple or switched" > gpurun_out/t_dev.log 2>&1; rc=$?; tail -3 gpurun_out/t_dev.log; [ $rc -eq 0 ] || exit 1
TC_CHUNK=0 timeout -k 10 120 python bench.py --no-cpu-baseline --steps 256 --warmup 32 --steps-per-launch 32 --preroll-ms 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_us']; print('unpipelined: step %.2f us  sim %.2f  frame %.2f us/step' % (d['ms_per_step']*1e3, list(k.values())[0]/32, list(k.values())[-1]/32))"
timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg3 default', round(d['ms_per_step']*1e3,2), 'us/step', round(d['value']/1e6,1), 'M', {k: round(v,1) for k,v in d['roofline']['kernels_us'].items()}, 'frac', round(d['roofline']['frac'],3))"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg3 20-step call', round(d['ms_per_step']*1e3,2), 'us/step', round(d['value']/1e6,1), 'M')"
