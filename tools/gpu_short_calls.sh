#!/bin/bash
# short K-step calls (the driver's 20-step command): one or two frame streams
cd "$(dirname "$0")/.."
timeout -k 10 400 python -m pytest tests/test_gpu_step_multi.py -x -q -k "simple_layout-r64-classes or stress or against_oracle or wrap or noise" 2>&1 | tail -2
for fs in 2 1; do
for K in 8 20 40 64; do
TC_FRAME_STREAMS=$fs timeout -k 10 300 python bench.py --steps $K --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('frame streams=$fs K=$K', round(d['ms_per_step']*1e3,2), 'us/step', round(d['value']/1e6,1), 'M; dispatches', r['dispatches_per_call'], 'of', r['steps_per_dispatch'], 'steps')"
done; done
