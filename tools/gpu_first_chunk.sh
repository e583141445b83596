#!/bin/bash
cd "$(dirname "$0")/.."
timeout -k 10 400 python -m pytest tests/test_gpu_step_multi.py -x -q -k "simple_layout-r64-classes or stress or against_oracle or wrap or noise" 2>&1 | tail -2
for f in 0 1; do
for K in 20 1024; do
TC_FIRST_CHUNK_PER_ENV=$f timeout -k 10 300 python bench.py --steps $K --warmup $((K<128?5:128)) --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('first chunk per-env=$f K=$K', round(d['ms_per_step']*1e3,2), 'us/step', round(d['value']/1e6,1), 'M')"
done; done
