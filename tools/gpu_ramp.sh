#!/bin/bash
cd "$(dirname "$0")/.."
for r in 0 3; do
for K in 20 40 1024; do
TC_RAMP=$r timeout -k 10 300 python bench.py --steps $K --warmup $((K<128?5:128)) --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ramp=$r K=$K', round(d['ms_per_step']*1e3,2), 'us/step', round(d['value']/1e6,1), 'M')"
done; done
