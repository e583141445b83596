#!/bin/bash
cd "$(dirname "$0")/.."
for r in 0 2 3; do
TC_RAMP=$r timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('ramp=$r K=20', round(d['ms_per_step']*1e3,2), 'us/step', round(d['value']/1e6,1), 'M')"
done
