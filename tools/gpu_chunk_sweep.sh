#!/bin/bash
cd "$(dirname "$0")/.."
for ch in 16 8 12 24 32; do
TC_CHUNK=$ch timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('TC_CHUNK=$ch cfg3', round(d['ms_per_step']*1e3,2), 'us/step', round(d['value']/1e6,1), 'M')"
done
for m in 64 256; do
timeout -k 10 300 python bench.py --steps-per-launch $m --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps per call $m cfg3', round(d['ms_per_step']*1e3,2), 'us/step', round(d['value']/1e6,1), 'M')"
done
