#!/bin/bash
cd "$(dirname "$0")/.."
for r in 1 0; do
for K in 20 40 100; do
TC_RAMP=$r timeout -k 10 300 python bench.py --steps $K --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ramp=$r cfg3 $K steps', round(d['ms_per_step']*1e3,2), 'us/step')"
done
done
TC_CHUNK=0 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('no pipelining, grouped sim, 20 steps', round(d['ms_per_step']*1e3,2), 'us/step')"
TC_CHUNK=0 TC_ENV_GROUPED=0 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('no pipelining, per-env sim, 20 steps', round(d['ms_per_step']*1e3,2), 'us/step')"
