#!/bin/bash
# cost-aware env order of single-step launches: closed-loop step time with the order refreshed every n-th step (0 = off)
cd "$(dirname "$0")/.."
for n in 0 1 4 8 32; do
  TC_STEP_ORDER=$n timeout -k 10 100 python bench.py --no-cpu-baseline --steps 64 --warmup 64 --preroll-ms 150 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); v=d['value_single_step']
print('TC_STEP_ORDER=$n single step', round(v['ms_per_step']*1e3,1), 'us', round(v['value']/1e6,1), 'M env-steps/s')"
done
