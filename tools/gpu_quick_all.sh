#!/bin/bash
# one bench line per workload (no profiles): value, us/step, dominant kernel, fractions
cd "$(dirname "$0")/.."
for wl in cfg3 cfg2 cfg4; do
timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$wl', round(d['value']/1e6,2), 'M env-steps/s', round(d['ms_per_step']*1e3,2), 'us/step', {k: round(v,1) for k,v in r['kernels_us'].items()}, 'frac', round(r['frac'],3), 'step_frac', round(r['step_frac'],3))"
done
timeout -k 10 400 python bench.py --workload cfg5 --steps 64 --warmup 8 --preroll-ms 50 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('cfg5', round(d['value']/1e6,2), 'M env-steps/s', round(d['ms_per_step']*1e3,2), 'us/step', {k: round(v,1) for k,v in r['kernels_us'].items()}, 'frac', round(r['frac'],3), 'step_frac', round(r['step_frac'],3))"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg3 driver-style 20 steps', round(d['value']/1e6,2), 'M', round(d['ms_per_step']*1e3,2), 'us/step')"
timeout -k 10 300 python bench.py --steps-per-launch 0 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg3 tc_step', round(d['value']/1e6,2), 'M', round(d['ms_per_step']*1e3,2), 'us/step')"
