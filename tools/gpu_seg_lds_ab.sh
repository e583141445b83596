#!/bin/bash
cd "$(dirname "$0")/.."
for sl in 1 0; do
TC_SEG_LDS=$sl timeout -k 10 400 python bench.py --workload cfg5 --steps 32 --warmup 8 --preroll-ms 50 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('TC_SEG_LDS=$sl cfg5', round(d['value']/1e6,3), 'M', round(d['ms_per_step']*1e3,1), 'us/step  lds', d['config']['lds_bytes_per_env'])"
TC_SEG_LDS=$sl timeout -k 10 300 python bench.py --workload cfg4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('TC_SEG_LDS=$sl cfg4', round(d['value']/1e6,2), 'M', round(d['ms_per_step']*1e3,2), {k: round(v,1) for k,v in d['roofline']['kernels_us'].items()})"
done
