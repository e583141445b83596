#!/bin/bash
cd "$(dirname "$0")/.."
for zf in 0 1; do
TC_RGB_ZERO_FIRST=$zf timeout -k 10 400 python bench.py --workload cfg5 --steps 32 --warmup 8 --preroll-ms 50 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('adaptive band, zero_first=$zf: cfg5', round(d['value']/1e6,3), 'M', round(d['ms_per_step']*1e3,1), 'us/step  lds', d['config']['lds_bytes_per_env'], 'frac', round(r['frac'],3))"
TC_RGB_ZERO_FIRST=$zf TC_BAND_BYTES=16384 timeout -k 10 400 python bench.py --workload cfg5 --steps 32 --warmup 8 --preroll-ms 50 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('16 KB band, zero_first=$zf: cfg5', round(d['value']/1e6,3), 'M', round(d['ms_per_step']*1e3,1), 'us/step  lds', d['config']['lds_bytes_per_env'], 'frac', round(r['frac'],3))"
done
