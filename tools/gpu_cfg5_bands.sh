#!/bin/bash
# cfg5 (480x640 rgb, 8192 envs): LDS band size of the raster stage vs workgroups per CU
cd "$(dirname "$0")/.."
for bb in 16384 10240 8192 6144; do
TC_BAND_BYTES=$bb timeout -k 10 400 python bench.py --workload cfg5 --steps 32 --warmup 8 --preroll-ms 50 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('band bytes $bb: cfg5', round(d['value']/1e6,3), 'M', round(d['ms_per_step']*1e3,1), 'us/step  lds', d['config']['lds_bytes_per_env'], 'frac', round(r['frac'],3))"
done
