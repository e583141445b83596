#!/bin/bash
# cfg4 (knuffingen, 128x128 classes, 4096 envs) under the launch / grouping switches; one line per variant
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for v in "TC_GROUPS=1 TC_FUSE=1" "TC_GROUPS=1 TC_FUSE=0" "TC_GROUPS=0 TC_FUSE=0" "TC_GROUPS=1 TC_FUSE=1 TC_BAND_BYTES=8192"; do
  env $v timeout -k 10 120 python bench.py --workload cfg4 --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/v.json 2> gpurun_out/v.err || { echo "FAILED $v"; tail -3 gpurun_out/v.err; exit 1; }
  python -c "import json,sys;d=json.load(open('gpurun_out/v.json'));print(sys.argv[1],'|',round(d['ms_per_step']*1000,1),'us',round(d['value']/1e6,2),'M/s',d['roofline']['kernels_us'],d['config']['lds_bytes_per_env'])" "$v"
done
