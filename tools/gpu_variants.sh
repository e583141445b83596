#!/bin/bash
# A/B of build variants of the grouped simulate kernel beside the frame kernel (dev builds, cfg3 default command)
cd "$(dirname "$0")/.."
for v in "" nt64 nt128 prio0 prio1; do
if [ -z "$v" ]; then lib=""; else lib=$PWD/tinycarlo_amd/libtc_var_$v.so; fi
TINYCARLO_HIP_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant [$v]: cfg3', round(d['ms_per_step']*1e3,2), 'us/step', {k: round(x,1) for k,x in d['roofline']['kernels_us'].items()})"
done
