#!/bin/bash
# 20-step calls (the driver's command line): how the call is cut into pipelined chunks
cd "$(dirname "$0")/.."
for dv in 4 2 3 1 5 7 10; do
for K in 20; do
TC_CHUNK_DIV=$dv timeout -k 10 300 python bench.py --steps $K --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('div=$dv K=$K', round(d['ms_per_step']*1e3,2), 'us/step; dispatches', r['dispatches_per_call'], {k: round(v,1) for k,v in r['kernels_us'].items()})"
done; done
TC_CHUNK=0 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('no pipelining K=20', round(d['ms_per_step']*1e3,2), 'us/step')"
for dv in 4 2; do
TC_CHUNK_DIV=$dv timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('div=$dv K=40', round(d['ms_per_step']*1e3,2), 'us/step')"
done
