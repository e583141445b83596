#!/bin/bash
# simulate / frame dispatch time against the steps per dispatch (no pipelining): fixed cost per dispatch = intercept
cd "$(dirname "$0")/.."
for K in 2 4 8 16 32 64; do
TC_CHUNK=0 timeout -k 10 300 python bench.py --steps $((K*8)) --warmup $K --steps-per-launch $K --preroll-ms 100 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; k=r['kernels_us']; print('K=$K per dispatch: sim %.1f us  frame %.1f us   step %.2f us' % (list(k.values())[0], list(k.values())[-1], d['ms_per_step']*1e3))"
done
