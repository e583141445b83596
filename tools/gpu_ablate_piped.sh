#!/bin/bash
# pipelined K-step calls (default chunking) with parts of the frame kernel switched off: is the call bound by the frame
# kernel or by the simulate kernel running beside it?  (ablation build: make dev-ablate)
cd "$(dirname "$0")/.."
for f in 0 0x100 0x300 0x100300; do
  TC_DEBUG_FLAGS=$f timeout -k 10 120 python bench.py --no-cpu-baseline --steps 512 --warmup 128 --preroll-ms 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_us']; print('piped flags %-9s step %.2f us  per dispatch: sim %.1f  frame %.1f us' % ('$f', d['ms_per_step']*1e3, list(k.values())[0], list(k.values())[-1]))"
done
TC_CHUNK=0 timeout -k 10 120 python bench.py --workload cfg2 --no-cpu-baseline --steps 512 --warmup 128 --preroll-ms 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg2 (tc_env_kernel, no obs)', round(d['ms_per_step']*1e3,2))"
