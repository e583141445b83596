#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for g in 1 0; do
  rm -rf /tmp/et_$g
  TC_CAND_GRID=$g TC_CHUNK=0 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/et_$g -- python3 $R/bench.py --steps 256 --warmup 32 --steps-per-launch 32 --preroll-ms 100 --no-cpu-baseline > /tmp/et_$g.log 2>&1 || { echo "grid $g failed"; tail -3 /tmp/et_$g.log; continue; }
  echo "grid=$g"; grep -h "tc_" /tmp/et_$g/*/*_kernel_stats.csv | cut -c1-150
  python3 -c "
import json,sys
for l in open('/tmp/et_$g.log'):
    if l.startswith('{'):
        d=json.loads(l); print('  bench line: step', round(d['ms_per_step']*1e3,2), d['roofline']['kernels_us'])
"
done
