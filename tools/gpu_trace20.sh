#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/t20
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/t20 -- python3 $R/bench.py --steps ${1:-20} --warmup 5 --no-cpu-baseline > /tmp/t20.log 2>&1 || { tail -3 /tmp/t20.log; exit 1; }
grep '^{' /tmp/t20.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench: ms_per_step', round(d['ms_per_step']*1e3,2), 'us; call =', round(d['ms_per_step']*d['steps']*1e3), 'us')"
python3 - <<'PY'
import csv, glob
rows=[]
for p in glob.glob("/tmp/t20/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(p)):
        if "tc_" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:22]))
rows.sort()
last=rows[-8:]
t0=last[0][0]
for s,e,n in last: print("%-24s start %7.1f  end %7.1f  dur %6.1f us" % (n,(s-t0)/1e3,(e-t0)/1e3,(e-s)/1e3))
PY
