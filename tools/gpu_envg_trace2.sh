#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/et2
TC_CAND_GRID=${1:-1} TC_CHUNK=0 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/et2 -- python3 $R/bench.py --steps 3072 --warmup 32 --steps-per-launch 32 --preroll-ms 0 --no-cpu-baseline > /tmp/et2.log 2>&1 || { tail -3 /tmp/et2.log; exit 1; }
python3 - <<'PY'
import csv, glob
rows=[]
for p in glob.glob("/tmp/et2/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(p)):
        if "tc_envg" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])-int(r["Start_Timestamp"])))
rows.sort()
print("envg dispatch durations (us), in launch order:", [round(d/1e3) for _,d in rows])
PY
