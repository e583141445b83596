#!/bin/bash
# instruction counts of the grouped simulate kernel per wavefront-step (8 envs), candidate grid on / off
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for g in 1 0; do
  rm -rf /tmp/eg_$g
  TC_CAND_GRID=$g TC_CHUNK=0 timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d /tmp/eg_$g -- python3 $R/bench.py --steps 64 --warmup 2048 --steps-per-launch 16 --preroll-ms 0 --no-cpu-baseline > /tmp/eg_$g.log 2>&1 || { echo "grid $g failed"; tail -3 /tmp/eg_$g.log; continue; }
  python3 - "$g" <<'PY'
import csv, glob, sys, collections
g = sys.argv[1]
agg = collections.defaultdict(float)
for p in glob.glob(f"/tmp/eg_{g}/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(p)):
        if "tc_envg_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
w = (agg.get("SQ_WAVES", 0) or 1) * 16  # 16 steps per dispatch
print("grid=%s per wavefront-step: " % g + "  ".join("%s %.0f" % (k.replace("SQ_", ""), v / w) for k, v in sorted(agg.items()) if k != "SQ_WAVES"))
PY
done
