#!/bin/bash
# dynamic VALU / SALU instructions per frame wavefront of tc_frame_kernel with parts of it switched off (TC_DEBUG_FLAGS):
# the difference to flags 0 is what the part executes.  usage (through gpurun): bash tools/gpu_pmc_ablate.sh [workload] [flags...]
R=$GRAFT_REPO_ROOT; WL=${1:-cfg3}; shift
FL=${@:-0 0x200 0x100 0x300 0x1000 0x2000 0x4000 0x10000 0x20000 0x40000 0x80000 0x100000}
cd /tmp && export TMPDIR=/tmp
for f in $FL; do
  rm -rf /tmp/pa_$f
  TC_DEBUG_FLAGS=$f TC_CHUNK=0 timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d /tmp/pa_$f -- python3 $R/bench.py --workload $WL --steps 32 --warmup 2048 --steps-per-launch 16 --preroll-ms 0 --no-cpu-baseline > /tmp/pa_$f.log 2>&1 || { echo "flags $f failed"; tail -3 /tmp/pa_$f.log; continue; }
  python3 - "$f" <<'PY'
import csv, glob, sys, collections
f = sys.argv[1]
agg = collections.defaultdict(float)
for p in glob.glob(f"/tmp/pa_{f}/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(p)):
        if "tc_frame_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
w = agg.get("SQ_WAVES", 0) or 1
print("flags %-9s per frame wavefront: VALU %7.0f  SALU %7.0f  LDS %6.0f   (%d wavefronts)" % (f, agg["SQ_INSTS_VALU"] / w, agg["SQ_INSTS_SALU"] / w, agg["SQ_INSTS_LDS"] / w, w))
PY
done
