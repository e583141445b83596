#!/bin/bash
# instruction-cache counters of the frame / simulate kernels, alone (TC_CHUNK=0) and co-running (pipelined chunks)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -E "ICACHE|IFETCH|INST_LEVEL|SQ_INSTS_VALU\b" | head -30 > $R/gpurun_out/icache_counters.txt
cat $R/gpurun_out/icache_counters.txt | cut -c1-160
for ch in 0 16; do
  rm -rf /tmp/ic_$ch
  TC_CHUNK=$ch timeout -k 10 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d /tmp/ic_$ch -- python3 $R/bench.py --steps 256 --warmup 128 --preroll-ms 0 --no-cpu-baseline > /tmp/ic_$ch.log 2>&1 || { echo "chunk $ch failed"; tail -5 /tmp/ic_$ch.log; continue; }
  python3 - "$ch" <<'PY'
import csv, glob, sys, collections
ch = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for p in glob.glob(f"/tmp/ic_{ch}/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(p)):
        k = "frame" if "tc_frame_kernel" in r["Kernel_Name"] else "sim" if ("tc_envg" in r["Kernel_Name"]) else None
        if k: agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in agg.items():
    print(f"TC_CHUNK={ch} {k}: " + "  ".join(f"{c}={v:.3g}" for c, v in sorted(d.items())), " miss rate %.3f" % (d["SQC_ICACHE_MISSES"] / max(d["SQC_ICACHE_REQ"], 1)))
PY
done
