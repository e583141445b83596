#!/bin/bash
# Where a frame wavefront waits: mean latencies of instruction fetch, scalar memory and LDS from the SQ level counters
# (accumulated occupancy of the queue / number of requests), plus the instruction-cache hit rate, for the kernels of the
# default bench command, pipelined (TC_CHUNK=16: frames beside the simulate kernel) and not (TC_CHUNK=0).
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for ch in ${CHUNKS:-16 0}; do
  i=0
  for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVES" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVES SQ_WAVE_CYCLES" \
             "SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_WAVES SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" \
             "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_WAIT_ANY"; do
    i=$((i+1))
    rm -rf /tmp/lp_${ch}_$i
    TC_CHUNK=$ch timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d /tmp/lp_${ch}_$i -- python3 $R/bench.py --steps 128 --warmup 1024 --preroll-ms 0 --no-cpu-baseline --no-single-step > /tmp/lp_${ch}_$i.log 2>&1 || { echo "chunk $ch set $i failed"; tail -3 /tmp/lp_${ch}_$i.log; }
  done
  python3 - "$ch" <<'PY'
import csv, glob, sys, collections
ch = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for p in glob.glob(f"/tmp/lp_{ch}_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(p)):
        k = "frame" if "tc_frame_kernel" in r["Kernel_Name"] else "sim" if ("tc_envg" in r["Kernel_Name"]) else None
        if k: agg[k][r["Counter_Name"]] = agg[k][r["Counter_Name"]] + float(r["Counter_Value"])
for k, d in sorted(agg.items()):
    w = max(d.get("SQ_WAVES", 0) / 4, 1)   # SQ_WAVES was collected in 4 of the passes... per pass value: each pass sums all dispatches
    g = lambda c: d.get(c, 0.0)
    print(f"TC_CHUNK={ch} {k}:")
    print("   icache: req/wave %.0f  hit rate %.5f  misses/wave %.2f  duplicate misses/wave %.2f" % (g("SQC_ICACHE_REQ") / w, g("SQC_ICACHE_HITS") / max(g("SQC_ICACHE_REQ"), 1), g("SQC_ICACHE_MISSES") / w, g("SQC_ICACHE_MISSES_DUPLICATE") / w))
    print("   ifetch: %.0f per wave, mean latency %.0f clk (level/requests)" % (g("SQ_IFETCH") / w, g("SQ_IFETCH_LEVEL") / max(g("SQ_IFETCH"), 1)))
    print("   smem:   %.0f per wave, mean latency %.0f clk" % (g("SQ_INSTS_SMEM") / w, g("SQ_INST_LEVEL_SMEM") / max(g("SQ_INSTS_SMEM"), 1)))
    print("   lds:    %.0f per wave, mean latency %.0f clk; wave-cycles waiting for LDS %.0f, for any instruction %.0f per wave" % (g("SQ_INSTS_LDS") / w, g("SQ_INST_LEVEL_LDS") / max(g("SQ_INSTS_LDS"), 1), g("SQ_WAIT_INST_LDS") / w, g("SQ_WAIT_INST_ANY") / w))
    print("   valu:   %.0f per wave, active cycles %.0f per wave; wave cycles %.0f per wave, waiting (any) %.0f" % (g("SQ_INSTS_VALU") / w, g("SQ_ACTIVE_INST_VALU") / w, g("SQ_WAVE_CYCLES") / w, g("SQ_WAIT_ANY") / w))
PY
done
