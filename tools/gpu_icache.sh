#!/bin/bash
# instruction-cache behaviour of the step kernel: which counters exist, then request / hit / miss counts at 64 and 4096 envs
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -E "icache|ifetch|SQ_INST_LEVEL|INSTS_ISSUED|SQ_WAIT_INST|INST_CYCLES" | cut -c1-160 | sort -u | head -30 > $R/gpurun_out/icache_counters.txt
cat $R/gpurun_out/icache_counters.txt
for n in 64 4096; do
  for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES"; do
    tag=$(echo $set | cut -c1-12 | tr ' ' '_')
    timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/icache_${n}_${tag} -- python3 $R/bench.py --envs $n --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/icache_${n}_${tag}.log 2>&1 || echo "pass failed: $n $set"
  done
done
python3 - <<PY
import csv, glob, collections
for n in (64, 4096):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"$R/gpurun_out/icache_{n}_*/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if 'tc_step_kernel' in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print(f"--- {n} envs (per launch means)")
    for k in sorted(agg):
        v = agg[k]; print(f"  {k:32s} {sum(v)/len(v):16.1f}   per wave {sum(v)/len(v)/n:12.1f}")
PY
