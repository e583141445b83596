# VALU/SALU/LDS instruction counts per env for each ablation variant (phase budget)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for f in 0 0x100 0x200 0x300 0x400 0xC00; do
  TC_DEBUG_FLAGS=$f timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD --output-format csv -d $R/gpurun_out/pmcph/f$f -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > /dev/null 2>&1
  python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$R/gpurun_out/pmcph/f$f/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if 'tc_env_kernel' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
n = agg['SQ_WAVES'][0]
print("flags $f:", " ".join(f"{k[3:]}={sum(v)/len(v)/n:.0f}" for k, v in sorted(agg.items()) if k != 'SQ_WAVES'))
PY
done
