# VALU/SALU/LDS instruction counts per env (summed over all tc_* kernels of a step) for each ablation variant
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for f in 0 0x1000 0x2000 0x4000 0x7000 0x100 0x200 0x300 0x400 0xC00; do
  TC_DEBUG_FLAGS=$f timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $R/gpurun_out/pmcph/f$f -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > /dev/null 2>&1
  python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("$R/gpurun_out/pmcph/f$f/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if 'tc_' in r['Kernel_Name'] and 'kernel' in r['Kernel_Name']:
            agg[r['Dispatch_Id']][r['Counter_Name']] += float(r['Counter_Value'])
# a step = consecutive tc_ dispatches; average per dispatch then per env, summing kernels by counting waves of the first
tot = collections.defaultdict(float); nd = 0
for d, c in agg.items():
    nd += 1
    for k, v in c.items(): tot[k] += v
waves = 4096.0
steps = 13.0 + 1  # reset + warmup + steps launches
import sys
nk = nd / steps
print("flags $f: kernels/step=%.1f" % nk, " ".join(f"{k[3:]}={v/steps/waves:.0f}" for k, v in sorted(tot.items()) if k != 'SQ_WAVES'))
PY
done
