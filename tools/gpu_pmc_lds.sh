#!/bin/bash
# LDS bank conflicts by phase: rocprofv3 --pmc on single-step launches of cfg3 with parts of the kernel switched off
# (TC_DEBUG_FLAGS ablations; results of those runs are wrong on purpose, only the counters matter).
# usage: bash tools/gpu_pmc_lds.sh <out.txt>   (run through gpurun)
R=$GRAFT_REPO_ROOT
OUT=${1:-$R/gpurun_out/pmc_lds.txt}
cd /tmp && export TMPDIR=/tmp
: > $OUT
for f in 0 0xC00 0x400 0x100 0x200 0x1000 0x2000 0x4000 0x7000 0x10000 0x20000; do
  rm -rf $R/gpurun_out/pmclds/f$f
  TC_DEBUG_FLAGS=$f timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/pmclds/f$f -- python3 $R/bench.py --steps 12 --warmup 4 --steps-per-launch 0 --preroll-ms 0 --no-cpu-baseline > /dev/null 2>&1 || echo "flags $f: run failed" >> $OUT
  python3 - >> $OUT <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$R/gpurun_out/pmclds/f$f/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if 'tc_step_kernel' in r['Kernel_Name'] or 'tc_env_kernel' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
if agg:
    n = agg['SQ_WAVES'][0]
    names = {"0": "full step", "0xC00": "phase A only (no distances, no camera)", "0x400": "A + B (no camera, no raster)", "0x100": "no raster pixel work (setup skipped too)",
             "0x200": "no expand/store", "0x1000": "no outline pixels", "0x2000": "no fill rows", "0x4000": "no caps", "0x7000": "no pixel work at all",
             "0x10000": "no fill events", "0x20000": "no outline clip/DDA setup"}
    print("flags %-8s %-42s" % ("$f", names.get("$f", "")), " ".join(f"{k[3:]}={sum(v)/len(v)/n:.0f}" for k, v in sorted(agg.items()) if k != 'SQ_WAVES'))
PY
done
cat $OUT
