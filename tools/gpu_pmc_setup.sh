R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for f in 0x7000 0x17000 0x27000 0x47000 0x67000 0x77000 0xF7000; do
  TC_DEBUG_FLAGS=$f timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/pmcph2/f$f -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
  python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(float)
for f in glob.glob("$R/gpurun_out/pmcph2/f$f/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if 'tc_' in r['Kernel_Name'] and 'kernel' in r['Kernel_Name']:
            tot[r['Counter_Name']] += float(r['Counter_Value'])
print("flags $f:", " ".join(f"{k[3:]}={v/14/4096:.0f}" for k, v in sorted(tot.items()) if k != 'SQ_WAVES'))
PY
done
