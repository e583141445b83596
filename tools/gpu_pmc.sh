# usage: bash tools/gpu_pmc.sh <tag> [bench args...]   -> gpurun_out/pmc_<tag>/*.csv + printed per-launch means
R=$GRAFT_REPO_ROOT; tag=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_${tag}/p$i -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > $R/gpurun_out/pmc_${tag}_p$i.log 2>&1 || echo "pmc pass $i failed"
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$R/gpurun_out/pmc_${tag}/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if 'tc_env_kernel' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(agg):
    v = agg[k]; print(f"{k:24s} {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
