# usage: bash tools/gpu_prof.sh <tag> [bench args] -> kernel stats + PMC means per kernel
R=$GRAFT_REPO_ROOT; tag=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag} -- python3 $R/bench.py --steps 300 --warmup 50 --no-cpu-baseline "$@" > $R/gpurun_out/prof_${tag}.log 2>&1
tail -1 $R/gpurun_out/prof_${tag}.log | cut -c1-400
grep -h "tc_" $R/gpurun_out/prof_${tag}/*/*_kernel_stats.csv | cut -c1-200
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_${tag}/p$i -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > $R/gpurun_out/pmc_${tag}_p$i.log 2>&1 || echo "pmc pass $i failed"
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/gpurun_out/pmc_${tag}/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if 'tc_' in r['Kernel_Name']:
            agg[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
for kn in agg:
    print(kn)
    n = agg[kn]['SQ_WAVES'][0] if agg[kn]['SQ_WAVES'] else 1
    for k in sorted(agg[kn]):
        v = agg[kn][k]; print(f"   {k:24s} per-launch {sum(v)/len(v):14.1f}   per-wave {sum(v)/len(v)/n:10.1f}")
PY
