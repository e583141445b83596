#!/bin/bash
# cfg5 at its full size (8192 envs, knuffingen, 480x640 rgb: 7.55 GB of observations per step): bench line, kernel stats, PMC
R=$GRAFT_REPO_ROOT; tag=r02_cfg5
cd /tmp && export TMPDIR=/tmp
A="--workload cfg5 --steps 64 --warmup 8 --preroll-ms 50"
timeout -k 10 400 python3 $R/bench.py $A --cpu-budget 4000 > $R/gpurun_out/bench_${tag}.json 2> $R/gpurun_out/bench_${tag}.err || { tail -3 $R/gpurun_out/bench_${tag}.err; exit 1; }
cut -c1-900 $R/gpurun_out/bench_${tag}.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag} -- python3 $R/bench.py $A --no-cpu-baseline > $R/gpurun_out/prof_${tag}.log 2>&1 || { tail -3 $R/gpurun_out/prof_${tag}.log; exit 1; }
grep -h "tc_" $R/gpurun_out/prof_${tag}/*/*_kernel_stats.csv | cut -c1-220
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_${tag}/p$i -- python3 $R/bench.py --workload cfg5 --steps 16 --warmup 4 --preroll-ms 0 --no-cpu-baseline > $R/gpurun_out/pmc_${tag}_p$i.log 2>&1 || echo "pmc pass $i failed"
done
