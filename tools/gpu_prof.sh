#!/bin/bash
# usage: bash tools/gpu_prof.sh <tag> [bench args]   (through gpurun)
#   gpurun_out/prof_<tag>/  rocprofv3 --kernel-trace --stats of bench.py (the same command the bench line comes from)
#   gpurun_out/pmc_<tag>/p1..p4  separate --pmc passes (SQ counters x 2, FETCH_SIZE, WRITE_SIZE)
#   gpurun_out/bench_<tag>.json  the bench line of an un-profiled run
R=$GRAFT_REPO_ROOT; tag=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $R/bench.py --steps 512 --warmup 128 "$@" > $R/gpurun_out/bench_${tag}.json 2> $R/gpurun_out/bench_${tag}.err || { tail -3 $R/gpurun_out/bench_${tag}.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag} -- python3 $R/bench.py --steps 512 --warmup 128 --no-cpu-baseline "$@" > $R/gpurun_out/prof_${tag}.log 2>&1 || { tail -3 $R/gpurun_out/prof_${tag}.log; exit 1; }
grep -h "tc_" $R/gpurun_out/prof_${tag}/*/*_kernel_stats.csv | cut -c1-220
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_${tag}/p$i -- python3 $R/bench.py --steps 256 --warmup 128 --preroll-ms 0 --no-cpu-baseline "$@" > $R/gpurun_out/pmc_${tag}_p$i.log 2>&1 || echo "pmc pass $i failed"
done
cut -c1-700 $R/gpurun_out/bench_${tag}.json
