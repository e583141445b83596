#!/bin/bash
# round 3 evidence, part 1: GPU suite, then per workload the bench line, rocprofv3 --kernel-trace --stats of the same command
# and the separate --pmc passes (SQ counters x 2, FETCH_SIZE, WRITE_SIZE).  Summarise with tools/summarize_pmc.py
# (--round r03), commit, then run tools/gpu_r3_part2.sh for the lines that quote the PMC traffic.
R=$GRAFT_REPO_ROOT
cd $R && mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; rc=$?; tail -3 gpurun_out/gpu_tests.log; [ $rc -eq 0 ] || exit 1
cd /tmp && export TMPDIR=/tmp
prof() {  # tag, bench args of the line / kernel stats, bench args of the PMC passes
  tag=$1; A="$2"; P="$3"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag} -- python3 $R/bench.py $A --no-cpu-baseline --no-single-step > $R/gpurun_out/prof_${tag}.log 2>&1 || { tail -3 $R/gpurun_out/prof_${tag}.log; return 1; }
  grep -h "tc_" $R/gpurun_out/prof_${tag}/*/*_kernel_stats.csv | cut -c1-200
  i=0
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout -k 10 400 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_${tag}/p$i -- python3 $R/bench.py $P --no-cpu-baseline --no-single-step > $R/gpurun_out/pmc_${tag}_p$i.log 2>&1 || echo "pmc pass $i of $tag failed"
  done
}
prof r03_cfg3 "" "--steps 256 --warmup 128" || exit 1
prof r03_cfg3_driver "--steps 20 --warmup 5" "--steps 20 --warmup 5" || exit 1
prof r03_cfg2 "--workload cfg2" "--workload cfg2 --steps 256 --warmup 128" || exit 1
prof r03_cfg4 "--workload cfg4" "--workload cfg4 --steps 256 --warmup 128" || exit 1
prof r03_cfg5 "--workload cfg5 --steps 64 --warmup 8 --preroll-ms 50" "--workload cfg5 --steps 16 --warmup 4 --preroll-ms 0" || exit 1
