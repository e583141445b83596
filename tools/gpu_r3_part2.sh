#!/bin/bash
# round 3 evidence, part 2: the bench lines -- default command per workload, the driver's 20-step command, the closed-loop
# entry point, a --preroll-ms 0 line, an on-road line -- the wrapper timings, and the phase clock / ablation tables
R=$GRAFT_REPO_ROOT
cd $R && mkdir -p gpurun_out/r3lines && O=gpurun_out/r3lines
line() { f=$1; shift; timeout -k 10 500 python bench.py "$@" > $O/$f.json 2> $O/$f.err || { tail -3 $O/$f.err; return 1; }; cut -c1-140 $O/$f.json; }
line cfg3 || exit 1
line cfg3_driver_style --steps 20 --warmup 5 || exit 1
line cfg3_single_step --steps-per-launch 0 --no-cpu-baseline || exit 1
line cfg3_preroll0 --preroll-ms 0 --no-cpu-baseline || exit 1
line cfg3_on_road --on-road --no-cpu-baseline || exit 1
line cfg2 --workload cfg2 || exit 1
line cfg4 --workload cfg4 || exit 1
line cfg5 --workload cfg5 --steps 64 --warmup 8 --preroll-ms 50 --cpu-budget 4000 || exit 1
timeout -k 10 600 python tools/bench_wrappers.py --steps 640 > $O/wrappers_bench.json 2> $O/wrappers_bench.err || { tail -5 $O/wrappers_bench.err; }
cat $O/wrappers_bench.json | cut -c1-600
