#!/bin/bash
# round 2 evidence, part 2: the bench lines (default command per workload, the driver's 20-step command, the single-step
# entry point) and the wrapper timings; `traffic` comes from the PMC summaries committed by part 1
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for wl in cfg3 cfg2 cfg4; do
  timeout -k 10 400 python bench.py --workload $wl > gpurun_out/line_$wl.json 2> gpurun_out/line_$wl.err || { tail -3 gpurun_out/line_$wl.err; exit 1; }
  cut -c1-160 gpurun_out/line_$wl.json
done
timeout -k 10 500 python bench.py --workload cfg5 --steps 64 --warmup 8 --preroll-ms 50 --cpu-budget 4000 > gpurun_out/line_cfg5.json 2> gpurun_out/line_cfg5.err || { tail -3 gpurun_out/line_cfg5.err; exit 1; }
cut -c1-160 gpurun_out/line_cfg5.json
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/line_cfg3_driver_style.json 2>/dev/null || exit 1
cut -c1-160 gpurun_out/line_cfg3_driver_style.json
timeout -k 10 300 python bench.py --steps-per-launch 0 --no-cpu-baseline > gpurun_out/line_cfg3_single_step.json 2>/dev/null || exit 1
cut -c1-160 gpurun_out/line_cfg3_single_step.json
timeout -k 10 600 python tools/bench_wrappers.py --steps 640 > gpurun_out/wrappers_bench.json 2> gpurun_out/wrappers_bench.err || { tail -5 gpurun_out/wrappers_bench.err; exit 1; }
cat gpurun_out/wrappers_bench.json
