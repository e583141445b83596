#!/bin/bash
# per-phase shader clocks of one wavefront (instrumented build): single-step launches and K-step launches, 64 and 4096 envs
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 300 python tools/phase_clock.py --envs 64 4096 --steps 30 > gpurun_out/phase_clock_single.txt 2>&1 || { tail -5 gpurun_out/phase_clock_single.txt; exit 1; }
timeout -k 10 300 python tools/phase_clock.py --envs 64 4096 --steps 12 --multi 32 > gpurun_out/phase_clock_multi32.txt 2>&1 || { tail -5 gpurun_out/phase_clock_multi32.txt; exit 1; }
cat gpurun_out/phase_clock_multi32.txt
