#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 300 python tools/phase_clock.py --envs 64 1024 4096 16384 --steps 12 --multi 32 > gpurun_out/phase_clock_multi32.txt 2>&1 || { tail -5 gpurun_out/phase_clock_multi32.txt; exit 1; }
grep -E "^---|shader clock" gpurun_out/phase_clock_multi32.txt
