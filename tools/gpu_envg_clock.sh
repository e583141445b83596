#!/bin/bash
cd "$(dirname "$0")/.."
for g in 1 0; do for ch in 0 16; do
TC_CAND_GRID=$g TC_CHUNK=$ch timeout -k 10 300 python tools/phase_clock.py --envs 4096 --multi 64 --envg --steps 6 --bench-actions 2>&1 | grep -v "amdgpu.ids\|launch " | sed "s/^---/--- grid=$g/"
done; done
