#!/bin/bash
# round 2 evidence: kernel stats + PMC + bench lines for cfg3, cfg2, cfg4 (cfg5 has its own script: it needs 70 GB)
cd "$(dirname "$0")/.."
for wl in cfg3 cfg2 cfg4; do
  bash tools/gpu_prof.sh r02_$wl --workload $wl || exit 1
done
