#!/bin/bash
# same-box A/B of library builds: tools/gpu_ab.sh "<bench args>" libA.so libB.so ...  (each build twice, interleaved)
R=$GRAFT_REPO_ROOT
cd $R
ARGS="$1"; shift
for rep in 1 2; do
  for lib in "$@"; do
    TINYCARLO_HIP_LIB=$R/tinycarlo_amd/$lib timeout -k 10 300 python bench.py $ARGS --no-cpu-baseline --no-single-step 2>/dev/null | python tools/ab_line.py $lib
  done
done
