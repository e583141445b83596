#!/bin/bash
# kernel timeline of one streamed 20-step call (and of one 128-step call): rocprofv3 --kernel-trace, last call of the run
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/stream_trace; mkdir -p $O
cd $R
rocprofv3 --kernel-trace --output-format csv -d $O/d20 -- python3 bench.py --steps 20 --warmup 5 --no-single-step --no-cpu-baseline > $O/d20.json 2> $O/d20.err || { tail -5 $O/d20.err; exit 1; }
rocprofv3 --kernel-trace --output-format csv -d $O/d128 -- python3 bench.py --steps 128 --warmup 128 --no-single-step --no-cpu-baseline > $O/d128.json 2> $O/d128.err || { tail -5 $O/d128.err; exit 1; }
python3 tools/stream_timeline.py $O/d20 && python3 tools/stream_timeline.py $O/d128
