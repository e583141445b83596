#!/bin/bash
# dev build (cfg3 kernels only): parity on simple_layout / stress map, then the frame kernel alone (TC_CHUNK=0) and the default bench
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_step_multi.py tests/test_gpu_parity.py -x -q -k "simple_layout-r64-classes or stress or against_oracle or without_rollout or no_observation or wrap or noise or (golden and simple_layout) or camera_sweep_frames and simple_layout or batch_sizes or per_env_cameras" > gpurun_out/t_dev.log 2>&1; rc=$?; tail -4 gpurun_out/t_dev.log; [ $rc -eq 0 ] || exit 1
TC_CHUNK=0 timeout -k 10 120 python bench.py --no-cpu-baseline --steps 256 --warmup 32 --steps-per-launch 32 --preroll-ms 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_us']; print('unpipelined: step %.2f us  sim %.2f  frame %.2f us/step' % (d['ms_per_step']*1e3, list(k.values())[0]/32, list(k.values())[-1]/32))"
timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg3 default', round(d['ms_per_step']*1e3,2), 'us/step', d['roofline']['kernels_us'], 'frac', round(d['roofline']['frac'],3))"
