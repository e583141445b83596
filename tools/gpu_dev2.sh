#!/bin/bash
cd "$(dirname "$0")/.."
timeout -k 10 300 python -m pytest tests/test_gpu_step_multi.py -x -q -k "simple_layout-r64-classes or stress or fused_terms or against_oracle or without_rollout or no_observation or wrap or noise" 2>&1 | tail -2
timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg3 default', round(d['ms_per_step']*1e3,2), 'us/step', d['roofline']['kernels_us'])"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg3 driver-style 20 steps', round(d['ms_per_step']*1e3,2), 'us/step', d['roofline']['kernels_us'], d['config']['preroll_steps'])"
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg3 100 steps', round(d['ms_per_step']*1e3,2), 'us/step')"
timeout -k 10 300 python bench.py --workload cfg2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg2', round(d['ms_per_step']*1e3,2), 'us/step', d['roofline']['kernels_us'])"
