cd /root/repo
timeout -k 10 300 python tools/loop_period.py --no-obs 2>&1 | grep -v amdgpu | tail -3 | cut -c1-400
timeout -k 10 300 python bench.py --workload cfg2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg2', round(d['ms_per_step']*1e3,2), 'us/step', d['roofline']['kernels_us'])"
timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg3', round(d['ms_per_step']*1e3,2), 'us/step', d['roofline']['kernels_us'])"
