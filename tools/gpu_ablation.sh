cd $GRAFT_REPO_ROOT
timeout -k 10 200 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/gpu_tests.log
for f in 0 0x100 0x200 0x300 0x400 0xC00; do
  TC_DEBUG_FLAGS=$f timeout -k 10 120 python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$f', round(d['roofline']['kernel_us'],1),'us', round(d['value']/1e6,2),'M/s')"
done
for w in cfg2 cfg4; do timeout -k 10 200 python bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', round(d['roofline']['kernel_us'],1),'us', round(d['value']/1e6,2),'M/s', round(d['roofline']['frac'],4))"; done
timeout -k 10 400 python bench.py --workload cfg5 --envs 1024 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5(1024 envs)', round(d['roofline']['kernel_us'],1),'us', round(d['value']/1e6,3),'M/s', round(d['roofline']['frac'],4))"
