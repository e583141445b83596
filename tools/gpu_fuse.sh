cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; echo "pytest(fused default) rc=$?"; tail -2 gpurun_out/gpu_tests.log
for fu in 0 1; do
  for w in cfg3 cfg2 cfg4; do
  TC_FUSE=$fu timeout -k 10 200 python bench.py --workload $w --steps 500 --warmup 50 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fuse=$fu $w', round(d['roofline']['step_us'],1),'us', round(d['value']/1e6,2),'M/s')"
  done
done
for n in 2048 8192 16384; do TC_FUSE=1 python bench.py --envs $n --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused envs=$n', round(d['roofline']['step_us'],1),'us', round(d['value']/1e6,2),'M/s')"; done
