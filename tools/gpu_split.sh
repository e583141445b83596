cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/gpu_tests.log
for sp in 1 2 3 4 8; do
  for w in cfg3 cfg4; do
  TC_SPLIT=$sp timeout -k 10 200 python bench.py --workload $w --steps 500 --warmup 50 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split=$sp $w', round(d['roofline']['step_us'],1),'us', round(d['value']/1e6,2),'M/s')"
  done
done
