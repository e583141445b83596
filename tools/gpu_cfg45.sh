cd $GRAFT_REPO_ROOT
timeout -k 10 200 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/gpu_tests.log
for bb in 16384 32768 65536; do
 TC_BAND_BYTES=$bb timeout -k 10 400 python bench.py --workload cfg5 --envs 2048 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('cfg5(2048 envs) band=$bb', round(r['step_us'],1),'us', round(d['value']/1e6,3),'M/s step GB/s', round(r['step_achieved_GBs'],1), r['kernels_us'], d['config']['lds_bytes_per_env'])"
done
for bb in 16384 32768; do
TC_BAND_BYTES=$bb timeout -k 10 200 python bench.py --workload cfg4 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('cfg4 band=$bb', round(r['step_us'],1),'us', round(d['value']/1e6,3),'M/s step GB/s', round(r['step_achieved_GBs'],1), r['kernels_us'], d['config']['lds_bytes_per_env'])"
done
timeout -k 10 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('cfg3', round(r['step_us'],1),'us', round(d['value']/1e6,3),'M/s', r)"
