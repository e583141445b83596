# usage: bash tools/gpu_variants.sh "<EXTRA flags 1>" "<EXTRA flags 2>" ...   (rebuilds on the GPU box and times cfg3)
cd $GRAFT_REPO_ROOT
for extra in "$@"; do
  make -C tinycarlo_amd/csrc clean >/dev/null
  make -C tinycarlo_amd/csrc EXTRA="$extra -Rpass-analysis=kernel-resource-usage" 2>&1 | grep -E "VGPRs:|Scratch|error" | tr '\n' ' '
  echo
  for w in cfg3; do timeout -k 10 200 python bench.py --workload $w --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  [$extra] $w', round(d['roofline']['kernel_us'],1),'us', round(d['value']/1e6,2),'M/s')"; done
done
