# dense gram: persistent workgroups (gram_persistent = workgroups per LDS slot; 0 = one workgroup per tile) x tile size
cd $GRAFT_REPO_ROOT
g() { timeout 300 python tools/bench_ops.py gram --dense $@ 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   %-60s %9.3f ms' % (d['config'][:60], d['ms']))"; }
for opts in gram_persistent=1 gram_persistent=2 gram_persistent=4; do
  export MI_BENCH_OPTS=$opts; echo "== $opts"
  g --cols 262144 --rows-log2 22 --reps 2
  g --cols 65536 --rows-log2 20 --reps 3
  g --reps 3
done
unset MI_BENCH_OPTS
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_baseline_configs.py -m gpu -q -x -p no:cacheprovider -k "gram or config4 or syrk" 2>&1 | tail -3
