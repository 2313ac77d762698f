# dense gram, sliced walk: cluster queue (gram_cluster workgroups of one XCD pull the tiles of their rows) vs static order
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_baseline_configs.py tests/test_gpu_reference_matrix.py -m gpu -q -x -p no:cacheprovider -k "gram or config4 or syrk" 2>&1 | tail -5
g() { timeout 300 python tools/bench_ops.py gram --dense $@ 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   %-60s %9.3f ms  diag err %s' % (d['config'][:60], d['ms'], d.get('diag_max_rel_err')))"; }
for opts in gram_cluster=0 gram_cluster=8 gram_cluster=16 gram_cluster=8,gram_sliced=2; do
  export MI_BENCH_OPTS=$opts; echo "== $opts"
  g --cols 262144 --rows-log2 22 --reps 2
  g --cols 65536 --rows-log2 20 --reps 3
done
