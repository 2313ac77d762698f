# Round-4 session L: gram with the per-XCD pull queue -- parity tests, then A/B against the fixed stride on one box (option gram_queue)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04l; O=$GRAFT_REPO_ROOT/gpurun_out/r04l
( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_baseline_configs.py -m gpu -q -x -p no:cacheprovider -k "gram or syrk or deterministic" ) > $O/pytest_gram.log 2>&1; echo "gram tests rc=$?"; tail -3 $O/pytest_gram.log
g() { timeout 400 python tools/bench_ops.py gram --dense $@ 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   %-60s %9.3f ms  %s' % (d['config'][:60], d['ms'], d.get('checks')))"; }
for round in 1 2; do for q in 1 0; do
  echo "== gram_queue=$q"; export MI_BENCH_OPTS=gram_queue=$q
  g --cols 262144 --rows-log2 22 --reps 3; g --cols 65536 --rows-log2 20 --reps 3
done; done 2>&1 | tee $O/gram_queue_ab.log
