# Round-3 session P: dense gram with 16-byte LDS accesses in the tile flush; then the full suite
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03p; O=$GRAFT_REPO_ROOT/gpurun_out/r03p
g() { timeout 300 python tools/bench_ops.py gram --dense $@ 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   %-60s %9.3f ms  diag err %s' % (d['config'][:60], d['ms'], d.get('diag_max_rel_err')))"; }
for opts in gram_heads=1 gram_heads=0 gram_heads=1; do
  export MI_BENCH_OPTS=$opts; echo "== $opts"
  g --cols 262144 --rows-log2 22 --reps 3
done 2>&1 | tee $O/gram_flush_ab.log
unset MI_BENCH_OPTS
g --cols 65536 --rows-log2 20 --reps 3; g --cols 16384 --rows-log2 20 --reps 3
( time timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -2; grep -n "^E  \|FAILED" $O/pytest.log | head
