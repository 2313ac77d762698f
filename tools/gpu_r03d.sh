# Round-3 session D: pipelined gram -- 64 KiB tiles (two workgroups per CU: one walks while the other writes out) vs 128 KiB
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03d; O=$GRAFT_REPO_ROOT/gpurun_out/r03d
g() { timeout 300 python tools/bench_ops.py gram --dense $@ 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   %-60s %9.3f ms  diag err %s' % (d['config'][:60], d['ms'], d.get('diag_max_rel_err')))"; }
for opts in gram_tile_kb=64 gram_tile_kb=64,gram_persistent=2 gram_tile_kb=128,gram_persistent=2; do
  export MI_BENCH_OPTS=$opts; echo "== $opts"
  g --cols 262144 --rows-log2 22 --reps 3
  g --cols 65536 --rows-log2 20 --reps 3
done 2>&1 | tee $O/gram_ab2.log
