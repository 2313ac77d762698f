# Round-3 session V: gram head decode, 64-bit shift (default build) vs masks, 152 KiB tiles, one box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03v; O=$GRAFT_REPO_ROOT/gpurun_out/r03v
g() { timeout 300 python tools/bench_ops.py gram --dense $@ 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   %-60s %9.3f ms  diag err %s' % (d['config'][:60], d['ms'], d.get('diag_max_rel_err')))"; }
for tag in shift masks shift masks; do
  unset MI_SPARSE_RT MI_BENCH_OPTS
  if [ $tag = masks ]; then export MI_SPARSE_RT=$GRAFT_REPO_ROOT/sparse_dot_amd/csrc/build/var/libmi_sparse_masks.so; fi
  echo "== $tag"; g --cols 262144 --rows-log2 22 --reps 3; MI_BENCH_OPTS=gram_tile_kb=128 g --cols 262144 --rows-log2 22 --reps 3
done 2>&1 | tee $O/gram_decode_ab.log
