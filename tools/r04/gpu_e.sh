# Round-4 session E: cache policy of the gram write-out (does the output stream evict the lines of X the tiles of a row share?)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04f; O=$GRAFT_REPO_ROOT/gpurun_out/r04f
g() { timeout 400 python tools/bench_ops.py gram --dense $@ 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   %-60s %9.3f ms  %s' % (d['config'][:60], d['ms'], d.get('checks')))"; }
for tag in default ${VARIANTS}; do
  if [ $tag = default ]; then unset MI_SPARSE_RT; else export MI_SPARSE_RT=$GRAFT_REPO_ROOT/sparse_dot_amd/csrc/build/var/libmi_sparse_$tag.so; fi
  echo "== $tag"; g --cols 262144 --rows-log2 22 --reps 3
done 2>&1 | tee $O/gram_store_policy.log
