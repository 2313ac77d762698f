cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
g() { timeout 300 python tools/bench_ops.py gram --dense $@ 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   %-60s %9.3f ms' % (d['config'][:60], d['ms']))"; }
for tag in default nowrite nowalk neither; do
  if [ $tag = default ]; then unset MI_SPARSE_RT; else export MI_SPARSE_RT=$GRAFT_REPO_ROOT/sparse_dot_amd/csrc/build/var/libmi_sparse_$tag.so; fi
  echo "== $tag"; g --cols 262144 --rows-log2 22 --reps 2
done 2>&1 | tee gpurun_out/gram_variants.log
