# Round-3 session F: dense gram, same box: round-2 library vs the pipelined kernel and its knock-outs
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03f; O=$GRAFT_REPO_ROOT/gpurun_out/r03f
g() { timeout 300 python tools/bench_ops.py gram --dense $@ 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   %-60s %9.3f ms' % (d['config'][:60], d['ms']))"; }
for tag in r02 default slicesnow nostore noacc noboth r02 default; do
  if [ $tag = default ]; then unset MI_SPARSE_RT; else export MI_SPARSE_RT=$GRAFT_REPO_ROOT/sparse_dot_amd/csrc/build/var/libmi_sparse_$tag.so; fi
  echo "== $tag"; g --cols 262144 --rows-log2 22 --reps 3
done 2>&1 | tee $O/gram_variants.log
