# Round-4 session B: dense gram knock-outs on the final round-3 kernel (where do the 69 ms go?) + the new staged-product test
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04b; O=$GRAFT_REPO_ROOT/gpurun_out/r04b
( timeout 900 python -m pytest tests/test_gpu_staged_sypr.py -m gpu -q -x -p no:cacheprovider ) > $O/pytest_staged.log 2>&1; echo "staged rc=$?"; tail -3 $O/pytest_staged.log
g() { timeout 400 python tools/bench_ops.py gram --dense $@ 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   %-60s %9.3f ms' % (d['config'][:60], d['ms']))"; }
for tag in default nostore nowalk noatomic writeonly; do
  if [ $tag = default ]; then unset MI_SPARSE_RT; else export MI_SPARSE_RT=$GRAFT_REPO_ROOT/sparse_dot_amd/csrc/build/var/libmi_sparse_$tag.so; fi
  echo "== $tag"; g --cols 262144 --rows-log2 22 --reps 3
done 2>&1 | tee $O/gram_knockouts.log
