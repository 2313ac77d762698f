# Round-3 session R: dense gram, lane groups read their rows' bounds themselves (no ds_bpermute) vs the previous commit, same box; tests
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03r; O=$GRAFT_REPO_ROOT/gpurun_out/r03r
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_baseline_configs.py tests/test_gpu_reference_matrix.py -m gpu -q -x -p no:cacheprovider -k "gram or config4 or syrk" > $O/pytest_gram.log 2>&1; echo "pytest gram rc=$?"; tail -3 $O/pytest_gram.log
g() { timeout 300 python tools/bench_ops.py gram --dense $@ 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   %-60s %9.3f ms  diag err %s' % (d['config'][:60], d['ms'], d.get('diag_max_rel_err')))"; }
for tag in default prev default prev; do
  if [ $tag = default ]; then unset MI_SPARSE_RT; else export MI_SPARSE_RT=$GRAFT_REPO_ROOT/sparse_dot_amd/csrc/build/var/libmi_sparse_$tag.so; fi
  echo "== $tag"; g --cols 262144 --rows-log2 22 --reps 3; g --cols 131072 --rows-log2 21 --reps 3
done 2>&1 | tee $O/gram_nobpermute_ab.log
