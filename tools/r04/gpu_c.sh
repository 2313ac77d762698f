# Round-4 session C: dense gram after the pipeline rewrite -- parity tests that touch the gram kernels, then the literal configs[3] timing
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04c; O=$GRAFT_REPO_ROOT/gpurun_out/r04c
( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_reference_matrix.py -m gpu -q -x -p no:cacheprovider -k "gram or syrk" ) > $O/pytest_gram.log 2>&1; echo "gram tests rc=$?"; tail -3 $O/pytest_gram.log
g() { timeout 400 python tools/bench_ops.py gram --dense $@ 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   %-60s %9.3f ms  %s' % (d['config'][:60], d['ms'], d.get('checks')))"; }
for tag in default ${VARIANTS}; do
  if [ $tag = default ]; then unset MI_SPARSE_RT; else export MI_SPARSE_RT=$GRAFT_REPO_ROOT/sparse_dot_amd/csrc/build/var/libmi_sparse_$tag.so; fi
  echo "== $tag"; g --cols 262144 --rows-log2 22 --reps 3; g --cols 65536 --rows-log2 20 --reps 3; g --cols 16384 --rows-log2 20 --reps 3
done 2>&1 | tee $O/gram_times.log
