# Round-4 session F: A/B of gram variants on ONE box (interleaved, 2 rounds), plus the gram parity tests' durations
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04g; O=$GRAFT_REPO_ROOT/gpurun_out/r04g
g() { timeout 400 python tools/bench_ops.py gram --dense $@ 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   %-60s %9.3f ms  %s' % (d['config'][:60], d['ms'], d.get('checks')))"; }
for round in 1 2; do
for tag in default ${VARIANTS}; do
  if [ $tag = default ]; then unset MI_SPARSE_RT; else export MI_SPARSE_RT=$GRAFT_REPO_ROOT/sparse_dot_amd/csrc/build/var/libmi_sparse_$tag.so; fi
  echo "== $tag"; g --cols 262144 --rows-log2 22 --reps 3
done; done 2>&1 | tee $O/gram_ab.log
unset MI_SPARSE_RT
( timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "gram or syrk" --durations=6 ) > $O/pytest_gram.log 2>&1; tail -12 $O/pytest_gram.log
