cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
g() { timeout 300 python tools/bench_ops.py gram --dense $@ 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   %-60s %9.3f ms' % (d['config'][:60], d['ms']))"; }
for tag in default sub16 sub4; do
  if [ $tag = default ]; then unset MI_SPARSE_RT; else export MI_SPARSE_RT=$GRAFT_REPO_ROOT/sparse_dot_amd/csrc/build/var/libmi_sparse_$tag.so; fi
  for opts in gram_sliced=1 gram_sliced=2; do
  echo "== $tag $opts"; export MI_BENCH_OPTS=$opts; g --cols 262144 --rows-log2 22 --reps 2; g --cols 65536 --rows-log2 20 --reps 3; g --reps 3
  done
done 2>&1 | tee gpurun_out/gram_variants.log
unset MI_SPARSE_RT MI_BENCH_OPTS
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_baseline_configs.py -m gpu -q -x -p no:cacheprovider -k "gram or config4 or syrk" 2>&1 | tail -3
