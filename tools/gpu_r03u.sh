# Round-3 session U: dense gram -- slice bounds decoded in registers (the 16-byte record no longer sits in an LDS-promoted
# alloca) and 152 KiB tiles (7 instead of 8 per output row at configs[3]); A/B against the previous commit on one box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03u; O=$GRAFT_REPO_ROOT/gpurun_out/r03u
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_baseline_configs.py tests/test_gpu_reference_matrix.py -m gpu -q -x -p no:cacheprovider -k "gram or config4 or syrk" > $O/pytest_gram.log 2>&1; echo "pytest gram rc=$?"; tail -3 $O/pytest_gram.log
g() { timeout 300 python tools/bench_ops.py gram --dense $@ 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   %-60s %9.3f ms  diag err %s' % (d['config'][:60], d['ms'], d.get('diag_max_rel_err')))"; }
for tag in new152 new128 prev new152 new128 prev; do
  unset MI_SPARSE_RT MI_BENCH_OPTS
  if [ $tag = prev ]; then export MI_SPARSE_RT=$GRAFT_REPO_ROOT/sparse_dot_amd/csrc/build/var/libmi_sparse_prev.so; fi
  if [ $tag = new128 ]; then export MI_BENCH_OPTS=gram_tile_kb=128; fi
  echo "== $tag"; g --cols 262144 --rows-log2 22 --reps 3; g --cols 131072 --rows-log2 21 --reps 3; g --cols 65536 --rows-log2 20 --reps 3
done 2>&1 | tee $O/gram_regs_tile152_ab.log
