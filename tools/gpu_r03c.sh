# Round-3 session C: pipelined dense gram (A/B at the literal configs[3] and two smaller shapes), gram / SpMM tests, full suite,
# 2-rank dry run of the multi-GPU forms, allocation probe.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03c; O=$GRAFT_REPO_ROOT/gpurun_out/r03c
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_baseline_configs.py tests/test_gpu_reference_matrix.py -m gpu -q -x -p no:cacheprovider -k "gram or config4 or syrk" > $O/pytest_gram.log 2>&1; echo "pytest gram rc=$?"; tail -5 $O/pytest_gram.log
g() { timeout 300 python tools/bench_ops.py gram --dense $@ 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   %-60s %9.3f ms  diag err %s' % (d['config'][:60], d['ms'], d.get('diag_max_rel_err')))"; }
for opts in gram_sliced=1 gram_sliced=0; do
  export MI_BENCH_OPTS=$opts; echo "== $opts"
  g --cols 262144 --rows-log2 22 --reps 3
  g --cols 65536 --rows-log2 20 --reps 3
  g --cols 16384 --rows-log2 20 --reps 3
done 2>&1 | tee $O/gram_ab.log
unset MI_BENCH_OPTS
( time timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=12 ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -22 $O/pytest.log
timeout 300 tools/probes/alloc_probe > $O/alloc_probe.log 2>&1; echo "alloc rc=$?"; cat $O/alloc_probe.log
echo "== 2-rank dry run (gloo, both ranks on GPU 0)"
BENCH_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 > $O/bench_2rank_dry.log 2>&1; echo "rc=$?"; grep '^{' $O/bench_2rank_dry.log | tail -1 | cut -c1-3000; grep -v "^\[W\|amdgpu.ids" $O/bench_2rank_dry.log | tail -5 | cut -c1-300
