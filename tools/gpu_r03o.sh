# Round-3 session O: dense gram with the slice bounds carried by the entries of X^T (gram_heads=1) vs the per-row table (=0), same box;
# tests; counters at the literal configs[3]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03o; O=$GRAFT_REPO_ROOT/gpurun_out/r03o
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_baseline_configs.py tests/test_gpu_reference_matrix.py tests/test_gpu_staged_sypr.py tests/test_gpu_bsr.py -m gpu -q -x -p no:cacheprovider -k "gram or config4 or syrk or sypr or bsr or config3" > $O/pytest_gram.log 2>&1; echo "pytest gram rc=$?"; tail -4 $O/pytest_gram.log
g() { timeout 300 python tools/bench_ops.py gram --dense $@ 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   %-60s %9.3f ms  diag err %s' % (d['config'][:60], d['ms'], d.get('diag_max_rel_err')))"; }
for opts in gram_heads=1 gram_heads=0 gram_heads=1 gram_heads=0; do
  export MI_BENCH_OPTS=$opts; echo "== $opts"
  g --cols 262144 --rows-log2 22 --reps 3
  g --cols 131072 --rows-log2 21 --reps 3
done 2>&1 | tee $O/gram_heads_ab.log
export MI_BENCH_OPTS=gram_heads=1,gram_sliced=2; echo "== $MI_BENCH_OPTS (sliced forced on the long-slice shapes)"; g --cols 65536 --rows-log2 20 --reps 3; g --cols 16384 --rows-log2 20 --reps 3
unset MI_BENCH_OPTS
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "WRITE_SIZE" "TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/gram/p$i -o p$i -- python $R/tools/bench_ops.py gram --dense --cols 262144 --rows-log2 22 --reps 1 > $O/gram_p$i.log 2>&1; echo "gram pass $i rc=$?"
done
python $R/tools/pmc_kernels.py $O/gram 2>&1 | grep -i "syrkd\|gram" | tee $O/pmc_gram_dense_kernels.jsonl | cut -c1-900
find $O -name "*.csv" -size +4M -delete
