cd $GRAFT_REPO_ROOT
run() { python tools/bench_ops.py $@ 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   %-50s %9.3f ms  %s' % (d['config'][:50], d['ms'], d.get('rowsum_max_rel_err')))"; }
for pipe in 1 0; do export MI_BENCH_OPTS=spgemm_pipe=$pipe; echo "== spgemm_pipe=$pipe"
  run spgemm --no-order --reps 5
  run spgemm --kind rmat --scale 18 --per-row 16 --no-order
  run spgemm --kind rmat --scale 20 --per-row 16 --no-order --reps 2
  run gram --reps 2
done
unset MI_BENCH_OPTS
MI_BENCH_OPTS=trace_phases=1 python tools/bench_ops.py spgemm --no-order --reps 2 2>&1 | grep -E "mi_sparse spgemm" | tail -4
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_staged_sypr.py -m gpu -q -x -p no:cacheprovider -k "spgemm or gram or staged or sypr or sparse_sparse" 2>&1 | tail -3
