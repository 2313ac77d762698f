cd $GRAFT_REPO_ROOT
run() { python tools/bench_ops.py $@ 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   %-40s %9.3f ms  rowsum %.3g  nnzC %d order %.1f' % (d['config'][:40], d['ms'], d['rowsum_max_rel_err'], d['nnzC'], d['order_ms']))"; }
run spgemm --kind rmat --scale 20 --per-row 16 --no-order --reps 2
MI_BENCH_OPTS=spgemm_slice_table=0 run spgemm --kind rmat --scale 20 --per-row 16 --no-order --reps 1
run spgemm --kind rmat --scale 17 --per-row 16
run spgemm --kind rmat --scale 18 --per-row 16
run spgemm
timeout 1700 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest.log
