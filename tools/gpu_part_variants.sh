cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "spgemm or sparse_sparse or gram or golden or order" 2>&1 | tail -2
run() { python tools/bench_ops.py $@ 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   %-50s %9.3f ms  order %8.3f  %s' % (d['config'][:50], d['ms'], d.get('order_ms', 0), d.get('rowsum_max_rel_err')))"; }
run spgemm --reps 5
run spgemm --kind rmat --scale 17 --per-row 16
run spgemm --kind rmat --scale 18 --per-row 16
run spgemm --kind rmat --scale 20 --per-row 16 --no-order --reps 2
run gram --reps 2
bash tools/gpu_prof_s20.sh 2>&1 | head -6
