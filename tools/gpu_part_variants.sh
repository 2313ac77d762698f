cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "spgemm or sparse_sparse or gram or golden or order" 2>&1 | tail -2
run() { python tools/bench_ops.py $@ 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   %-50s %9.3f ms' % (d['config'][:50], d['ms']))"; }
for bias in 0 1 -1; do
  echo "== log2s bias $bias"
  export MI_BENCH_OPTS="spgemm_part_log2s_bias=$bias"
  run spgemm --no-order --reps 3
  run spgemm --kind rmat --scale 17 --per-row 16 --no-order
  run spgemm --kind rmat --scale 18 --per-row 16 --no-order
  run spgemm --kind rmat --scale 20 --per-row 16 --no-order --reps 2
  run gram --reps 2
done 2>&1 | tee gpurun_out/part_variants.log
