cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "spgemm or sparse_sparse or golden or gram" 2>&1 | tail -2
for m in 0 1; do
  echo "== global mode $m"
  timeout 600 python tools/bench_ops.py spgemm --kind rmat --scale 17 --per-row 16 --global-mode $m --no-order 2>&1 | tail -1 | cut -c1-300
  timeout 600 python tools/bench_ops.py spgemm --kind rmat --scale 18 --per-row 16 --global-mode $m --no-order 2>&1 | tail -1 | cut -c1-300
done
