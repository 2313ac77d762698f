cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "spgemm or sparse_sparse or order or golden or gram" > gpurun_out/pytest_sp.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest_sp.log
for a in "spgemm" "spgemm --kind rmat --scale 17 --per-row 16" "spgemm --kind rmat --scale 18 --per-row 16"; do
  echo "== ops $a"; timeout 900 python tools/bench_ops.py $a 2>&1 | tail -1 | cut -c1-500
done 2>&1 | tee gpurun_out/ops3.log
bash tools/gpu_prof_spgemm_rmat.sh
