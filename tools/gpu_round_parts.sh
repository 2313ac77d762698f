cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "spgemm or sparse_sparse or gram or golden" > gpurun_out/pytest_parts.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_parts.log
for a in "spgemm --kind rmat --scale 17 --per-row 16 --lds-parts 0" "spgemm --kind rmat --scale 17 --per-row 16" "spgemm --kind rmat --scale 18 --per-row 16" "spgemm"; do
  echo "== ops $a"; timeout 900 python tools/bench_ops.py $a 2>&1 | tail -1 | cut -c1-600
done 2>&1 | tee gpurun_out/ops_parts.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_parts -- python $GRAFT_REPO_ROOT/tools/bench_ops.py spgemm --kind rmat --scale 17 --per-row 16 --reps 2 --no-order > /dev/null 2>&1
f=$(ls $GRAFT_REPO_ROOT/gpurun_out/prof_parts/*/*kernel_stats.csv | head -1); head -14 $f | cut -c1-200
