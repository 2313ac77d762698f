cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for a in "spgemm --kind rmat --scale 17 --per-row 16" "spgemm --kind rmat --scale 18 --per-row 16" "spgemm --kind rmat --scale 20 --per-row 16 --no-order --reps 2" "spgemm --kind rmat --scale 20 --per-row 16 --no-order --reps 1 --lds-parts 0"; do
  echo "== ops $a"; timeout 900 python tools/bench_ops.py $a 2>&1 | tail -1 | cut -c1-700
done 2>&1 | tee gpurun_out/ops_parts2.log
