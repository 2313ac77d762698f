# tests + bench + SpGEMM/gram op benches (no PMC passes)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"
timeout 1700 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/pytest.log
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench.log | cut -c1-1500
for a in "spgemm" "spgemm --kind rmat --scale 17 --per-row 16" "spgemm --kind rmat --scale 18 --per-row 16" "spgemm --kind rmat --scale 20 --per-row 16 --no-order --reps 2" "gram --dense" "gram" "gram --dense --cols 65536 --rows-log2 20"; do
  echo "== ops $a"; timeout 900 python tools/bench_ops.py $a 2>&1 | tail -1 | cut -c1-700
done 2>&1 | tee gpurun_out/ops.log
