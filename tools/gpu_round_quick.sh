cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest.log
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'], d['cpu_baseline']['value'], d.get('secondary'))"
for v in "--chunk 128" "--chunk 512" "--hot-kb 0" "--workload uniform" "--ncols 256"; do
  echo "== variant $v"; timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu $v 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['hot_cold_tagged_gather'])"
done 2>&1 | tee gpurun_out/variants.log
for a in "gram --dense" "gram --dense --cols 65536 --rows-log2 20" "spgemm"; do
  echo "== ops $a"; timeout 900 python tools/bench_ops.py $a 2>&1 | tail -1 | cut -c1-400
done 2>&1 | tee gpurun_out/ops2.log
