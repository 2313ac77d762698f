cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest.log
for v in "" "--hot-kb 0" "--hot-kb 1024" "--hot-kb 2048" "--hot-kb 4096" "--hot-kb 6144" "--hot-kb 8192" "--hot-kb 16384" "--chunk 1024" "--workload uniform" "--ncols 256" ; do
  echo "== variant $v"; timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu $v 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['hot_cold_tagged_gather'], d['config']['hot_column_coverage'])"
done 2>&1 | tee gpurun_out/variants2.log
