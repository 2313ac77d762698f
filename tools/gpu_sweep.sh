cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for v in "--hot-kb 6144" "--hot-kb 8192" "--hot-kb 10240" "--hot-kb 12288" "--hot-kb 8192 --unroll 8" "--hot-kb 10240 --chunk 128" "--hot-kb 8192 --chunk 512"; do
  echo "== variant $v"; timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu --no-secondary $v 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['hot_column_coverage'])"
done 2>&1 | tee gpurun_out/sweep.log
