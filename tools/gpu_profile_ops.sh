cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider -k "hot_cold or golden" > gpurun_out/pytest_sub.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest_sub.log
for v in "" "--hot-kb 12288" "--ncols 256 --hot-kb 16384"; do
  echo "== variant $v"; timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu $v 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['hot_cold_tagged_gather'], d['config']['hot_column_coverage'])"
done 2>&1 | tee gpurun_out/variants3.log
cd /tmp && export TMPDIR=/tmp
for op in "spgemm" "gram --dense" "gram"; do
  tag=$(echo $op | tr -d ' -')
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -o $tag -- python $GRAFT_REPO_ROOT/tools/bench_ops.py $op --reps 2 > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1
  echo "== $op"; python3 - $GRAFT_REPO_ROOT/gpurun_out/prof_$tag <<'PY'
import csv,sys,glob
f=glob.glob(sys.argv[1]+'/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:14]:
    if 'at::' in r['Name'] or 'rocprim' in r['Name']: continue
    print("%-70s calls=%4s avg_us=%10.1f pct=%s" % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
done
