cd /tmp && export TMPDIR=/tmp
for spec in "s20:spgemm --kind rmat --scale 20 --per-row 16 --reps 1 --no-order" "s18:spgemm --kind rmat --scale 18 --per-row 16 --reps 1 --no-order" "gram:gram --reps 1"; do
  tag=${spec%%:*}; a=${spec#*:}
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -- python $GRAFT_REPO_ROOT/tools/bench_ops.py $a > /dev/null 2>&1
  f=$(ls $GRAFT_REPO_ROOT/gpurun_out/prof_$tag/*/*kernel_stats.csv | head -1)
  echo "== $tag"; python - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:9]:
    print("  %-62s calls %3s avg %10.3f ms  %5s%%" % (r["Name"][:62], r["Calls"], float(r["AverageNs"])/1e6, r["Percentage"]))
PY
  rm -f $GRAFT_REPO_ROOT/gpurun_out/prof_$tag/*/*kernel_trace.csv
done
