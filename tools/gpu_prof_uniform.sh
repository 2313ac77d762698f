cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_uni -- python $GRAFT_REPO_ROOT/tools/bench_ops.py spgemm --reps 2 > /dev/null 2>&1
f=$(ls $GRAFT_REPO_ROOT/gpurun_out/prof_uni/*/*kernel_stats.csv | head -1)
python - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:22]:
    print("  %-70s calls %3s avg %9.3f ms  %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e6, r["Percentage"]))
PY
rm -f $GRAFT_REPO_ROOT/gpurun_out/prof_uni/*/*kernel_trace.csv
