cd $GRAFT_REPO_ROOT; O=gpurun_out/s5; mkdir -p $O
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   %-50s %9.3f ms  %s' % (d['config'][:50], d['ms'], d.get('rowsum_max_rel_err')))"; }
for round in 1 2; do for op in "spgemm_group=0,spgemm_packed=0" "spgemm_group=1,spgemm_packed=0" "spgemm_group=1,spgemm_packed=1"; do
  echo "== $op"
  MI_BENCH_OPTS=$op timeout 300 python tools/bench_ops.py spgemm --no-order --reps 5 2>&1 | tail -1 | line
done; done 2>&1 | tee $O/ab.log
( cd /tmp && TMPDIR=/tmp timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o u -- python $GRAFT_REPO_ROOT/tools/bench_ops.py spgemm --no-order --reps 5 > $O/st.log 2>&1 ); cp $(find $O/st -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv; rm -rf $O/st
python - <<PY
import csv
for r in csv.DictReader(open("$O/kernel_stats.csv")):
    n=r["Name"]
    if "mi::" in n and float(r["AverageNs"])>20000: print("%-70s calls %3s avg %10.1f us" % (n[:70], r["Calls"], float(r["AverageNs"])/1e3))
PY
( time timeout 2400 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "spgemm or gram or sparse or golden or staged or determin" ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
