cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o r01 -- python $R/bench.py --steps 20 --warmup 3 --no-cpu > $R/gpurun_out/prof.log 2>&1; echo "prof rc=$?"
python3 - $R/gpurun_out/prof <<'PY'
import csv,sys,glob
f=glob.glob(sys.argv[1]+'/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:6]:
    print("%-60s calls=%4s avg_us=%10.1f pct=%s" % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
cd $R
echo "== cfg5-like single GPU: R-MAT scale 24, N=256"
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu --scale 24 --ncols 256 2>&1 | tail -1 | cut -c1-900
echo "== scale 22 N=128"
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu --scale 22 2>&1 | tail -1 | cut -c1-700
