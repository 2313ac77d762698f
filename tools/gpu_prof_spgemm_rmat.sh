cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_spgemm_rmat -o sr -- python $R/tools/bench_ops.py spgemm --kind rmat --scale 17 --per-row 16 --reps 2 > $R/gpurun_out/prof_spgemm_rmat.log 2>&1
python3 - $R/gpurun_out/prof_spgemm_rmat <<'PY'
import csv,sys,glob
f=glob.glob(sys.argv[1]+'/*kernel_stats.csv')[0]
n=0
for r in csv.DictReader(open(f)):
    if 'at::' in r['Name'] or 'rocprim' in r['Name']: continue
    print("%-64s calls=%4s avg_us=%10.1f tot_ms=%8.1f" % (r['Name'][:64], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6)); n+=1
    if n>=12: break
PY
