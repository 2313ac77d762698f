# Round-3 session I: full GPU suite + the driver's bench line (live PMC traffic, fresh-process first call) + kernel stats
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03i; O=$GRAFT_REPO_ROOT/gpurun_out/r03i
( time timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=8 ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -14 $O/pytest.log
( time timeout 1800 python bench.py --steps 20 --warmup 3 ) > $O/bench.log 2>&1; echo "bench rc=$?"; grep '^{' $O/bench.log | tail -1 > $O/bench_line.json; python - <<'PY'
import json,os
d=json.load(open(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r03i/bench_line.json"))
print(json.dumps({k:d[k] for k in ("value","ms_per_step","roofline")},indent=0)[:1500])
s=d.get("secondary",{})
for k,v in s.items():
    print(k, json.dumps(v)[:700])
PY
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o r03 -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu --no-secondary --no-pmc > $O/prof_bench.log 2>&1; echo "prof bench rc=$?"
f=$(find $O/prof_bench -name "*kernel_stats.csv" | head -1); head -6 $f | cut -c1-220; cp $f $O/bench_kernel_stats.csv
find $O -name "*.csv" -size +4M -delete
