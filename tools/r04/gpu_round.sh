# Round-4 full session: GPU test suite, bench.py (driver form), rocprof kernel stats of the bench
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04r; O=$GRAFT_REPO_ROOT/gpurun_out/r04r
( time timeout 2400 python -m pytest tests -m gpu -q -x -p no:cacheprovider --durations=10 ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -16 $O/pytest_gpu.log
( time timeout 1800 python bench.py --steps 20 --warmup 5 ) > $O/bench.log 2> $O/bench.err; echo "bench rc=$?"; grep '^{' $O/bench.log | tail -1 > $O/bench_line.json
python - <<'PY'
import json,os
d=json.load(open(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r04r/bench_line.json"))
print(json.dumps({k:d[k] for k in ("value","ms_per_step","roofline","cpu_baseline")})[:1500])
for k,v in d.get("secondary",{}).items():
    print(k, json.dumps({kk:vv for kk,vv in v.items() if kk in ("ms","value","kernel","first_call_ms","ms_per_step","error","roofline","gpu_on_cpu_sample_shape","cpu_baseline")})[:1400])
PY
tail -5 $O/bench.err
