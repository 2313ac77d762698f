# Round-3 session Z: the committed bench line of the final tree (fresh-process first call measured before the parent's large allocations)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03z; O=$GRAFT_REPO_ROOT/gpurun_out/r03z
( time timeout 1800 python bench.py --steps 20 --warmup 3 ) > $O/bench.log 2> $O/bench.err; echo "bench rc=$?"; grep '^{' $O/bench.log | tail -1 > $O/bench_line.json
python - <<'PY'
import json,os
d=json.load(open(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r03z/bench_line.json"))
print(json.dumps({k:d[k] for k in ("value","ms_per_step","roofline","cpu_baseline")})[:1500])
for k,v in d.get("secondary",{}).items():
    print(k, json.dumps({kk:vv for kk,vv in v.items() if kk in ("ms","value","kernel","first_call_ms","first_call_fresh_process","ms_per_step","error")})[:500])
PY
tail -4 $O/bench.err
