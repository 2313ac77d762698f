# Round-3 session Y: the driver's bench line again (gram secondary fixed), then session X
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03y; O=$GRAFT_REPO_ROOT/gpurun_out/r03y
timeout 1500 python bench.py --steps 20 --warmup 3 > $O/bench.log 2> $O/bench.err; echo "bench rc=$?"; grep '^{' $O/bench.log | tail -1 > $O/bench_line.json
python - <<'PY'
import json,os
d=json.load(open(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r03y/bench_line.json"))
print(json.dumps({k:d[k] for k in ("value","ms_per_step")}))
for k,v in d.get("secondary",{}).items():
    print(k, json.dumps({kk:vv for kk,vv in v.items() if kk in ("ms","value","roofline","kernel","first_call_fresh_process","ms_per_step","error")})[:500])
PY
bash tools/gpu_r03x.sh
