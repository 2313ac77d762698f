# Round-4 session K: 8-rank gloo dry run of bench.py (control flow of the N = 8 line incl. configs[4] at a small scale), 2-rank too
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04k; O=$GRAFT_REPO_ROOT/gpurun_out/r04k
export BENCH_DIST_BACKEND=gloo BENCH_CFG5_SCALE=14 HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 8 --steps 2 --warmup 1 --scale 16 --no-cpu --no-pmc ) > $O/bench_8rank_dryrun.log 2> $O/bench_8rank_dryrun.err; echo "8-rank rc=$?"
grep '^{' $O/bench_8rank_dryrun.log | tail -1 > $O/bench_8rank_line.json
python - <<'PY'
import json,os
d=json.load(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r04k/bench_8rank_line.json"))
print({k:d.get(k) for k in ('value','n_gpus','ms_per_step','compute_only_value','resident_B_value','scaling_basis')})
print('variants',json.dumps(d.get('variants'))[:700])
print('cfg5',json.dumps((d.get('secondary') or {}).get('spmm_config5_8gpu'))[:900])
PY
grep -n "Traceback" -A 12 $O/bench_8rank_dryrun.err | head -30
