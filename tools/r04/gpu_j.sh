# Round-4 session J: literal SpGEMM test (bounded host work), host-array call breakdown (stager on / off), 8-rank gloo dry run of bench.py
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04k; O=$GRAFT_REPO_ROOT/gpurun_out/r04k
( time timeout 1500 python -m pytest tests/test_gpu_baseline_configs.py -m gpu -q -x -p no:cacheprovider -k "literal_rmat" --durations=3 ) > $O/pytest.log 2>&1; echo "rc=$?"; tail -6 $O/pytest.log
timeout 600 python tools/gpu_api_overhead.py > $O/api_overhead.log 2>&1; grep "####\|== \|hipMemcpy\|mi_sparse_s_mm \|python" $O/api_overhead.log | head -40
export BENCH_DIST_BACKEND=gloo BENCH_CFG5_SCALE=14 HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 8 --steps 2 --warmup 1 --scale 16 --no-cpu --no-pmc ) > $O/bench_8rank_dryrun.log 2> $O/bench_8rank_dryrun.err; echo "8-rank rc=$?"
grep '^{' $O/bench_8rank_dryrun.log | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print({k:d.get(k) for k in ('value','n_gpus','ms_per_step','compute_only_value','resident_B_value','scaling_basis')})
print('variants',json.dumps(d.get('variants'))[:600])
print('cfg5',json.dumps((d.get('secondary') or {}).get('spmm_config5_8gpu'))[:900])
"
tail -3 $O/bench_8rank_dryrun.err
