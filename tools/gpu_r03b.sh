# Round-3 session B: full GPU suite, bench line, 2-rank dry run of the new multi-GPU forms (gloo, one GPU), allocation probe.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03b; O=$GRAFT_REPO_ROOT/gpurun_out/r03b
( time timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=12 -x ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -22 $O/pytest.log
timeout 300 tools/probes/alloc_probe > $O/alloc_probe.log 2>&1; echo "alloc rc=$?"; cat $O/alloc_probe.log
echo "== 2-rank dry run (gloo, both ranks on GPU 0)"
BENCH_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 > $O/bench_2rank_dry.log 2>&1; echo "rc=$?"; grep '^{' $O/bench_2rank_dry.log | tail -1 | cut -c1-2500; tail -5 $O/bench_2rank_dry.log | cut -c1-300
( time timeout 1500 python bench.py --steps 20 --warmup 3 ) > $O/bench.log 2>&1; echo "bench rc=$?"; grep '^{' $O/bench.log | tail -1 | cut -c1-3500
