# One GPU-box session: parity tests, bench, secondary op benches, multi-rank dry run, rocprof stats, PMC passes.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"
timeout 1700 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/pytest.log
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench.log | cut -c1-1500
for v in "--hot-kb 0" "--workload uniform" "--ncols 256" "--ncols 64" "--chunk 512" "POOL0"; do
  echo "== variant $v"; if [ "$v" = POOL0 ]; then export MI_BENCH_OPTS=pool_enable=0; v=""; else unset MI_BENCH_OPTS; fi; timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu $v 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['hot_cold_tagged_gather'])"
done 2>&1 | tee gpurun_out/variants.log
echo "== 2-rank dry run (gloo, both ranks on GPU 0)"
BENCH_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/bench_2rank_dry.log 2>&1; echo "rc=$?"; tail -1 gpurun_out/bench_2rank_dry.log | cut -c1-300
unset MI_BENCH_OPTS
for a in "spgemm" "spgemm --kind rmat --scale 17 --per-row 16" "spgemm --kind rmat --scale 18 --per-row 16" "spgemm --kind rmat --scale 20 --per-row 16 --no-order --reps 2" "gram --dense" "gram --dense --cols 65536 --rows-log2 20" "gram" ; do
  echo "== ops $a"; timeout 900 python tools/bench_ops.py $a 2>&1 | tail -1 | cut -c1-500
done 2>&1 | tee gpurun_out/ops.log
python tools/gpu_spmv.py 2>&1 | grep "SpMV" | tee gpurun_out/spmv.log
cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu --no-secondary > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT; rm -rf gpurun_out/pmc; BENCH_EXTRA="--no-secondary" bash tools/prof_pmc.sh 2>&1 | tail -9
