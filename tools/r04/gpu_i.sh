# Round-4 session I: deterministic option test, literal SpGEMM test timing, host-register probe, det cost
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04j; O=$GRAFT_REPO_ROOT/gpurun_out/r04j
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -m gpu -q -x -p no:cacheprovider -k "deterministic or literal_rmat" --durations=5 ) > $O/pytest.log 2>&1; echo "rc=$?"; tail -12 $O/pytest.log
timeout 300 python tools/probes/host_register_probe.py > $O/host_register_probe.log 2>&1; cat $O/host_register_probe.log | tail -6
for det in 0 1; do
  echo "== deterministic=$det"
  MI_BENCH_OPTS=deterministic=$det timeout 600 python tools/bench_ops.py spgemm --no-order --reps 2 2>&1 | tail -1 | cut -c1-260
  MI_BENCH_OPTS=deterministic=$det timeout 600 python tools/bench_ops.py spgemm --kind rmat --scale 16 --per-row 16 --no-order --reps 2 2>&1 | tail -1 | cut -c1-260
  MI_BENCH_OPTS=deterministic=$det timeout 600 python tools/bench_ops.py gram --dense --cols 16384 --rows-log2 20 --reps 2 2>&1 | tail -1 | cut -c1-260
done 2>&1 | tee $O/deterministic_cost.log
