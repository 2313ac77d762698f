cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02d; O=gpurun_out/r02d
for kb in 128 64; do echo "== gram_tile_kb=$kb"; MI_BENCH_OPTS=gram_tile_kb=$kb timeout 600 python tools/bench_ops.py gram --dense --cols 262144 --rows-log2 22 --reps 2 2>&1 | tail -1 | cut -c1-260; done
timeout 600 python tools/bench_ops.py gram --dense --cols 65536 --rows-log2 20 --reps 2 2>&1 | tail -1 | cut -c1-260
timeout 600 python tools/bench_ops.py gram --dense --reps 3 2>&1 | tail -1 | cut -c1-260
timeout 600 python tools/spmm_sweep.py --variants "2:8192:256,2:12288:256,2:16384:256,2:6144:256,2:8192:512,2:8192:128,2:16384:512" > $O/sweep_budget.log 2>&1; cut -c1-200 $O/sweep_budget.log
timeout 600 python tools/spmm_sweep.py --ncols 256 --variants "4:8192:256,4:16384:256,4:4096:256,2:8192:256,2:16384:256" > $O/sweep_budget_n256.log 2>&1; cut -c1-200 $O/sweep_budget_n256.log
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -m gpu -q -x -p no:cacheprovider -k "gram" 2>&1 | tail -3
