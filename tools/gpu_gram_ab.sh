cd $GRAFT_REPO_ROOT
for kb in 128 64; do echo "== gram_tile_kb=$kb"; MI_BENCH_OPTS=gram_tile_kb=$kb timeout 600 python tools/bench_ops.py gram --dense --cols 262144 --rows-log2 22 --reps 2 2>&1 | tail -1 | cut -c1-400; MI_BENCH_OPTS=gram_tile_kb=$kb timeout 600 python tools/bench_ops.py gram --dense --cols 65536 --rows-log2 20 --reps 2 2>&1 | tail -1 | cut -c1-300; done
timeout 600 python tools/bench_ops.py gram --dense --reps 3 2>&1 | tail -1 | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_baseline_configs.py -m gpu -q -x -p no:cacheprovider --durations=5 2>&1 | tail -12
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "gram" 2>&1 | tail -3
