cd $GRAFT_REPO_ROOT
export MI_SPARSE_RT=$GRAFT_REPO_ROOT/sparse_dot_amd/csrc/build/var/libmi_sparse_gt.so
for opts in gram_sliced=1 gram_sliced=1,gram_persistent=1; do
  for shape in "--cols 262144 --rows-log2 22 --reps 1" "--cols 65536 --rows-log2 20 --reps 1" "--reps 1"; do
    echo "== $opts $shape"
    MI_BENCH_OPTS=$opts timeout 300 python tools/bench_ops.py gram --dense $shape 2>&1 | grep -E "gram timing|^\{" | cut -c1-260 | tail -4
  done
done
unset MI_SPARSE_RT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "multi_tile" 2>&1 | tail -5
