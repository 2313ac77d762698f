cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s9; mkdir -p $O
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   %-50s %9.3f ms  %s' % (d['config'][:50], d['ms'], d.get('rowsum_max_rel_err')))"; }
for round in 1 2; do for v in old default u2 u8; do
  unset MI_SPARSE_RT; [ $v = default ] || export MI_SPARSE_RT=$R/sparse_dot_amd/csrc/build/var/libmi_sparse_$v.so
  echo "== $v"
  timeout 300 python tools/bench_ops.py spgemm --kind rmat --scale 18 --per-row 16 --no-order 2>&1 | tail -1 | line
  timeout 300 python tools/bench_ops.py spgemm --kind rmat --scale 20 --per-row 16 --no-order --reps 2 2>&1 | tail -1 | line
done; done 2>&1 | tee $O/ab.log
