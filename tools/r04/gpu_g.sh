# Round-4 session G: SpGEMM variants A/B on one box (interleaved), literal R-MAT configs[2], R-MAT 2^18, uniform
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04h; O=$GRAFT_REPO_ROOT/gpurun_out/r04h
run() { timeout 300 python tools/bench_ops.py $@ 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   %-50s %9.3f ms  %s' % (d['config'][:50], d['ms'], d.get('rowsum_max_rel_err')))"; }
for round in 1 2; do
for tag in default ${VARIANTS}; do
  if [ $tag = default ]; then unset MI_SPARSE_RT; else export MI_SPARSE_RT=$GRAFT_REPO_ROOT/sparse_dot_amd/csrc/build/var/libmi_sparse_$tag.so; fi
  echo "== $tag"
  run spgemm --no-order --reps 5
  run spgemm --kind rmat --scale 18 --per-row 16 --no-order
  run spgemm --kind rmat --scale 20 --per-row 16 --no-order --reps 2
done; done 2>&1 | tee $O/spgemm_ab.log
unset MI_SPARSE_RT
MI_BENCH_OPTS=trace_phases=1 timeout 300 python tools/bench_ops.py spgemm --kind rmat --scale 20 --per-row 16 --no-order --reps 1 2>&1 | grep "mi_sparse spgemm" | tail -12 | tee $O/spgemm_phases.log
