cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { timeout 120 python tools/bench_ops.py $@ 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   %-50s %9.3f ms  %s' % (d['config'][:50], d['ms'], d.get('rowsum_max_rel_err')))"; }
for tag in default xr0 xr8 xr128 xr32g4; do
  if [ $tag = default ]; then unset MI_SPARSE_RT; else export MI_SPARSE_RT=$GRAFT_REPO_ROOT/sparse_dot_amd/csrc/build/var/libmi_sparse_$tag.so; fi
  echo "== $tag"
  run spgemm --no-order --reps 5
  run spgemm --kind rmat --scale 18 --per-row 16 --no-order
  run spgemm --kind rmat --scale 20 --per-row 16 --no-order --reps 2
done 2>&1 | tee gpurun_out/part_variants_r02.log
unset MI_SPARSE_RT
MI_BENCH_OPTS=trace_phases=1 timeout 120 python tools/bench_ops.py spgemm --kind rmat --scale 20 --per-row 16 --no-order --reps 1 2>&1 | grep "mi_sparse spgemm" | tail -12
