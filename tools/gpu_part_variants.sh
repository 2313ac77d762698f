cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { python tools/bench_ops.py $@ 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   %-50s %9.3f ms  %s' % (d['config'][:50], d['ms'], d.get('rowsum_max_rel_err')))"; }
for tag in default sp2 sp4 sp16; do
  if [ $tag = default ]; then unset MI_SPARSE_RT; else export MI_SPARSE_RT=$GRAFT_REPO_ROOT/sparse_dot_amd/csrc/build/var/libmi_sparse_$tag.so; fi
  echo "== $tag"
  run spgemm --kind rmat --scale 18 --per-row 16 --no-order
  run spgemm --kind rmat --scale 20 --per-row 16 --no-order --reps 2
  run gram --reps 2
done 2>&1 | tee gpurun_out/part_variants.log
