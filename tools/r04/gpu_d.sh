# Round-4 session D: knock-outs of the pipelined gram kernel + LDS atomic rate probe
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04d; O=$GRAFT_REPO_ROOT/gpurun_out/r04d
tools/probes/lds_atomic_probe > $O/lds_atomic_probe.log 2>&1; cat $O/lds_atomic_probe.log
g() { timeout 400 python tools/bench_ops.py gram --dense $@ 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   %-60s %9.3f ms  %s' % (d['config'][:60], d['ms'], d.get('checks')))"; }
for tag in default nostore nowalk noatomic intatomic plainrmw noatomic_nostore; do
  if [ $tag = default ]; then unset MI_SPARSE_RT; else export MI_SPARSE_RT=$GRAFT_REPO_ROOT/sparse_dot_amd/csrc/build/var/libmi_sparse_$tag.so; fi
  echo "== $tag"; g --cols 262144 --rows-log2 22 --reps 3
done 2>&1 | tee $O/gram_knockouts_pipelined.log
