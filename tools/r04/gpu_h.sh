# Round-4 session H: where do the 4.0 ms of the uniform SpGEMM go (phase trace + rocprof kernel stats of that call alone)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04i; O=$GRAFT_REPO_ROOT/gpurun_out/r04i
MI_BENCH_OPTS=trace_phases=1 timeout 300 python tools/bench_ops.py spgemm --no-order --reps 2 2>&1 | grep "mi_sparse spgemm\|^{" | tail -24 > $O/uniform_phases.log; cat $O/uniform_phases.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o u -- python $GRAFT_REPO_ROOT/tools/bench_ops.py spgemm --no-order --reps 5 > $O/prof.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); grep "mi::" $f | cut -c1-60,200- | head -20; cp $f $O/uniform_kernel_stats.csv; rm -rf $O/prof
