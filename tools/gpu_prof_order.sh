cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_order -- python $GRAFT_REPO_ROOT/tools/bench_ops.py spgemm --kind rmat --scale 17 --per-row 16 --reps 1 > /dev/null 2>&1
f=$(ls $GRAFT_REPO_ROOT/gpurun_out/prof_order/*/*kernel_stats.csv | head -1); head -8 $f | cut -c1-150
