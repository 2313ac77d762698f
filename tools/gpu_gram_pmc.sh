# PMC passes on the literal dense gram: HBM fetch / write and L2 hit rate of the gram kernel, static tile order vs cluster queue
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/gram_pmc; mkdir -p $O
for opts in gram_cluster=0 gram_cluster=8 gram_sliced=0; do
  i=0
  for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    MI_BENCH_OPTS=$opts timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/$opts/p$i -o p$i -- python $GRAFT_REPO_ROOT/tools/bench_ops.py gram --dense --cols 262144 --rows-log2 22 --reps 1 > $O/${opts}_p$i.log 2>&1
    echo "$opts pass $i rc=$?"
  done
  python $GRAFT_REPO_ROOT/tools/pmc_kernels.py $O/$opts 2>&1 | grep -i "syrkd" | cut -c1-900
done
find $O -name "*.csv" -size +4M -delete
