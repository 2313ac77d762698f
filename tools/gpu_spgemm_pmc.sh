# PMC passes on the literal configs[2] SpGEMM (R-MAT 2^20, 16/row, fp64): L2 hit rate and HBM fetch / write of the big-row kernels
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/spgemm_pmc; mkdir -p $O
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/s20/p$i -o p$i -- python $GRAFT_REPO_ROOT/tools/bench_ops.py spgemm --kind rmat --scale 20 --per-row 16 --no-order --reps 1 > $O/s20_p$i.log 2>&1
  echo "pass $i rc=$?"
done
python $GRAFT_REPO_ROOT/tools/pmc_kernels.py $O/s20 2>&1 | grep -E "k_spgemm_part|k_spgemm_bitmap|k_part_slices" | cut -c1-600
find $O -name "*.csv" -size +4M -delete
