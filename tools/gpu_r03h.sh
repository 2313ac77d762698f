# Round-3 session H: SpGEMM literal configs[2] -- (row, range) items in column order vs row order (same box), kernel times, counters
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03h; O=$GRAFT_REPO_ROOT/gpurun_out/r03h
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -m gpu -q -x -p no:cacheprovider -k "spgemm or config3 or gram_sparse" > $O/pytest_spgemm.log 2>&1; echo "pytest spgemm rc=$?"; tail -4 $O/pytest_spgemm.log
for s in 0 1 0 1; do
  echo "== spgemm_part_sorted=$s"
  MI_BENCH_OPTS=spgemm_part_sorted=$s timeout 600 python tools/bench_ops.py spgemm --kind rmat --scale 20 --per-row 16 --reps 3 --no-order 2>&1 | tail -1 | cut -c1-400
done 2>&1 | tee $O/spgemm_sorted_ab.log
MI_BENCH_OPTS=spgemm_part_sorted=1 timeout 600 python tools/bench_ops.py spgemm --kind rmat --scale 18 --per-row 16 --reps 3 --no-order 2>&1 | tail -1 | cut -c1-300
MI_BENCH_OPTS=spgemm_part_sorted=0 timeout 600 python tools/bench_ops.py spgemm --kind rmat --scale 18 --per-row 16 --reps 3 --no-order 2>&1 | tail -1 | cut -c1-300
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for s in 0 1; do
  MI_BENCH_OPTS=spgemm_part_sorted=$s timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$s -o st -- python $R/tools/bench_ops.py spgemm --kind rmat --scale 20 --per-row 16 --reps 2 --no-order > $O/stats_$s.log 2>&1
  f=$(find $O/stats_$s -name "*kernel_stats.csv" | head -1); echo "-- sorted=$s"; head -12 $f | cut -c1-200
done
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "WRITE_SIZE" "TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  MI_BENCH_OPTS=spgemm_part_sorted=1 timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc/p$i -o p$i -- python $R/tools/bench_ops.py spgemm --kind rmat --scale 20 --per-row 16 --reps 1 --no-order > $O/pmc_p$i.log 2>&1; echo "pmc pass $i rc=$?"
done
python $R/tools/pmc_kernels.py $O/pmc 2>&1 | grep "mi::" | tee $O/pmc_spgemm_literal_kernels.jsonl | cut -c1-500
find $O -name "*.csv" -size +4M -delete
