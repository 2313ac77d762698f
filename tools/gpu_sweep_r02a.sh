# Round-2 session A: slice / hot-budget sweep of k_spmm with timing and PMC passes; literal cfg4 trial.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02a; O=$GRAFT_REPO_ROOT/gpurun_out/r02a
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
V="1:0:256,1:8192:256,2:0:256,2:4096:256,2:8192:256,4:0:256,4:2048:256,4:3072:256,4:4096:256,4:8192:256,8:0:256,8:2048:256,8:3072:256,8:4096:256,8:8192:256,4:3072:128,4:3072:512,8:3072:512"
timeout 600 python tools/spmm_sweep.py --variants $V > $O/sweep_rmat.log 2>&1; echo "sweep rc=$?"; cat $O/sweep_rmat.log | cut -c1-260
timeout 600 python tools/spmm_sweep.py --workload uniform --variants "1:0:256,2:0:256,4:0:256,8:0:256" > $O/sweep_uniform.log 2>&1; cat $O/sweep_uniform.log | cut -c1-260
timeout 600 python tools/spmm_sweep.py --ncols 256 --variants "1:0:256,1:8192:256,4:3072:256,8:3072:256,8:6144:256" > $O/sweep_n256.log 2>&1; cat $O/sweep_n256.log | cut -c1-260
timeout 600 python tools/spmm_sweep.py --ncols 64 --variants "1:0:256,1:8192:256,2:3072:256,4:3072:256" > $O/sweep_n64.log 2>&1; cat $O/sweep_n64.log | cut -c1-260
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 900 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc/p$i -o p$i -- python $GRAFT_REPO_ROOT/tools/spmm_sweep.py --launches 2 --variants $V > $O/pmc_p$i.log 2>&1
  echo "pmc pass $i [$grp] rc=$?"
done
cd $GRAFT_REPO_ROOT
python tools/pmc_sweep_summary.py $O/pmc_p1.log $O/pmc > $O/pmc_table.jsonl 2>&1; cut -c1-600 $O/pmc_table.jsonl
find $O/pmc -name "*.csv" -size +20M -delete
echo "== cfg4 literal trial"
timeout 900 python tools/bench_ops.py gram --dense --cols 262144 --rows-log2 22 --reps 1 > $O/cfg4_literal.log 2>&1; echo "rc=$?"; tail -3 $O/cfg4_literal.log | cut -c1-600
