cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s6; mkdir -p $O
export MI_BENCH_OPTS=spgemm_packed=0
( cd /tmp && TMPDIR=/tmp timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o u -- python $R/tools/bench_ops.py spgemm --no-order --reps 5 > $O/st.log 2>&1 ); cp $(find $O/st -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv; rm -rf $O/st
python - <<PY
import csv
for r in csv.DictReader(open("$O/kernel_stats.csv")):
    n=r["Name"]
    if "mi::" in n and float(r["AverageNs"])>8000: print("%-70s calls %3s avg %10.1f us" % (n[:70], r["Calls"], float(r["AverageNs"])/1e3))
PY
pmc() {  # $1 = output tag, rest = command
  local tag=$1; shift; local i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
    i=$((i+1)); ( cd /tmp && TMPDIR=/tmp timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/$tag/p$i -o p -- "$@" > $O/$tag.p$i.log 2>&1 )
  done
  python tools/pmc_kernels.py $O/$tag | grep "mi::" > $O/$tag.jsonl; rm -rf $O/$tag; cut -c1-1600 $O/$tag.jsonl | grep 'k_spgemm_grp'
}
pmc grp python $R/tools/bench_ops.py spgemm --no-order --reps 2
MI_BENCH_OPTS=trace_phases=1,spgemm_packed=0 timeout 300 python tools/bench_ops.py spgemm --no-order --reps 1 2>&1 | grep "mi_sparse spgemm" | tail -5
