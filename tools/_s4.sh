cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s4; mkdir -p $O
pmc() {  # $1 = output tag, rest = command
  local tag=$1; shift; local i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAVES" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
    i=$((i+1)); ( cd /tmp && TMPDIR=/tmp timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/$tag/p$i -o p -- "$@" > $O/$tag.p$i.log 2>&1 )
  done
  python tools/pmc_kernels.py $O/$tag | grep "mi::" > $O/$tag.jsonl; rm -rf $O/$tag; cut -c1-1500 $O/$tag.jsonl | grep 'k_spgemm_lds\|k_row_ub\|onepass'
}
MI_BENCH_OPTS=spgemm_onepass=0,spgemm_packed=0 pmc unpacked python $R/tools/bench_ops.py spgemm --no-order --reps 2
MI_BENCH_OPTS=spgemm_onepass=0,spgemm_packed=1 pmc packed python $R/tools/bench_ops.py spgemm --no-order --reps 2
( cd /tmp && TMPDIR=/tmp MI_BENCH_OPTS=spgemm_onepass=0,spgemm_packed=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o u -- python $R/tools/bench_ops.py spgemm --no-order --reps 5 > $O/st.log 2>&1 ); cp $(find $O/st -name "*kernel_stats.csv" | head -1) $O/packed_kernel_stats.csv; rm -rf $O/st; head -12 $O/packed_kernel_stats.csv | cut -c1-200
( cd /tmp && TMPDIR=/tmp MI_BENCH_OPTS=spgemm_onepass=0,spgemm_packed=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o u -- python $R/tools/bench_ops.py spgemm --no-order --reps 5 > $O/st.log 2>&1 ); cp $(find $O/st -name "*kernel_stats.csv" | head -1) $O/unpacked_kernel_stats.csv; rm -rf $O/st; head -12 $O/unpacked_kernel_stats.csv | cut -c1-200
