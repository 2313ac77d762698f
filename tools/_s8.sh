cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s8; mkdir -p $O
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   %-50s %9.3f ms  %s' % (d['config'][:50], d['ms'], d.get('rowsum_max_rel_err')))"; }
for round in 1 2; do for op in spgemm_group=0 spgemm_group=1; do
  echo "== $op"
  MI_BENCH_OPTS=$op timeout 300 python tools/bench_ops.py spgemm --no-order --reps 5 2>&1 | tail -1 | line
  MI_BENCH_OPTS=$op timeout 300 python tools/bench_ops.py spgemm --kind rmat --scale 18 --per-row 16 --no-order 2>&1 | tail -1 | line
done; done 2>&1 | tee $O/ab.log
MI_BENCH_OPTS=trace_phases=1 timeout 300 python tools/bench_ops.py spgemm --no-order --reps 1 2>&1 | grep "mi_sparse spgemm" | tail -5
( time timeout 2400 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "spgemm or gram or sparse or golden or staged or determin" ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
pmc() {  # $1 = output tag, rest = command
  local tag=$1; shift; local i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"; do
    i=$((i+1)); ( cd /tmp && TMPDIR=/tmp timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/$tag/p$i -o p -- "$@" > $O/$tag.p$i.log 2>&1 )
  done
  python tools/pmc_kernels.py $O/$tag | grep "mi::" > $O/$tag.jsonl; rm -rf $O/$tag; cut -c1-1700 $O/$tag.jsonl | grep 'k_spgemm_part\|k_spgemm_bitmap\|k_part_slices'
}
pmc literal python $R/tools/bench_ops.py spgemm --kind rmat --scale 20 --per-row 16 --no-order --reps 1
