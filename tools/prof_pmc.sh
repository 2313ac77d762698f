# PMC passes for the headline SpMM (one rocprofv3 run per counter group; kernel-trace only).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc
mkdir -p $OUT
rocprofv3 -L > $OUT/counters_list.txt 2>&1
BENCH="python $R/bench.py --steps 3 --warmup 1 --no-cpu ${BENCH_EXTRA}"
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -o p$i -- $BENCH > $OUT/p$i.log 2>&1
  echo "pass $i [$grp] rc=$?"
done <<'GRPS'
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
FETCH_SIZE
WRITE_SIZE
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUSY_avr
GRBM_GUI_ACTIVE TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum
GRPS
ls $OUT
