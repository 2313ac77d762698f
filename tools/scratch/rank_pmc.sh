#!/bin/bash
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; shift
[ -n "$1" ] && export MI_BENCH_OPTS=$1
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1)); ( cd /tmp && TMPDIR=/tmp timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc/p$i -o p -- python $R/tools/bench_ops.py spgemm --kind rmat --scale 20 --per-row 16 --no-order --reps 1 > $O/pmc.p$i.log 2>&1 )
done
python tools/pmc_kernels.py $O/pmc | grep "k_spgemm_rank\|k_spgemm_bitmap\|k_spgemm_part\|k_part_slices" > $O/pmc.jsonl; rm -rf $O/pmc; cat $O/pmc.jsonl
