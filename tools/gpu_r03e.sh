# Round-3 session E: FETCH_SIZE calibration for gather patterns; PMC rows of the dense gram kernel at the literal configs[3];
# bsr / sp2m evidence; fresh-process first call of the literal SpGEMM.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03e; mkdir -p $O
i=0
for grp in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/calib/p$i -o p$i -- $R/tools/probes/fetch_calib > $O/calib_p$i.log 2>&1; echo "calib pass $i rc=$?"
done
cat $O/calib_p1.log | grep "^k_"; python $R/tools/pmc_kernels.py $O/calib 2>&1 | cut -c1-600 | tee $O/calib_kernels.jsonl
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/gram/p$i -o p$i -- python $R/tools/bench_ops.py gram --dense --cols 262144 --rows-log2 22 --reps 1 > $O/gram_p$i.log 2>&1; echo "gram pass $i rc=$?"
done
python $R/tools/pmc_kernels.py $O/gram 2>&1 | grep -i "syrkd" | tee $O/pmc_gram_dense_kernels.jsonl | cut -c1-1500
cd $R
for b in 4 8; do timeout 300 python tools/bench_ops.py bsr --rows-log2 18 --block $b --ncols 128 2>&1 | tail -1; done | tee $O/bsr.log | cut -c1-900
timeout 300 python tools/bench_ops.py sp2m --scale 20 --per-row 16 --reps 3 2>&1 | tail -1 | tee $O/sp2m_uniform.log | cut -c1-600
timeout 300 python tools/bench_ops.py sp2m --scale 18 --per-row 16 --kind rmat --reps 3 2>&1 | tail -1 | tee $O/sp2m_rmat18.log | cut -c1-600
timeout 300 python tools/gpu_first_call.py 2>&1 | tail -1 | tee $O/first_call.log
find $O -name "*.csv" -size +4M -delete
