# Round-3 session X: kernel stats of the bench run (headline + secondary with the 152 KiB gram tiles), gram counters, 2-rank dry run
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03x; O=$GRAFT_REPO_ROOT/gpurun_out/r03x; R=$GRAFT_REPO_ROOT
echo "== 2-rank dry run (gloo, both ranks on GPU 0)"
BENCH_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 > $O/bench_2rank_dry.log 2>&1; echo "rc=$?"; grep '^{' $O/bench_2rank_dry.log | tail -1 | cut -c1-400
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o r03 -- python $R/bench.py --steps 20 --warmup 3 --no-cpu --no-secondary --no-pmc > $O/prof_bench.log 2>&1; echo "prof bench rc=$?"
f=$(find $O/prof_bench -name "*kernel_stats.csv" | head -1); head -4 $f | cut -c1-200; cp $f $O/bench_kernel_stats.csv
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sec -o r03 -- python $R/bench.py --steps 5 --warmup 3 --no-cpu --no-pmc --secondary spgemm,spgemm_rmat,gram > $O/prof_sec.log 2>&1; echo "prof secondary rc=$?"
f=$(find $O/prof_sec -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-160; cp $f $O/secondary_kernel_stats.csv
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc_gram/p$i -o p$i -- python $R/tools/bench_ops.py gram --dense --cols 262144 --rows-log2 22 --reps 1 > $O/pmc_gram_p$i.log 2>&1; echo "pmc gram pass $i rc=$?"
done
python $R/tools/pmc_kernels.py $O/pmc_gram 2>&1 | grep "syrkd" | tee $O/pmc_gram_kernels.jsonl | cut -c1-1200
find $O -name "*.csv" -size +4M -delete
