# Round-3 session Q (end of round, after the last gram / SpMV / BSR changes): full GPU suite, the driver's bench line, kernel stats, SpMV counters, 2-rank dry run
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03q; O=$GRAFT_REPO_ROOT/gpurun_out/r03q
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
( time timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=15 ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -24 $O/pytest.log | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl"
( time timeout 1800 python bench.py --steps 20 --warmup 3 ) > $O/bench.log 2>&1; echo "bench rc=$?"; grep '^{' $O/bench.log | tail -1 > $O/bench_line.json; python - <<'PY'
import json,os
d=json.load(open(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r03q/bench_line.json"))
print(json.dumps({k:d[k] for k in ("value","ms_per_step","roofline","plan")})[:1800])
for k,v in d.get("secondary",{}).items():
    print(k, json.dumps({kk:vv for kk,vv in v.items() if kk in ("ms","value","roofline","first_call_ms","first_call_fresh_process","ms_per_step","workload","rows2e18_8.4Mnnz_fp32_x128_ms","error")})[:600])
PY
echo "== 2-rank dry run (gloo, both ranks on GPU 0)"
BENCH_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 > $O/bench_2rank_dry.log 2>&1; echo "rc=$?"; grep '^{' $O/bench_2rank_dry.log | tail -1 | cut -c1-600
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o r03 -- python $R/bench.py --steps 20 --warmup 3 --no-cpu --no-secondary --no-pmc > $O/prof_bench.log 2>&1; echo "prof bench rc=$?"
f=$(find $O/prof_bench -name "*kernel_stats.csv" | head -1); head -4 $f | cut -c1-200; cp $f $O/bench_kernel_stats.csv
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sec -o r03 -- python $R/bench.py --steps 5 --warmup 3 --no-cpu --no-pmc --secondary spgemm,spgemm_rmat,gram > $O/prof_sec.log 2>&1; echo "prof secondary rc=$?"
f=$(find $O/prof_sec -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-160; cp $f $O/secondary_kernel_stats.csv
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc_spmv/p$i -o p$i -- python $R/tools/gpu_spmv.py > $O/pmc_spmv_p$i.log 2>&1; echo "pmc spmv pass $i rc=$?"
done
python $R/tools/pmc_kernels.py $O/pmc_spmv 2>&1 | grep "k_spmv" | tee $O/pmc_spmv_kernels.jsonl | cut -c1-900
find $O -name "*.csv" -size +4M -delete
