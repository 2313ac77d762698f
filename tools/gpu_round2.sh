# Round-2 GPU session: parity tests, bench, rocprof stats, PMC passes (SpMM + SpGEMM + gram kernels), 2-rank dry run.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02f; O=$GRAFT_REPO_ROOT/gpurun_out/r02f
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
( time timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=15 ) > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -25 $O/pytest.log
( time timeout 1500 python bench.py --steps 20 --warmup 3 ) > $O/bench.log 2>&1; echo "bench rc=$?"; grep '^{' $O/bench.log | tail -1 | cut -c1-3000
echo "== 2-rank dry run (gloo, both ranks on GPU 0)"
BENCH_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 3 > $O/bench_2rank_dry.log 2>&1; echo "rc=$?"; grep '^{' $O/bench_2rank_dry.log | tail -1 | cut -c1-1500
echo "== host-array call overhead (staged copies on / off)"
timeout 600 python tools/gpu_api_overhead.py > $O/api_overhead.log 2>&1; echo "rc=$?"; grep "ms per call" $O/api_overhead.log | cut -c1-120
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o r02 -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu --no-secondary > $O/prof_bench.log 2>&1; echo "prof bench rc=$?"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sec -o r02 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --no-cpu --secondary spgemm,spgemm_rmat,gram > $O/prof_sec.log 2>&1; echo "prof secondary rc=$?"
# PMC: headline SpMM (final configuration), separate passes
V="0:8192:256"
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc_spmm/p$i -o p$i -- python $GRAFT_REPO_ROOT/tools/spmm_sweep.py --launches 3 --variants $V > $O/pmc_spmm_p$i.log 2>&1
  echo "pmc spmm pass $i rc=$?"
done
python $GRAFT_REPO_ROOT/tools/pmc_sweep_summary.py $O/pmc_spmm_p1.log $O/pmc_spmm > $O/pmc_spmm_table.jsonl 2>&1; cut -c1-700 $O/pmc_spmm_table.jsonl
# PMC: SpGEMM (uniform cfg3 + R-MAT scale 18: same kernels as the literal cfg3 at 1/8 of its memory) and dense gram
for wl in "spgemm --reps 1 --no-order" "spgemm --kind rmat --scale 18 --per-row 16 --reps 1 --no-order" "gram --dense --reps 1"; do
  tag=$(echo $wl | tr -c 'a-z0-9' '_' | cut -c1-24)
  i=0
  for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES"; do
    i=$((i+1))
    timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc_$tag/p$i -o p$i -- python $GRAFT_REPO_ROOT/tools/bench_ops.py $wl > $O/pmc_${tag}_p$i.log 2>&1
    echo "pmc $tag pass $i rc=$?"
  done
  python $GRAFT_REPO_ROOT/tools/pmc_kernels.py $O/pmc_$tag > $O/pmc_${tag}_kernels.jsonl 2>&1
done
find $O -name "*.csv" -size +8M -delete
ls $O | head -50
