# A/B of the flat SpMM kernel against k_spmm on the headline matrix (timing + one PMC pass group each)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02c; O=$GRAFT_REPO_ROOT/gpurun_out/r02c
V="1:8192:256:0,2:8192:256:0,2:8192:256:1,2:4096:256:1,2:0:256:1,4:8192:256:0,4:8192:256:1,4:4096:256:1,4:3072:256:1,4:2048:256:1,4:0:256:1,4:4096:128:1,4:4096:512:1,8:8192:256:1"
timeout 600 python tools/spmm_sweep.py --variants $V > $O/sweep_rmat.log 2>&1; echo "sweep rc=$?"; cut -c1-230 $O/sweep_rmat.log
timeout 600 python tools/spmm_sweep.py --workload uniform --variants "1:0:256:0,2:0:256:0,2:0:256:1,4:0:256:1" > $O/sweep_uniform.log 2>&1; cut -c1-230 $O/sweep_uniform.log
timeout 600 python tools/spmm_sweep.py --ncols 256 --variants "4:8192:256:0,4:8192:256:1,8:8192:256:1,8:4096:256:1" > $O/sweep_n256.log 2>&1; cut -c1-230 $O/sweep_n256.log
timeout 600 python tools/spmm_sweep.py --ncols 64 --variants "2:8192:256:0,2:8192:256:1,1:8192:256:1,1:8192:256:0" > $O/sweep_n64.log 2>&1; cut -c1-230 $O/sweep_n64.log
timeout 600 python tools/spmm_sweep.py --ncols 32 --variants "1:8192:256:0,1:8192:256:1" > $O/sweep_n32.log 2>&1; cut -c1-230 $O/sweep_n32.log
timeout 600 python tools/spmm_sweep.py --dtype f64 --variants "4:8192:256:0,4:8192:256:1,8:8192:256:1,2:8192:256:0" > $O/sweep_f64.log 2>&1; cut -c1-230 $O/sweep_f64.log
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 900 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc/p$i -o p$i -- python $GRAFT_REPO_ROOT/tools/spmm_sweep.py --launches 2 --variants $V > $O/pmc_p$i.log 2>&1
  echo "pmc pass $i rc=$?"
done
cd $GRAFT_REPO_ROOT
python tools/pmc_sweep_summary.py $O/pmc_p1.log $O/pmc > $O/pmc_table.jsonl 2>&1
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/r02c/pmc_table.jsonl") if l.startswith("{")]
for r in rows: print({k:r.get(k) for k in ("slices","hot_kb","chunk","flat","kernel_ms","l2_hit","fetch_GB_x2","traffic_GB")})
PY
find $O/pmc -name "*.csv" -size +20M -delete
timeout 300 python -m pytest tests/test_gpu_staged_sypr.py -m gpu -q -x -p no:cacheprovider --tb=long 2>&1 | tail -40
ONLY_STAGED= python tools/gpu_api_overhead.py 2>&1 | grep "####\|==\|mi_sparse_s_mm\|create\|python" | head -50
