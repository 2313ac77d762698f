# Round-3 session K: SpGEMM with precomputed B-row extents (k_row_ub writes them, symbolic / numeric read them): tests, uniform +
# literal + R-MAT 2^18 timings vs the round-2 library on the same box, counters of the uniform case
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03k; O=$GRAFT_REPO_ROOT/gpurun_out/r03k
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_baseline_configs.py tests/test_gpu_staged_sypr.py tests/test_gpu_bsr.py -m gpu -q -x -p no:cacheprovider -k "spgemm or config3 or gram or sypr or staged or bsr or syrk" > $O/pytest_spgemm.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_spgemm.log
s() { timeout 600 python tools/bench_ops.py spgemm $@ --no-order 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   %-50s %9.3f ms  rowsum err %.2e' % (d['config'][:50], d['ms'], d['rowsum_max_rel_err']))"; }
for tag in default r02 default r02; do
  if [ $tag = default ]; then unset MI_SPARSE_RT; else export MI_SPARSE_RT=$GRAFT_REPO_ROOT/sparse_dot_amd/csrc/build/var/libmi_sparse_$tag.so; fi
  echo "== $tag"
  s --reps 5
  s --kind rmat --scale 18 --per-row 16 --reps 3
  s --kind rmat --scale 20 --per-row 16 --reps 2
done 2>&1 | tee $O/spgemm_ext_ab.log
unset MI_SPARSE_RT
timeout 300 python tools/bench_ops.py gram --rows-log2 20 --cols 16384 --reps 2 2>&1 | tail -1 | cut -c1-300
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "WRITE_SIZE" "TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc_u/p$i -o p$i -- python $R/tools/bench_ops.py spgemm --reps 1 --no-order > $O/pmc_u_p$i.log 2>&1; echo "pmc uniform pass $i rc=$?"
done
python $R/tools/pmc_kernels.py $O/pmc_u 2>&1 | grep "mi::k_spgemm_lds\|mi::k_row_ub" | tee $O/pmc_spgemm_uniform_kernels.jsonl | cut -c1-420
find $O -name "*.csv" -size +4M -delete
