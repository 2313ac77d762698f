# Round-3 session M: SpMM with chunk descriptors in the plan + workgroup-per-task fix-up: tests, step time vs the round-2 library
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03m; O=$GRAFT_REPO_ROOT/gpurun_out/r03m
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_reference_matrix.py tests/test_gpu_baseline_configs.py -m gpu -q -x -p no:cacheprovider -k "spmm or spmv or config1 or config2 or config5 or determinism or device_matrix or c_abi or tagged" > $O/pytest_spmm.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_spmm.log
for tag in default r02 default r02; do
  if [ $tag = default ]; then unset MI_SPARSE_RT; else export MI_SPARSE_RT=$GRAFT_REPO_ROOT/sparse_dot_amd/csrc/build/var/libmi_sparse_$tag.so; fi
  echo "== $tag"
  timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu --no-secondary --no-pmc 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   step %.4f ms  kernel %.4f ms  value %.1f  first call %.3f  plan_ms %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['value'], d['plan']['first_call_ms'], d['plan']['plan_ms']))"
  timeout 300 python tools/gpu_spmv.py 2>&1 | grep "chunk  256"
done 2>&1 | tee $O/spmm_desc_ab.log
