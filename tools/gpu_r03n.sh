# Round-3 session N: full suite after the BSR export / SpMV descriptor / fix-up changes; SpMV timing; quick bench
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03n; O=$GRAFT_REPO_ROOT/gpurun_out/r03n
( time timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -2
timeout 300 python tools/gpu_spmv.py 2>&1 | grep "SpMV" | tee $O/spmv.log
timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu --no-secondary --no-pmc 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   step %.4f ms  kernel %.4f ms  value %.1f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['value']))"
