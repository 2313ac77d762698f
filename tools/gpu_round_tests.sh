cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for i in 1 2 3; do timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "concurrent" 2>&1 | tail -3; done
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest.log
