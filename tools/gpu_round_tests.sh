cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=8 > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?"; tail -14 gpurun_out/pytest.log
