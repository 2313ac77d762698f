cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for i in 1 2; do timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_$i.log 2>&1; echo "pytest run $i rc=$?"; tail -2 gpurun_out/pytest_$i.log; done
