cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for i in 1 2; do
  timeout 1700 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_$i.log 2>&1; echo "pytest $i rc=$?"; tail -2 gpurun_out/pytest_$i.log | cut -c1-300
done
grep -E "FAILED|Error" gpurun_out/pytest_1.log | head -20
