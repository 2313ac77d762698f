# Round-3 stability check on the final tree: the GPU suite twice in one process each, the second time in REVERSE file order
# (option state left behind by one test file must not change another's results), then smoke()
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03s2; O=$GRAFT_REPO_ROOT/gpurun_out/r03s2
timeout 1700 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/pytest_1.log 2>&1; echo "pytest forward rc=$?"; tail -1 $O/pytest_1.log
timeout 1700 python -m pytest $(ls tests/test_gpu_*.py | sort -r) -m gpu -q --tb=short -p no:cacheprovider > $O/pytest_2.log 2>&1; echo "pytest reverse rc=$?"; tail -1 $O/pytest_2.log
grep -E "FAILED|Error" $O/pytest_1.log $O/pytest_2.log | head -20
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
