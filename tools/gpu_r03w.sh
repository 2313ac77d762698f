# Round-3 session W (end of round): full GPU suite, smoke, bench (committed line), kernel stats, 2-rank dry run
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03w; O=$GRAFT_REPO_ROOT/gpurun_out/r03w
timeout 2400 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2
timeout 1500 python bench.py > $O/bench.log 2> $O/bench.err; echo "bench rc=$?"; tail -1 $O/bench.log | cut -c1-600
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-pmc --no-variants --steps 20 --warmup 5 > $O/bench_prof.log 2>&1; echo "prof rc=$?"
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv; find $O/prof -type f ! -name "*stats*" -delete 2>/dev/null
head -8 $O/bench_kernel_stats.csv | cut -c1-200
