cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03z
g() { timeout 300 python tools/bench_ops.py gram --dense --cols 262144 --rows-log2 22 --reps 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   %9.3f ms' % d['ms'])"; }
for o in gram_persistent=-1 gram_persistent=0 gram_persistent=2 gram_heads=0 gram_sliced=2 gram_persistent=-1; do echo "== $o"; MI_BENCH_OPTS=$o g; done 2>&1 | tee gpurun_out/r03z/gram_option_sweep.log
