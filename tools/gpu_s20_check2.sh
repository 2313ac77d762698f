cd $GRAFT_REPO_ROOT
run() { python tools/bench_ops.py $@ 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   %-40s %9.3f ms  rowsum %.3g' % (d['config'][:40], d['ms'], d['rowsum_max_rel_err']))"; }
for b in 0 1; do echo "== bias $b"; MI_BENCH_OPTS=spgemm_part_log2s_bias=$b run spgemm --kind rmat --scale 20 --per-row 16 --no-order --reps 2; done
bash tools/gpu_prof_s20.sh 2>&1 | head -11
