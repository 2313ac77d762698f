# Round-4 session M: counters of the gram kernel, pull queue on / off (FETCH_SIZE, WRITE_SIZE, L2 hit, wave-cycle split)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04m; mkdir -p $O
for q in 1 0; do
  i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES"; do
    i=$((i+1))
    MI_BENCH_OPTS=gram_queue=$q timeout 400 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/q$q/p$i -o p -- python $R/tools/bench_ops.py gram --dense --cols 262144 --rows-log2 22 --reps 1 > $O/q$q.p$i.log 2>&1
  done
  echo "== gram_queue=$q"; python $R/tools/pmc_kernels.py $O/q$q | grep k_syrkd | tee $O/pmc_gram_queue$q.jsonl | cut -c1-900
  rm -rf $O/q$q
done
