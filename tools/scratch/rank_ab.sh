#!/bin/bash
# per variant: rocprofv3 kernel stats of the literal R-MAT SpGEMM; prints the big-row kernels
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/$1; shift; mkdir -p $O
for v in "$@"; do
  unset MI_SPARSE_RT MI_BENCH_OPTS
  case "$v" in default) ;; *=*) export MI_BENCH_OPTS=$v ;; *) export MI_SPARSE_RT=$GRAFT_REPO_ROOT/sparse_dot_amd/csrc/build/var/libmi_sparse_$v.so ;; esac
  for sc in ${SCALES:-20}; do
  ( cd /tmp && TMPDIR=/tmp timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_$v -o p -- python $GRAFT_REPO_ROOT/tools/bench_ops.py spgemm --kind rmat --scale $sc --per-row 16 --no-order --reps 2 > $O/run_$v.log 2>&1 )
  echo "== $v scale $sc"; python - $O/p_$v/p_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if any(k in n for k in ('k_spgemm_rank', 'k_spgemm_bitmap', 'k_spgemm_part', 'k_part_slices', 'k_blkptr')):
        print('   %-44s calls %3s  avg %9.3f ms  total %9.3f ms' % (n[:44], r['Calls'], float(r['AverageNs']) / 1e6, float(r['TotalDurationNs']) / 1e6))
PY
  rm -rf $O/p_$v
  done
done 2>&1 | tee $O/ab.log
