#!/bin/bash
# Interleaved A/B on ONE GPU box (run through gpurun): the shipped library ("default") against variant libraries built here into
# sparse_dot_amd/csrc/build/var/libmi_sparse_<tag>.so (tools/build_gram_variants.sh, tools/build_part_variants.sh, or by hand) and / or
# against option settings.  Two rounds, variants interleaved, so that box-to-box and warm-up differences cancel.
#   usage: gpu_ab.sh gram|spgemm|spgemm-uniform|spgemm-literal  [tag ...]  [name=value ...]  [tag+name=value ...]    e.g.  gpu_ab.sh gram nostore noatomic      gpu_ab.sh gram gram_queue=0
# gram:   literal configs[3] (2^22 x 262144) and 2^20 x 65536 / x 16384, dense output;  spgemm: uniform, R-MAT 2^18, literal R-MAT 2^20.
cd $GRAFT_REPO_ROOT; what=$1; shift; O=$GRAFT_REPO_ROOT/gpurun_out/ab_$what; mkdir -p $O
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   %-56s %9.3f ms  %s' % (d['config'][:56], d['ms'], d.get('checks') or d.get('rowsum_max_rel_err')))"; }
one() {  # $1 = tag or option
  unset MI_SPARSE_RT MI_BENCH_OPTS
  case "$1" in default) ;; *+*=*) export MI_SPARSE_RT=$GRAFT_REPO_ROOT/sparse_dot_amd/csrc/build/var/libmi_sparse_${1%%+*}.so MI_BENCH_OPTS=${1#*+} ;;
    *=*) export MI_BENCH_OPTS=$1 ;; *) export MI_SPARSE_RT=$GRAFT_REPO_ROOT/sparse_dot_amd/csrc/build/var/libmi_sparse_$1.so ;; esac
  echo "== $1"
  if [ $what = gram ]; then
    for shape in "--cols 262144 --rows-log2 22" "--cols 65536 --rows-log2 20" "--cols 16384 --rows-log2 20"; do timeout 400 python tools/bench_ops.py gram --dense $shape --reps 3 2>&1 | tail -1 | line; done
  elif [ $what = spgemm-literal ]; then
    timeout 300 python tools/bench_ops.py spgemm --kind rmat --scale 18 --per-row 16 --no-order 2>&1 | tail -1 | line
    timeout 300 python tools/bench_ops.py spgemm --kind rmat --scale 20 --per-row 16 --no-order --reps 3 2>&1 | tail -1 | line
  elif [ $what = spgemm-uniform ]; then
    timeout 300 python tools/bench_ops.py spgemm --no-order --reps 7 2>&1 | tail -1 | line
  else
    timeout 300 python tools/bench_ops.py spgemm --no-order --reps 5 2>&1 | tail -1 | line
    timeout 300 python tools/bench_ops.py spgemm --kind rmat --scale 18 --per-row 16 --no-order 2>&1 | tail -1 | line
    timeout 300 python tools/bench_ops.py spgemm --kind rmat --scale 20 --per-row 16 --no-order --reps 2 2>&1 | tail -1 | line
  fi
}
for round in 1 2; do for v in default "$@"; do one $v; done; done 2>&1 | tee $O/ab.log
