cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s7; mkdir -p $O
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   %-50s %9.3f ms  %s' % (d['config'][:50], d['ms'], d.get('rowsum_max_rel_err')))"; }
for round in 1 2; do for v in default kost staged; do for op in spgemm_packed=0 spgemm_packed=1; do
  unset MI_SPARSE_RT; [ $v = default ] || export MI_SPARSE_RT=$R/sparse_dot_amd/csrc/build/var/libmi_sparse_$v.so
  echo "== $v $op"; MI_BENCH_OPTS=$op timeout 300 python tools/bench_ops.py spgemm --no-order --reps 5 2>&1 | tail -1 | line
done; done; done 2>&1 | tee $O/ab.log
export MI_SPARSE_RT=$R/sparse_dot_amd/csrc/build/var/libmi_sparse_staged.so
for op in spgemm_packed=0 spgemm_packed=1; do
( cd /tmp && TMPDIR=/tmp MI_BENCH_OPTS=$op timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum --output-format csv -d $O/pm/p1 -o p -- python $R/tools/bench_ops.py spgemm --no-order --reps 2 > $O/pm.log 2>&1 )
python tools/pmc_kernels.py $O/pm | grep "k_spgemm_grp" | cut -c1-400; rm -rf $O/pm
( cd /tmp && TMPDIR=/tmp MI_BENCH_OPTS=$op timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o u -- python $R/tools/bench_ops.py spgemm --no-order --reps 5 > $O/st.log 2>&1 ); grep k_spgemm_grp $(find $O/st -name "*kernel_stats.csv" | head -1) | cut -c1-60,330-420; rm -rf $O/st
done
