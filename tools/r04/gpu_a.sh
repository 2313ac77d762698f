# Round-4 session A: the SpMM gather's empirical ceiling (probe: cache policies, pinned hot set, two passes) + the two-pass library experiment
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04a; O=$GRAFT_REPO_ROOT/gpurun_out/r04a
( time timeout 600 tools/probes/spmm_gather_probe ) > $O/probe.log 2>&1; echo "probe rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d $O/pmc -o p -- $GRAFT_REPO_ROOT/tools/probes/spmm_gather_probe H=16384 > $O/probe_pmc.log 2>&1; echo "pmc rc=$?"
cd $GRAFT_REPO_ROOT
python tools/r04/probe_pmc_join.py $O/probe_pmc.log $O/pmc > $O/probe_pmc_joined.log 2>&1
rm -rf $O/pmc
( time timeout 900 python tools/r04/exp_twopass2.py ) > $O/twopass.log 2>&1; echo "twopass rc=$?"
tail -5 $O/probe.log; cat $O/probe_pmc_joined.log | head -50; cat $O/twopass.log | tail -12
