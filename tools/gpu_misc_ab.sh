cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02d; O=gpurun_out/r02d
timeout 600 python tools/spmm_sweep.py --variants "2:8192:256,2:12288:256,2:16384:256,2:24576:256,2:6144:256,2:8192:512,2:16384:512,1:8192:256,1:16384:256" > $O/sweep_budget.log 2>&1; cut -c1-200 $O/sweep_budget.log
timeout 600 python tools/spmm_sweep.py --ncols 256 --variants "4:8192:256,4:16384:256,4:32768:256,4:4096:256" > $O/sweep_budget_n256.log 2>&1; cut -c1-200 $O/sweep_budget_n256.log
timeout 600 python tools/spmm_sweep.py --ncols 64 --variants "2:8192:256,2:4096:256,2:16384:256,1:8192:256" > $O/sweep_budget_n64.log 2>&1; cut -c1-200 $O/sweep_budget_n64.log
