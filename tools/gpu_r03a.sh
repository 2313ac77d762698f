# (the option spmm_stream_nt existed only in this session's build: measured, rejected, removed -- profiles/r03_spmm_stream_nt_ab.log)
# Round-3 session A: the large-operand tagged gather test, SpMV goldens on the GPU, x-gather probe, SpMM stream-policy A/B.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03a; O=$GRAFT_REPO_ROOT/gpurun_out/r03a
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
( time timeout 900 python -m pytest tests/test_gpu_baseline_configs.py -m gpu -q --tb=short -p no:cacheprovider -k "tagged_gather_large" ) > $O/pytest_large.log 2>&1; echo "pytest large rc=$?"; tail -15 $O/pytest_large.log
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -m gpu -q --tb=short -p no:cacheprovider -k "tagged or spmv or spmm" ) > $O/pytest_spmm.log 2>&1; echo "pytest spmm rc=$?"; tail -5 $O/pytest_spmm.log
timeout 300 tools/probes/xgather_probe > $O/xgather_probe.log 2>&1; echo "xgather rc=$?"; cat $O/xgather_probe.log
timeout 600 python tools/spmm_sweep.py --launches 10 --variants "2:8192:256,2:8192:256:spmm_stream_nt=1,2:8192:256:spmm_tag_struct=1,2:4096:256:spmm_stream_nt=1,2:3072:256:spmm_stream_nt=1,4:3072:256,4:3072:256:spmm_stream_nt=1,4:4096:256:spmm_stream_nt=1,2:0:256,2:0:256:spmm_stream_nt=1" > $O/spmm_stream_nt.log 2>&1; echo "sweep rc=$?"; cat $O/spmm_stream_nt.log | cut -c1-330
timeout 300 python tools/gpu_spmv.py > $O/spmv.log 2>&1; cat $O/spmv.log
