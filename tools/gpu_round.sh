# One GPU-box session: parity tests, bench (+variants), calibration probe, rocprof summary, PMC passes.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"
timeout 1700 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/pytest.log
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench.log | cut -c1-700
for v in "--unroll 8" "--chunk 128" "--chunk 512" "--chunk 1024" "--unroll 8 --chunk 512" "--workload uniform" "--workload uniform --unroll 8"; do
  echo "== variant $v"; timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu $v 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
done 2>&1 | tee gpurun_out/variants.log
timeout 300 tools/probes/gather_probe > gpurun_out/gather_probe.log 2>&1; cat gpurun_out/gather_probe.log
cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT; ls gpurun_out/prof | head
bash tools/prof_pmc.sh 2>&1 | tail -12
