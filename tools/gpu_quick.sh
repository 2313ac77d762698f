# quick GPU check: the tests named in $T (default: baseline configs + config3 scale), bench with selected secondaries
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/quick; O=$GRAFT_REPO_ROOT/gpurun_out/quick
( time timeout 1800 python -m pytest ${T:-tests/test_gpu_baseline_configs.py tests/test_gpu_parity.py} -m gpu -q --tb=short -p no:cacheprovider -x --durations=8 ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -16 $O/pytest.log
( time timeout 1500 python bench.py --steps 20 --warmup 3 ${BENCH_ARGS} ) > $O/bench.log 2>&1; echo "bench rc=$?"
grep '^{' $O/bench.log | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('value',d['value'],'ms',d['ms_per_step'],'kernel_ms',d['roofline']['kernel_ms'],'frac',d['roofline']['frac'])
print('plan',json.dumps(d.get('plan',{}))[:400])
for k,v in (d.get('secondary') or {}).items():
    print(k, {kk:vv for kk,vv in v.items() if kk in ('ms','ms_per_step','value','first_call_ms','error','cfg1_10k_fp64_x64_ms','rows2e18_8.4Mnnz_fp32_x128_ms')}, (v.get('roofline') or {}).get('frac'))
"
