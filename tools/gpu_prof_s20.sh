cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_s20 -- python $GRAFT_REPO_ROOT/tools/bench_ops.py spgemm --kind rmat --scale 20 --per-row 16 --reps 1 --no-order 2>&1 | tail -1 | cut -c1-300
f=$(ls $GRAFT_REPO_ROOT/gpurun_out/prof_s20/*/*kernel_stats.csv | head -1); head -12 $f | cut -c1-40,200-330
rm -f $GRAFT_REPO_ROOT/gpurun_out/prof_s20/*/*kernel_trace.csv
cat > /tmp/m.py <<'PY'
import torch, time, ctypes
hip = ctypes.CDLL(torch.__path__[0] + "/lib/libamdhip64.so")
torch.zeros(1, device="cuda")
for gb in (1, 8, 32, 64):
    p = ctypes.c_void_p()
    t0 = time.perf_counter(); rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(gb << 30)); t1 = time.perf_counter()
    hip.hipFree(p); t2 = time.perf_counter()
    print("hipMalloc %d GiB rc=%d: %.1f ms, free %.1f ms" % (gb, rc, (t1 - t0) * 1e3, (t2 - t1) * 1e3))
PY
python /tmp/m.py
