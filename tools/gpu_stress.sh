cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
fails=0
for i in $(seq 1 12); do
  timeout 300 python -m pytest tests/test_gpu_golden.py tests/test_gpu_reference_matrix.py -m gpu -q --tb=line -p no:cacheprovider > gpurun_out/stress_$i.log 2>&1; rc=$?
  if [ $rc -ne 0 ]; then fails=$((fails+1)); echo "run $i rc=$rc"; tail -5 gpurun_out/stress_$i.log; fi
done
echo "stress failures: $fails / 12"
for i in 1 2 3; do timeout 600 python -m pytest tests -m gpu -q --tb=line -p no:cacheprovider 2>&1 | tail -1; done
python - <<'PY'
import time, numpy as np, scipy.sparse as sps, sys
sys.path.insert(0, '.')
import sparse_dot_amd as sda
a = sps.random(10000, 10000, density=0.01, format="csr", dtype=np.float64, random_state=0)
b = np.random.default_rng(1).random((10000, 64))
sda.dot_product_mkl(a, b)
ts=[]
for _ in range(7):
    t0=time.perf_counter(); sda.dot_product_mkl(a, b); ts.append(time.perf_counter()-t0)
print("cfg1 end-to-end through the Python API (host numpy in/out, PCIe included): median %.2f ms" % (sorted(ts)[3]*1e3))
out=np.zeros((10000,64))
ts=[]
for _ in range(7):
    t0=time.perf_counter(); sda.dot_product_mkl(a, b, out=out, out_scalar=0.0); ts.append(time.perf_counter()-t0)
print("cfg1 end-to-end with preallocated out: median %.2f ms" % (sorted(ts)[3]*1e3))
t0=time.perf_counter(); r = a @ b; print("scipy 1T: %.2f ms" % ((time.perf_counter()-t0)*1e3))
PY
