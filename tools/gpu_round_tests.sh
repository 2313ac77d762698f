cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest.log
echo "== gram dense 2^22 x 131072 (68 GB output)"
timeout 900 python tools/bench_ops.py gram --dense --cols 131072 --rows-log2 22 --reps 1 2>&1 | tail -1 | cut -c1-400
python - <<'PY'
import time, numpy as np, scipy.sparse as sps, sys
sys.path.insert(0, '.')
import sparse_dot_amd as sda
rng = np.random.default_rng(0)
n = 1 << 18
rows = np.repeat(np.arange(n), 32); cols = rng.integers(0, n, n * 32)
a = sps.csr_matrix((rng.random(n * 32).astype(np.float32) + 0.5, (rows, cols)), shape=(n, n)); a.sum_duplicates()
b = rng.random((n, 128), dtype=np.float32)
out = np.zeros((n, 128), np.float32)
sda.dot_product_mkl(a, b, out=out, out_scalar=0.0)
t0 = time.perf_counter(); [sda.dot_product_mkl(a, b, out=out, out_scalar=0.0) for _ in range(5)]; t_call = (time.perf_counter() - t0) / 5
A = sda.to_device(a); sda.dot_product_mkl(A, b, out=out, out_scalar=0.0)
t0 = time.perf_counter(); [sda.dot_product_mkl(A, b, out=out, out_scalar=0.0) for _ in range(5)]; t_dev = (time.perf_counter() - t0) / 5
print("host-array API, 2^18 x 2^18 (8.4 M nnz) x N=128 fp32: per-call handle %.1f ms, resident DeviceMatrix %.1f ms (B and C still cross PCIe: %d MB)" % (t_call * 1e3, t_dev * 1e3, 2 * n * 128 * 4 >> 20))
PY
