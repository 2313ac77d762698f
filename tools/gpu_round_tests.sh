cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "rccl" 2>&1 | tail -12
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest.log
python - <<'PY'
# SpMV throughput (f1): N = 1 through the same kernel
import ctypes as ct, sys, time
sys.path.insert(0, '.')
import torch, bench
import sparse_dot_amd as sda
from sparse_dot_amd._mi_interface import MI, matrix_descr, sparse_matrix_t, _check_return_value
dev = torch.device("cuda", 0)
sda.mi_set_stream(torch.cuda.current_stream().cuda_stream)
ip, idx, val, n = bench.rmat_csr(torch, 20, 32, 7, dev)
h = sparse_matrix_t()
_check_return_value(MI.call("mi_sparse_s_create_csr", ct.byref(h), 0, n, n, ip.data_ptr(), ip.data_ptr() + 4, idx.data_ptr(), val.data_ptr()), "create")
x = torch.rand(n, device=dev); y = torch.empty(n, device=dev)
def mv(): _check_return_value(MI.call("mi_sparse_s_mv", 10, 1.0, h, matrix_descr(), x.data_ptr(), 0.0, y.data_ptr()), "mv")
for _ in range(3): mv()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): mv()
torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 20
nnz = idx.numel()
print("SpMV R-MAT 2^20 (%d nnz) fp32: %.3f ms, %.1f GFLOP/s, %.1f GB/s algorithmic" % (nnz, t * 1e3, 2 * nnz / t / 1e9, (nnz * 8 + n * 16) / t / 1e9))
PY
