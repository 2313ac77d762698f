# SpMV throughput (f1): N = 1, lanes over nonzeros; sweep of the work-chunk size
import ctypes as ct, sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench
import sparse_dot_amd as sda
from sparse_dot_amd._mi_interface import MI, matrix_descr, sparse_matrix_t, _check_return_value
dev = torch.device("cuda", 0)
sda.mi_set_stream(torch.cuda.current_stream().cuda_stream)
for kv in os.environ.get("MI_BENCH_OPTS", "").split(","):  # e.g. MI_BENCH_OPTS=spmv_rows=0
    if "=" in kv:
        sda.mi_set_option(kv.split("=")[0], int(kv.split("=")[1]))
for kind in ("rmat", "uniform"):
    ip, idx, val, n = bench.rmat_csr(torch, 20, 32, 7, dev) if kind == "rmat" else bench.uniform_csr(torch, 1 << 20, 32, 7, dev)
    x = torch.rand(n, device=dev); y = torch.empty(n, device=dev)
    nnz = idx.numel()
    for chunk in (256, 512, 1024):
        sda.mi_set_option("spmm_chunk", chunk)
        h = sparse_matrix_t()
        _check_return_value(MI.call("mi_sparse_s_create_csr", ct.byref(h), 0, n, n, ip.data_ptr(), ip.data_ptr() + 4, idx.data_ptr(), val.data_ptr()), "create")
        def mv(): _check_return_value(MI.call("mi_sparse_s_mv", 10, 1.0, h, matrix_descr(), x.data_ptr(), 0.0, y.data_ptr()), "mv")
        for _ in range(3): mv()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): mv()
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 20
        print("SpMV %s 2^20 (%d nnz) fp32 chunk %4d: %.3f ms, %.1f GFLOP/s, %.1f GB/s algorithmic" % (kind, nnz, chunk, t * 1e3, 2 * nnz / t / 1e9, (nnz * 8 + n * 16) / t / 1e9))
        MI.call("mi_sparse_destroy", h)
sda.mi_set_option("spmm_chunk", 256)
