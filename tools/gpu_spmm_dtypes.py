# SpMM throughput across value types / widths on the headline matrix (device pointers through the C ABI)
import ctypes as ct, sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench
import sparse_dot_amd as sda
from sparse_dot_amd._mi_interface import MI, matrix_descr, sparse_matrix_t, _check_return_value
dev = torch.device("cuda", 0)
sda.mi_set_stream(torch.cuda.current_stream().cuda_stream)
ip, idx, val, n = bench.rmat_csr(torch, 20, 32, 7, dev)
nnz = idx.numel()
for letter, dt, vb in (("s", torch.float32, 4), ("d", torch.float64, 8), ("c", torch.complex64, 8), ("z", torch.complex128, 16)):
    v = val.to(dt)
    h = sparse_matrix_t()
    _check_return_value(MI.call("mi_sparse_%s_create_csr" % letter, ct.byref(h), 0, n, n, ip.data_ptr(), ip.data_ptr() + 4, idx.data_ptr(), v.data_ptr()), "create")
    for N in (32, 128):
        B = torch.rand(n, N, device=dev, dtype=torch.float64).to(dt); C = torch.empty(n, N, device=dev, dtype=dt)
        one, zero = (1.0, 0.0)
        if letter in "cz":
            cx = MI.fn["mi_sparse_%s_mm" % letter].argtypes[1]
            one, zero = cx(1 + 0j), cx(0j)
        def mm(): _check_return_value(MI.call("mi_sparse_%s_mm" % letter, 10, one, h, matrix_descr(), 101, B.data_ptr(), N, N, zero, C.data_ptr(), N), "mm")
        for _ in range(3): mm()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): mm()
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 10
        gathered = nnz * N * vb
        print("SpMM %s N=%3d: %.3f ms, %.0f GFLOP/s (real-op count x%d), gathered B bytes %.1f GB -> %.2f TB/s" % (letter, N, t * 1e3, 2 * nnz * N * (4 if letter in "cz" else 1) / t / 1e9, 4 if letter in "cz" else 1, gathered / 1e9, gathered / t / 1e12))
        del B, C
    MI.call("mi_sparse_destroy", h)
