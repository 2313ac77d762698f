"""Experiment: SpMM as two passes, hot columns first (their B rows stay L2 resident), then the rest
with beta = 1.  Splits A with torch on the device, uses the unmodified library for both passes."""
import ctypes as ct, os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch, bench
import sparse_dot_amd as sda
from sparse_dot_amd._mi_interface import MI, matrix_descr, sparse_matrix_t, _check_return_value

dev = torch.device("cuda", 0)
sda.mi_set_stream(torch.cuda.current_stream().cuda_stream)
wl = sys.argv[1] if len(sys.argv) > 1 else "rmat"
indptr, indices, vals, n = bench.rmat_csr(torch, 20, 32, 7, dev) if wl == "rmat" else bench.uniform_csr(torch, 1 << 20, 32, 7, dev)
N = 128
B = torch.rand((n, N), device=dev) + 0.5
C = torch.empty((n, N), device=dev)
Cref = torch.empty((n, N), device=dev)

def handle(ip, idx, v):
    h = sparse_matrix_t()
    _check_return_value(MI.call("mi_sparse_s_create_csr", ct.byref(h), 0, n, n, ip.data_ptr(), ip.data_ptr() + 4, idx.data_ptr(), v.data_ptr()), "create")
    return h

def mm(h, out, beta):
    _check_return_value(MI.call("mi_sparse_s_mm", 10, 1.0, h, matrix_descr(), 101, B.data_ptr(), N, N, beta, out.data_ptr(), N), "mm")

def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3

h_all = handle(indptr, indices, vals)
t_all = timeit(lambda: mm(h_all, Cref, 0.0))
print("single pass: %.3f ms" % t_all)
counts = torch.bincount(indices.long(), minlength=n)
order = torch.argsort(counts, descending=True)
rows = torch.repeat_interleave(torch.arange(n, device=dev), (indptr[1:] - indptr[:-1]).long())
keep = []
for H in (512, 1024, 2048, 4096, 6144, 8192, 16384, 65536):
    hot_col = torch.zeros(n, dtype=torch.bool, device=dev); hot_col[order[:H]] = True
    is_hot = hot_col[indices.long()]
    parts = []
    for mask in (is_hot, ~is_hot):
        r = rows[mask]; ip = torch.zeros(n + 1, dtype=torch.int64, device=dev); ip[1:] = torch.cumsum(torch.bincount(r, minlength=n), 0)
        parts.append((ip.to(torch.int32), indices[mask].contiguous(), vals[mask].contiguous()))
    keep.append(parts)
    hh, hc = handle(*parts[0]), handle(*parts[1])
    frac = float(is_hot.float().mean())
    t_hot = timeit(lambda: mm(hh, C, 0.0)); t_cold = timeit(lambda: mm(hc, C, 1.0))
    def both(): mm(hh, C, 0.0); mm(hc, C, 1.0)
    t_both = timeit(both)
    err = float(((C - Cref).abs() / Cref.abs().clamp(min=1e-30)).max())
    print("H=%6d hot nnz %.1f%%: hot %.3f ms cold %.3f ms both %.3f ms (single %.3f) maxrel vs single %.2e" % (H, 100 * frac, t_hot, t_cold, t_both, t_all, err))
    MI.call("mi_sparse_destroy", hh); MI.call("mi_sparse_destroy", hc)
