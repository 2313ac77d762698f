import sys, time, ctypes as ct, json
sys.path.insert(0, "/root/repo")
import torch, bench
import sparse_dot_amd as sda
from sparse_dot_amd._mi_interface import MI, sparse_matrix_t, _check_return_value
dev = torch.device("cuda", 0)
sda.mi_set_stream(torch.cuda.current_stream().cuda_stream)
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 18
a = bench.rmat_csr(torch, scale, 16, 21, dev); b = bench.rmat_csr(torch, scale, 16, 23, dev)
n = a[3]; av, bv = a[2].double(), b[2].double()
# shuffle the entries inside every row of B: sort by (row, random key)
ip = b[0].to(torch.int64); rows = torch.repeat_interleave(torch.arange(n, device=dev), ip[1:] - ip[:-1])
key = rows.double() + torch.rand(rows.numel(), device=dev, dtype=torch.float64) * 0.999
o = torch.argsort(key); bidx_s, bval_s = b[1][o].contiguous(), bv[o].contiguous()
def mk(ipt, idx, v):
    h = sparse_matrix_t(); _check_return_value(MI.call("mi_sparse_d_create_csr", ct.byref(h), 0, n, n, ipt.data_ptr(), ipt.data_ptr() + 4, idx.data_ptr(), v.data_ptr()), "create"); return h
ha = mk(a[0], a[1], av)
for name, hb, opt in (("sorted B", mk(b[0], b[1], bv), 1), ("shuffled B, sort on ingest", mk(b[0], bidx_s, bval_s), 1), ("shuffled B, global hash", mk(b[0], bidx_s, bval_s), 0)):
    sda.mi_set_option("spgemm_sort_ingest", opt)
    ts = []
    for rep in range(3):
        hc = sparse_matrix_t(); torch.cuda.synchronize(); t0 = time.perf_counter()
        _check_return_value(MI.call("mi_sparse_spmm", 10, ha, hb, ct.byref(hc)), "spmm"); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3); MI.call("mi_sparse_destroy", hc)
    print(json.dumps({"case": name, "scale": scale, "ms": [round(t, 2) for t in ts]}), flush=True)
