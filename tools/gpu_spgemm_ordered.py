"""Developer tool (round 5): a sparse x sparse product whose rows must come back ordered (the reference's reorder_output=True):
mi_sparse_spmm + mi_sparse_order against mi_sparse_spmm_ordered (long rows accumulated by rank, mi_sparse_order skips them).

    python tools/gpu_spgemm_ordered.py [scale=18]        two R-MAT 2^scale, 16 edges/row, fp64; 3 calls each, ms"""
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
def mk(ipt, idx, v):
    h = sparse_matrix_t(); _check_return_value(MI.call("mi_sparse_d_create_csr", ct.byref(h), 0, n, n, ipt.data_ptr(), ipt.data_ptr() + 4, idx.data_ptr(), v.data_ptr()), "create"); return h
ha, hb = mk(a[0], a[1], av), mk(b[0], b[1], bv)
ref = None
for name in ("spmm + order", "spmm_ordered", "spmm + order", "spmm_ordered"):
    ts = []
    for rep in range(3):
        hc = sparse_matrix_t(); torch.cuda.synchronize(); t0 = time.perf_counter()
        if name == "spmm_ordered":
            _check_return_value(MI.call("mi_sparse_spmm_ordered", 10, ha, hb, ct.byref(hc)), "spmm_ordered")
        else:
            _check_return_value(MI.call("mi_sparse_spmm", 10, ha, hb, ct.byref(hc)), "spmm")
            _check_return_value(MI.call("mi_sparse_order", hc), "order")
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        if rep == 2 and scale <= 18:  # same ordered pattern either way, values to 1e-12
            r_, c_, z_ = ct.c_int64(), ct.c_int64(), ct.c_int64()
            _check_return_value(MI.call("mi_sparse_get_info", hc, ct.byref(r_), ct.byref(c_), ct.byref(z_), None, None), "info")
            nnz = z_.value
            ptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
            col = torch.empty(nnz, dtype=torch.int32, device=dev); val = torch.empty(nnz, dtype=torch.float64, device=dev)
            _check_return_value(MI.call("mi_sparse_copy_out", hc, 0, 4, ptr.data_ptr(), col.data_ptr(), val.data_ptr()), "copy_out")
            torch.cuda.synchronize()
            if ref is None:
                ref = (col.clone(), val.clone())
            else:
                same = bool(torch.equal(col, ref[0])); err = float(((val - ref[1]).abs() / ref[1].abs()).max())
                print(json.dumps({"same_columns_as_first": same, "max_rel_value_diff": err}), flush=True)
        MI.call("mi_sparse_destroy", hc)
    print(json.dumps({"case": name, "scale": scale, "ms": [round(t, 2) for t in ts]}), flush=True)
