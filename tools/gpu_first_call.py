#!/usr/bin/env python
"""What a ONE-SHOT caller pays: the first mi_sparse_spmm of a fresh process on the literal BASELINE configs[2]
(two R-MAT 2^20 x 2^20, 16 edges/row, fp64 -> 9.7e9 entries, 116 GB of result arrays) against the second call.

Why a process of its own: a device block that was hipFree'd and is allocated again costs seconds on this driver
(tools/probes/alloc_probe.hip: hipMalloc of 78 GiB 0.25 ms fresh, 5.8 s right after a hipFree of the same size), so a
"first call" timed inside a long-lived process that has already released large blocks (bench.py's secondaries run after
the headline workload was freed) measures the driver's recycling, not the library.  Prints one JSON line."""
import ctypes as ct
import json
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    import sparse_dot_amd as sda
    from sparse_dot_amd._mi_interface import MI, sparse_matrix_t, _check_return_value
    dev = torch.device("cuda", 0)
    sda.mi_set_stream(torch.cuda.current_stream().cuda_stream)
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    a = bench.rmat_csr(torch, scale, 16, 21, dev)
    b = bench.rmat_csr(torch, scale, 16, 23, dev)
    av, bv = a[2].double(), b[2].double()
    n = a[3]

    def mk(ip, idx, v):
        h = sparse_matrix_t()
        _check_return_value(MI.call("mi_sparse_d_create_csr", ct.byref(h), 0, n, n, ip.data_ptr(), ip.data_ptr() + 4, idx.data_ptr(),
                                    v.data_ptr()), "create")
        return h
    ha, hb = mk(a[0], a[1], av), mk(b[0], b[1], bv)
    torch.cuda.synchronize()
    times = []
    nnz = 0
    for rep in range(3):
        hc = sparse_matrix_t()
        t0 = time.perf_counter()
        _check_return_value(MI.call("mi_sparse_spmm", 10, ha, hb, ct.byref(hc)), "spmm")
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
        r, c, z = ct.c_int64(), ct.c_int64(), ct.c_int64()
        MI.call("mi_sparse_get_info", hc, ct.byref(r), ct.byref(c), ct.byref(z), None, None)
        nnz = z.value
        MI.call("mi_sparse_destroy", hc)  # the result's blocks go to the library's cache, not back to the driver
    print(json.dumps({"workload": "R-MAT scale %d x same, 16 edges/row fp64, mi_sparse_spmm, fresh process" % scale, "nnzC": nnz,
                      "first_call_ms": round(times[0], 2), "second_call_ms": round(times[1], 2), "third_call_ms": round(times[2], 2),
                      "first_over_steady": round(times[0] / min(times[1:]), 2)}), flush=True)


if __name__ == "__main__":
    main()
