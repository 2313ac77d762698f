#!/usr/bin/env python
"""Developer tool (round 5): the 1-D row blocks of the headline matrix that `bench.py --gpus N` hands to the ranks, one after the
other on ONE GPU -- row-owned kernel vs column-partitioned long rows per block (does the partition still pay on 1 / N of the rows?).
    python tools/gpu_kpart_blocks.py [--world 8] [--scale 20] [--ncols 128]"""
import argparse
import ctypes as ct
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--scale", type=int, default=20)
    ap.add_argument("--ncols", type=int, default=128)
    args = ap.parse_args()
    import torch
    import bench
    import sparse_dot_amd as sda
    from sparse_dot_amd import distributed as D
    from sparse_dot_amd._mi_interface import MI, matrix_descr, sparse_matrix_t, _check_return_value
    dev = torch.device("cuda", 0)
    sda.mi_set_stream(torch.cuda.current_stream().cuda_stream)
    indptr, indices, vals, n = bench.rmat_csr(torch, args.scale, 32, 7, dev)
    N = args.ncols
    B = torch.rand((n, N), device=dev, dtype=torch.float32)
    ip64 = indptr.to(torch.int64)
    bounds = D.partition_rows(ip64.cpu().numpy(), args.world, dense_bytes=n * N * 4)
    tot = {"off": 0.0, "kpart": 0.0, "default": 0.0}
    for r in range(args.world):
        r0, r1 = int(bounds[r]), int(bounds[r + 1])
        lo, hi = int(ip64[r0]), int(ip64[r1])
        bp = (ip64[r0:r1 + 1] - lo).to(torch.int32).contiguous()
        bi, bv = indices[lo:hi].contiguous(), vals[lo:hi].contiguous()
        C = torch.empty((r1 - r0, N), device=dev, dtype=torch.float32)
        row = {"block": r, "rows": r1 - r0, "nnz": hi - lo}
        for name, opt in (("off", 0), ("kpart", 2), ("default", 1)):
            sda.mi_set_option("spmm_kpart", opt)
            h = sparse_matrix_t()
            _check_return_value(MI.call("mi_sparse_s_create_csr", ct.byref(h), 0, r1 - r0, n, bp.data_ptr(), bp.data_ptr() + 4,
                                        bi.data_ptr(), bv.data_ptr()), "create")

            def step():
                _check_return_value(MI.call("mi_sparse_s_mm", 10, 1.0, h, matrix_descr(), 101, B.data_ptr(), N, N, 0.0, C.data_ptr(), N), "mm")
            for _ in range(8):
                step()
                torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                step()
            e1.record()
            torch.cuda.synchronize()
            row[name + "_ms"] = round(e0.elapsed_time(e1) / 20, 4)
            row[name + "_parts"] = int(sda.mi_get_counter("spmm_last_kpart"))
            tot[name] = max(tot[name], row[name + "_ms"])
            MI.call("mi_sparse_destroy", h)
        row["long_share"] = round(sda.mi_get_counter("spmm_kpart_long_share"), 3)
        print(json.dumps(row), flush=True)
    sda.mi_set_option("spmm_kpart", 1)
    print(json.dumps({"world": args.world, "slowest_block_ms": tot}))


if __name__ == "__main__":
    main()
