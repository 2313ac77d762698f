#!/usr/bin/env python
"""Developer tool (round 5): the column-partitioned SpMM (SpmmKpart, csrc/spmm.hip) against the row-owned kernel on the
headline matrix, in ONE process.

    python tools/gpu_kpart.py [--workload rmat|uniform] [--scale 20] [--ncols 128] [--launches 10]
                              [--variants "off,32:8,64:8,128:8,32:4"]      (min_row:parts[+option=value...])

Every variant: a fresh handle, library defaults (plans adopted as a user's handle would adopt them: untimed calls first),
whole-product time from events around K calls (all kernels of a product: short rows, long rows, fix-ups, combine), the
result against the first variant's and against an fp64 evaluation of sampled rows.  Under rocprofv3 the kernel trace /
counters of the same run give the per-kernel split.
"""
import argparse
import ctypes as ct
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="rmat")
    ap.add_argument("--scale", type=int, default=20)
    ap.add_argument("--ncols", type=int, default=128)
    ap.add_argument("--launches", type=int, default=10)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--variants", default="off,32:8,64:8,128:8,32:4,64:4")
    ap.add_argument("--warm", type=int, default=8)
    args = ap.parse_args()
    import torch
    import bench
    import sparse_dot_amd as sda
    from sparse_dot_amd._mi_interface import MI, matrix_descr, sparse_matrix_t, _check_return_value

    dev = torch.device("cuda", 0)
    sda.mi_set_stream(torch.cuda.current_stream().cuda_stream)
    if args.workload == "rmat":
        indptr, indices, vals, n = bench.rmat_csr(torch, args.scale, 32, 7, dev)
    else:
        indptr, indices, vals, n = bench.uniform_csr(torch, 1 << args.scale, 32, 7, dev)
    tdt = torch.float32 if args.dtype == "f32" else torch.float64
    letter = "s" if args.dtype == "f32" else "d"
    vals = vals.to(tdt)
    N = args.ncols
    g = torch.Generator(device=dev)
    g.manual_seed(9)
    B = torch.rand((n, N), generator=g, device=dev, dtype=tdt)
    C = torch.empty((n, N), device=dev, dtype=tdt)
    nnz = int(indices.numel())
    alg = nnz * (4 + vals.element_size()) + (n + 1) * 8 + 2 * n * N * vals.element_size()
    print(json.dumps({"workload": args.workload, "n": n, "nnz": nnz, "N": N, "dtype": args.dtype, "algorithmic_bytes": alg}), flush=True)
    # fp64 evaluation of sampled rows (the longest ones included)
    ip = indptr.to(torch.int64)
    lens = ip[1:] - ip[:-1]
    sample = torch.cat([torch.topk(lens, 8).indices, torch.randint(0, n, (120,), device=dev)])
    want = torch.zeros((sample.numel(), N), device=dev, dtype=torch.float64)
    for k, r in enumerate(sample.tolist()):
        b, e = int(ip[r]), int(ip[r + 1])
        if e > b:
            want[k] = (vals[b:e].double()[:, None] * B[indices[b:e].long()].double()).sum(0)
    ref = None
    marker = torch.zeros(1, device=dev)
    defaults = {"spmm_chunk": 256, "spmm_kpart_chunk": 128, "spmm_slices": 0, "spmm_hot_kb": 8192, "spmm_unroll": 4}
    for full in args.variants.split(","):
        var, *extra = full.split("+")
        for name, value in defaults.items():
            sda.mi_set_option(name, value)
        for kv in extra:
            sda.mi_set_option(kv.split("=")[0], int(kv.split("=")[1]))
        if var == "off":
            sda.mi_set_option("spmm_kpart", 0)
        elif var == "default":  # the library's own choice (what bench.py times)
            sda.mi_set_option("spmm_kpart", 1)
            sda.mi_set_option("spmm_kpart_min_row", 64)
            sda.mi_set_option("spmm_kpart_parts", 8)
        else:
            t, p = var.split(":")
            sda.mi_set_option("spmm_kpart", 1)
            sda.mi_set_option("spmm_kpart_min_row", int(t))
            sda.mi_set_option("spmm_kpart_parts", int(p))
        h = sparse_matrix_t()
        _check_return_value(MI.call("mi_sparse_%s_create_csr" % letter, ct.byref(h), 0, n, n, indptr.data_ptr(),
                                    indptr.data_ptr() + 4, indices.data_ptr(), vals.data_ptr()), "create")

        def step():
            r = MI.call("mi_sparse_%s_mm" % letter, 10, 1.0, h, matrix_descr(), 101, B.data_ptr(), N, N, 0.0, C.data_ptr(), N)
            if r:
                _check_return_value(r, "mm")
        C.fill_(float("nan"))
        sda.mi_get_counter("reset")
        for _ in range(args.warm):
            step()
            torch.cuda.synchronize()
        got = C[sample].double()
        err64 = float(((got - want).abs() / want.abs().clamp(min=1e-30)).max())
        if ref is None:
            ref = C.clone()
            err = 0.0
        else:
            err = float(((C - ref).abs() / ref.abs().clamp(min=1e-30)).max())
        nan = int(torch.isnan(C).sum())
        marker.fill_(1.0)  # a one-element fill: the dispatches after the LAST of these are the timed products (bench.py's PMC passes)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.launches):
            step()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.launches
        print(json.dumps({"variant": full, "product_ms": round(ms, 4), "alg_GBps": round(alg / ms / 1e6, 1),
                          "frac_of_8TBps": round(alg / ms / 1e6 / 8000.0, 4),
                          "kpart": int(sda.mi_get_counter("spmm_last_kpart")),
                          "long_share": round(sda.mi_get_counter("spmm_kpart_long_share"), 4),
                          "build_ms": round(sda.mi_get_counter("spmm_kpart_build_ms"), 3),
                          "tagged_short": int(sda.mi_get_counter("spmm_last_tagged")),
                          "max_rel_vs_fp64_sample": err64, "max_rel_vs_first": err, "nan": nan}), flush=True)
        MI.call("mi_sparse_destroy", h)


if __name__ == "__main__":
    main()
