#!/usr/bin/env python
"""Developer tool: A/B the SpMM kernel variants on the headline matrix in ONE process.

    python tools/spmm_sweep.py [--workload rmat|uniform] [--launches K] [--variants "S:hot_kb:chunk,..."]

Every variant = spmm_slices:spmm_hot_kb:spmm_chunk[:option=value ...] (any mi_set_option knob, e.g. spmm_unroll=8).  For each one the plan is rebuilt (untimed), the
product is checked against the first variant's result, and K launches are timed with the hipEvents the
library records around the main kernel (profile_events).  Under `rocprofv3 --kernel-trace --pmc ...` the same
script gives per-dispatch counters: the k_spmm dispatches appear in the order printed here, K + 1 per
variant (1 warm-up + K), which tools/pmc_sweep_summary.py uses to attribute them.
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
    ap.add_argument("--launches", type=int, default=5)
    ap.add_argument("--variants", default="")
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--adopt-tags", action="store_true",
                    help="library defaults end to end: no synchronous plan; untimed calls first until the hot / cold tags are adopted")
    args = ap.parse_args()
    import torch
    import bench
    import sparse_dot_amd as sda
    from sparse_dot_amd._mi_interface import MI, matrix_descr, sparse_matrix_t, _check_return_value

    dev = torch.device("cuda", 0)
    sda.mi_set_stream(torch.cuda.current_stream().cuda_stream)
    if not args.adopt_tags:
        sda.mi_set_option("spmm_plan_sync", 1)  # A/B tool: every launch of a variant runs with its final plan
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
    if args.variants:
        variants = [tuple(int(x) if "=" not in x else x for x in v.split(":")) for v in args.variants.split(",")]
    else:
        variants = [(s, h, 256) for s in (1, 2, 4, 8) for h in (0, 2048, 3072, 4096, 8192)]
    alg = nnz * (4 + vals.element_size()) + (n + 1) * 8 + 2 * n * N * vals.element_size()
    ref = None
    print(json.dumps({"workload": args.workload, "n": n, "nnz": nnz, "N": N, "dtype": args.dtype,
                      "launches_per_variant": args.launches + 1, "algorithmic_bytes": alg}), flush=True)
    for var in variants:
        s, hot, chunk = var[:3]
        extra = dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in var[3:])
        for name, value in extra.items():
            sda.mi_set_option(name, value)
        sda.mi_set_option("spmm_slices", s)
        sda.mi_set_option("spmm_hot_kb", hot)
        sda.mi_set_option("spmm_chunk", chunk)
        h = sparse_matrix_t()
        _check_return_value(MI.call("mi_sparse_%s_create_csr" % letter, ct.byref(h), 0, n, n, indptr.data_ptr(),
                                    indptr.data_ptr() + 4, indices.data_ptr(), vals.data_ptr()), "create")

        def step():
            r = MI.call("mi_sparse_%s_mm" % letter, 10, 1.0, h, matrix_descr(), 101, B.data_ptr(), N, N, 0.0,
                        C.data_ptr(), N)
            if r:
                _check_return_value(r, "mm")
        C.zero_()
        step()  # plan + warm-up launch
        torch.cuda.synchronize()
        if args.adopt_tags:
            for _ in range(4):  # the analysis runs behind call 2 and is adopted by a later call
                step()
                torch.cuda.synchronize()
        if ref is None:
            ref = C.clone()
            err = 0.0
        else:
            err = float(((C - ref).abs() / ref.abs().clamp(min=1e-30)).max())
        sda.mi_set_option("profile_events", 1)
        sda.mi_get_counter("reset")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.launches):
            step()
        e1.record()
        torch.cuda.synchronize()
        k_ms = sda.mi_get_counter("spmm_kernel_ms") / max(1.0, sda.mi_get_counter("spmm_kernel_launches"))
        sda.mi_set_option("profile_events", 0)
        print(json.dumps({"slices": s, "hot_kb": hot, "chunk": chunk, "extra": extra, "kernel_ms": round(k_ms, 4),
                          "step_ms_with_event_sync": round(e0.elapsed_time(e1) / args.launches, 4),
                          "tagged": int(sda.mi_get_counter("spmm_last_tagged")),
                          "hot_coverage": round(sda.mi_get_counter("spmm_hot_coverage"), 4),
                          "alg_GBps": round(alg / k_ms / 1e6, 1), "max_rel_diff_vs_first": err}), flush=True)
        MI.call("mi_sparse_destroy", h)
        for name in extra:
            sda.mi_set_option(name, 0)


if __name__ == "__main__":
    main()
