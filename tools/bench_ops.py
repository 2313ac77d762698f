#!/usr/bin/env python
"""
Secondary measurements (not the driver's bench contract): SpGEMM and gram at BASELINE-config scale
on one MI355X, device-resident operands through the C ABI, each followed by size-independent
parity checks (the oracle would take minutes to hours at these sizes).

    python tools/bench_ops.py spgemm [--scale 20 --per-row 16 --kind uniform|rmat]
    python tools/bench_ops.py gram   [--rows-log2 20 --cols 16384 --per-row 64 --dense]
    python tools/bench_ops.py bsr    [--rows-log2 18 --block 4 --ncols 128]   BSR x dense: block kernel vs CSR expansion
    python tools/bench_ops.py sp2m   [--scale 20 --per-row 16]                staged product: full vs numeric-only re-run
    python tools/bench_ops.py spmv | syrk | spmmd  [--reps n]                  the workloads of bench.py's secondaries spmv / gram_sparse /
                                                                              spmmd, `reps` calls after one warm-up (counter passes)

Prints one JSON line per run.
SpGEMM checks:  C 1 == A (B 1)   (row sums, via SpMV on the same library, fp64 1e-12 rel);
                nnz(C) <= sum of per-row product counts;  no duplicate column inside a row after ordering.
Gram checks:    triu(C) 1-weighted sums:  sum_ij C_ij (j >= i)  and  diag(C) = column sums of A.^2.
"""
import argparse
import ctypes as ct
import json
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("op", choices=["spgemm", "gram", "bsr", "sp2m", "spmv", "syrk", "spmmd"])
    ap.add_argument("--block", type=int, default=4, help="bsr: block size")
    ap.add_argument("--ncols", type=int, default=128, help="bsr: dense columns")
    ap.add_argument("--scale", type=int, default=20)
    ap.add_argument("--per-row", type=int, default=None)
    ap.add_argument("--kind", default="uniform", choices=["uniform", "rmat"])
    ap.add_argument("--rows-log2", type=int, default=20)
    ap.add_argument("--cols", type=int, default=16384)
    ap.add_argument("--dense", action="store_true")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--global-mode", type=int, default=-1)
    ap.add_argument("--no-order", action="store_true", help="SpGEMM: skip the mi_sparse_order timing (needs 2 extra nnz-sized buffers)")
    ap.add_argument("--lds-parts", type=int, default=-1, help="SpGEMM: 0 disables the LDS bitmap / partitioned-class big-row path")
    ap.add_argument("--force-global", action="store_true", help="SpGEMM: send every row through the global-memory hash")
    args = ap.parse_args()

    import torch
    import bench
    import sparse_dot_amd as sda
    from sparse_dot_amd._mi_interface import MI, matrix_descr, sparse_matrix_t, _check_return_value

    dev = torch.device("cuda", 0)
    sda.mi_set_stream(torch.cuda.current_stream().cuda_stream)
    if args.force_global:
        sda.mi_set_option("spgemm_force_global", 1)
    if args.global_mode >= 0:
        sda.mi_set_option("spgemm_global_mode", args.global_mode)
    if args.lds_parts >= 0:
        sda.mi_set_option("spgemm_lds_parts", args.lds_parts)
    for kv in os.environ.get("MI_BENCH_OPTS", "").split(","):  # e.g. MI_BENCH_OPTS=spgemm_part_log2s_bias=1
        if "=" in kv:
            sda.mi_set_option(kv.split("=")[0], int(kv.split("=")[1]))

    def make(kind, n_rows_log2, ncols, per_row, seed, dtype):
        if kind == "rmat":
            ip, idx, v, n = bench.rmat_csr(torch, n_rows_log2, per_row, seed, dev)
            return ip, idx, v.to(dtype), n, n
        n = 1 << n_rows_log2
        g = torch.Generator(device=dev)
        g.manual_seed(seed)
        rows = torch.arange(n, device=dev, dtype=torch.int64).repeat_interleave(per_row)
        cols = torch.randint(0, ncols, (n * per_row,), generator=g, device=dev, dtype=torch.int64)
        key = torch.unique(rows * ncols + cols)
        r = key // ncols
        idx = (key % ncols).to(torch.int32)
        counts = torch.bincount(r, minlength=n)
        ip = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        ip[1:] = torch.cumsum(counts, 0)
        v = (torch.rand(idx.numel(), generator=g, device=dev, dtype=torch.float64) + 0.5).to(dtype)
        return ip.to(torch.int32), idx, v, n, ncols

    def handle(letter, ip, idx, v, m, k):
        h = sparse_matrix_t()
        _check_return_value(MI.call("mi_sparse_%s_create_csr" % letter, ct.byref(h), 0, m, k, ip.data_ptr(),
                                    ip.data_ptr() + 4, idx.data_ptr(), v.data_ptr()), "create")
        return h

    def dev_csr(h, dtype):
        rows, cols, nnz = ct.c_int64(), ct.c_int64(), ct.c_int64()
        _check_return_value(MI.call("mi_sparse_get_info", h, ct.byref(rows), ct.byref(cols), ct.byref(nnz), None, None), "info")
        p, c, v = ct.c_void_p(), ct.c_void_p(), ct.c_void_p()
        _check_return_value(MI.call("mi_sparse_get_device_csr", h, ct.byref(p), ct.byref(c), ct.byref(v)), "devcsr")
        return rows.value, cols.value, nnz.value, p.value, c.value, v.value

    def spmv(letter, h, x, y):
        _check_return_value(MI.call("mi_sparse_%s_mv" % letter, 10, 1.0, h, matrix_descr(), x.data_ptr(), 0.0,
                                    y.data_ptr()), "mv")

    out = {"op": args.op}
    if args.op in ("spmv", "syrk", "spmmd"):
        # bench.py's secondaries of the same names, repeated for the counter passes (bench.attach_traffic)
        abi = bench.Abi()
        if args.op == "spmv":
            ip, idx, val, n = bench.rmat_csr(torch, 20, 32, 7, dev)
            h = abi.create("s", ip, idx, val, n, n)
            x = torch.rand(n, device=dev, dtype=torch.float32)
            y = torch.empty(n, device=dev, dtype=torch.float32)
            for _ in range(args.reps + 1):
                abi.mv("s", h, x, y)
            torch.cuda.synchronize()
            abi.destroy(h)
        elif args.op == "syrk":
            m_, c_ = 1 << 20, 1 << 18
            u = bench.uniform_csr(torch, m_, 16, 5, dev, ncols=c_)
            uv = u[2].double()
            hu = abi.create("d", u[0], u[1], uv, m_, c_)
            for _ in range(args.reps + 1):
                hc = abi.handle_t()
                abi.check(MI.call("mi_sparse_syrk", 11, hu, ct.byref(hc)), "syrk")
                torch.cuda.synchronize()
                abi.destroy(hc)
            abi.destroy(hu)
        else:
            n = 1 << 14
            a = bench.uniform_csr(torch, n, 32, 11, dev)
            b = bench.uniform_csr(torch, n, 32, 12, dev)
            av, bv = a[2].double(), b[2].double()
            ha, hb = abi.create("d", a[0], a[1], av, n, n), abi.create("d", b[0], b[1], bv, n, n)
            C = torch.empty((n, n), device=dev, dtype=torch.float64)
            for _ in range(args.reps + 1):
                abi.check(MI.call("mi_sparse_d_spmmd", 10, ha, hb, 101, C.data_ptr(), n), "spmmd")
            torch.cuda.synchronize()
            abi.destroy(ha)
            abi.destroy(hb)
        out["calls"] = args.reps + 1
        print(json.dumps(out), flush=True)
        return
    if args.op == "bsr":
        # BSR x dense: 2^rows_log2 block rows, 8 random blocks per block row, bs x bs blocks, fp32,
        # row-major dense N columns.  The SAME handle through the block kernel (k_bsr_spmm) and through its CSR expansion.
        bs, N = args.block, args.ncols
        mb = 1 << args.rows_log2
        per = args.per_row or 8
        ipb, idxb, _, _, _ = make("uniform", args.rows_log2, mb, per, 5, torch.float32)
        nblocks = int(idxb.numel())
        g = torch.Generator(device=dev)
        g.manual_seed(6)
        vals = torch.rand(nblocks * bs * bs, generator=g, device=dev, dtype=torch.float32) + 0.5
        h = sparse_matrix_t()
        _check_return_value(MI.call("mi_sparse_s_create_bsr", ct.byref(h), 0, 101, mb, mb, bs, ipb.data_ptr(), ipb.data_ptr() + 4,
                                    idxb.data_ptr(), vals.data_ptr()), "create_bsr")
        m = mb * bs
        B = torch.rand((m, N), generator=g, device=dev, dtype=torch.float32) + 0.5
        C = torch.empty((m, N), device=dev, dtype=torch.float32)
        res = {}
        ref = None
        for native in (1, 0):
            sda.mi_set_option("bsr_native", native)
            sda.mi_get_counter("reset")
            for _ in range(3):
                _check_return_value(MI.call("mi_sparse_s_mm", 10, 1.0, h, matrix_descr(), 101, B.data_ptr(), N, N, 0.0, C.data_ptr(), N), "mm")
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.reps * 5):
                _check_return_value(MI.call("mi_sparse_s_mm", 10, 1.0, h, matrix_descr(), 101, B.data_ptr(), N, N, 0.0, C.data_ptr(), N), "mm")
            torch.cuda.synchronize()
            t = (time.perf_counter() - t0) / (args.reps * 5)
            used_native = sda.mi_get_counter("bsr_native_calls") > 0
            assert used_native == bool(native), (native, used_native)
            if ref is None:
                ref = C.clone()
            nnz = nblocks * bs * bs
            # algorithmic bytes: block form = values + one index per BLOCK; expansion = values + one index per VALUE
            ab = nnz * 4 + (nblocks * 4 if native else nnz * 4) + (mb + 1) * 8 + 2 * m * N * 4
            res["block_kernel" if native else "csr_expansion"] = {
                "ms": round(t * 1e3, 4), "gflops": round(2.0 * nnz * N / t / 1e9, 1), "algorithmic_bytes": ab,
                "algorithmic_GBps": round(ab / t / 1e9, 1), "frac_of_8TBps": round(ab / t / 1e9 / 8000.0, 4),
                "max_rel_diff_vs_block_kernel": float(((C - ref).abs() / ref.abs().clamp(min=1e-30)).max())}
        sda.mi_set_option("bsr_native", 1)
        # parity of the block kernel: block row sums (B = ones) against the values reduced on the host side of the GPU
        B.fill_(1.0)
        _check_return_value(MI.call("mi_sparse_s_mm", 10, 1.0, h, matrix_descr(), 101, B.data_ptr(), N, N, 0.0, C.data_ptr(), N), "mm")
        torch.cuda.synchronize()
        blk_rows = torch.repeat_interleave(torch.arange(mb, device=dev), (ipb[1:] - ipb[:-1]).long())
        want = torch.zeros((mb, bs), device=dev, dtype=torch.float64)
        want.index_add_(0, blk_rows, vals.view(nblocks, bs, bs).double().sum(2))
        err = float(((C[:, 0].double().view(mb, bs) - want).abs() / want.clamp(min=1e-30)).max())
        out.update({"config": "BSR 2^%d x 2^%d block rows, %d blocks/row, %dx%d blocks fp32 x dense %d x %d" % (args.rows_log2, args.rows_log2, per, bs, bs, m, N),
                    "blocks": nblocks, "nnz": nblocks * bs * bs, **res, "rowsum_max_rel_err": err,
                    "speedup_block_vs_expansion": round(res["csr_expansion"]["ms"] / res["block_kernel"]["ms"], 3),
                    "checks": {"rowsum_1e-5": err <= 1e-5, "variants_agree_1e-5": res["csr_expansion"]["max_rel_diff_vs_block_kernel"] <= 1e-5}})
        MI.call("mi_sparse_destroy", h)
    elif args.op == "sp2m":
        # staged product: symbolic + numeric once, then new values on the SAME pattern -> numeric only
        per_row = args.per_row or 16
        dt = torch.float64
        a = make(args.kind, args.scale, 1 << args.scale, per_row, 21 if args.kind == "rmat" else 1, dt)
        b = make(args.kind, args.scale, 1 << args.scale, per_row, 23 if args.kind == "rmat" else 2, dt)
        ha = handle("d", *a[:3], a[3], a[4])
        hb = handle("d", *b[:3], b[3], b[4])

        def sp2m(req, hc):
            _check_return_value(MI.call("mi_sparse_sp2m", 10, matrix_descr(), ha, 10, matrix_descr(), hb, req, ct.byref(hc)), "sp2m %d" % req)
        full, count, fin = [], [], []
        hc = None
        for rep in range(args.reps + 1):
            if hc is not None:
                MI.call("mi_sparse_destroy", hc)
            hc = sparse_matrix_t()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sp2m(90, hc)                       # FULL_MULT
            torch.cuda.synchronize()
            if rep:
                full.append(time.perf_counter() - t0)
        MI.call("mi_sparse_destroy", hc)
        hc = sparse_matrix_t()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sp2m(91, hc)                           # NNZ_COUNT (symbolic)
        torch.cuda.synchronize()
        count.append(time.perf_counter() - t0)
        for rep in range(args.reps + 1):
            a2 = a[2] * (rep + 2.0)
            _check_return_value(MI.call("mi_sparse_d_set_values", ha, a2.data_ptr()), "set_values")
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sp2m(92, hc)                       # FINALIZE_MULT on the unchanged pattern
            torch.cuda.synchronize()
            if rep:
                fin.append(time.perf_counter() - t0)
        # parity of the re-run: C 1 == A2 (B 1)
        m, n, nnzc, _, _, _ = dev_csr(hc, dt)
        ones = torch.ones(n, device=dev, dtype=dt)
        b1 = torch.empty(b[3], device=dev, dtype=dt)
        ab1 = torch.empty(m, device=dev, dtype=dt)
        c1 = torch.empty(m, device=dev, dtype=dt)
        spmv("d", hb, ones, b1)
        spmv("d", ha, b1, ab1)
        spmv("d", hc, ones, c1)
        torch.cuda.synchronize()
        rel = float(((c1 - ab1).abs() / ab1.abs().clamp(min=1e-300)).max())
        tf, tn = sorted(full)[len(full) // 2], sorted(fin)[len(fin) // 2]
        out.update({"config": "%s 2^%d x 2^%d, %d/row fp64 x same, staged (mi_sparse_sp2m)" % (args.kind, args.scale, args.scale, per_row),
                    "nnzC": int(nnzc), "full_mult_ms": round(tf * 1e3, 3), "nnz_count_ms": round(count[0] * 1e3, 3),
                    "finalize_only_ms": round(tn * 1e3, 3), "pattern_reuse_saves": round(1.0 - tn / tf, 3),
                    "rowsum_max_rel_err_after_set_values": rel, "checks": {"rowsum_1e-12": rel <= 1e-12}})
        for h in (ha, hb, hc):
            MI.call("mi_sparse_destroy", h)
    elif args.op == "spgemm":
        per_row = args.per_row or 16
        dt = torch.float64
        a = make(args.kind, args.scale, 1 << args.scale, per_row, 21 if args.kind == "rmat" else 1, dt)
        b = make(args.kind, args.scale, 1 << args.scale, per_row, 23 if args.kind == "rmat" else 2, dt)
        ha = handle("d", *a[:3], a[3], a[4])
        hb = handle("d", *b[:3], b[3], b[4])
        times, order_times = [], []
        hc = None
        for rep in range(args.reps + 1):
            if hc is not None:
                MI.call("mi_sparse_destroy", hc)
            hc = sparse_matrix_t()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _check_return_value(MI.call("mi_sparse_spmm", 10, ha, hb, ct.byref(hc)), "spmm")
            torch.cuda.synchronize()
            if rep:
                times.append(time.perf_counter() - t0)
            if not args.no_order and rep < args.reps:  # the last result is ordered after the row-sum check below
                t0 = time.perf_counter()
                _check_return_value(MI.call("mi_sparse_order", hc), "order")
                torch.cuda.synchronize()
                order_times.append(time.perf_counter() - t0)  # first one pays the scratch-arena growth (hipMalloc)
        t = sorted(times)[len(times) // 2]
        m, n, nnzc, _, _, _ = dev_csr(hc, dt)
        # products = sum_k colnnz_A(k) * rownnz_B(k)
        colnnz_a = torch.bincount(a[1].long(), minlength=a[4]).double()
        rownnz_b = (b[0][1:] - b[0][:-1]).double()
        products = float((colnnz_a * rownnz_b).sum())
        # parity: C 1 == A (B 1)
        ones = torch.ones(n, device=dev, dtype=dt)
        b1 = torch.empty(b[3], device=dev, dtype=dt)
        ab1 = torch.empty(m, device=dev, dtype=dt)
        c1 = torch.empty(m, device=dev, dtype=dt)
        spmv("d", hb, ones, b1)
        spmv("d", ha, b1, ab1)
        spmv("d", hc, ones, c1)
        torch.cuda.synchronize()
        rel = float(((c1 - ab1).abs() / ab1.abs().clamp(min=1e-300)).max())
        t0 = time.perf_counter()
        if not args.no_order:
            _check_return_value(MI.call("mi_sparse_order", hc), "order")
        torch.cuda.synchronize()
        order_times.append(time.perf_counter() - t0)
        t_order = min(order_times)
        nbytes = (a[1].numel() + b[1].numel() + nnzc) * 12 + 3 * (m + 1) * 8
        out.update({"config": "%s 2^%d x 2^%d, %d/row fp64 x same" % (args.kind, args.scale, args.scale, per_row),
                    "nnzA": int(a[1].numel()), "nnzB": int(b[1].numel()), "nnzC": int(nnzc), "products": products,
                    "ms": t * 1e3, "gflops": 2 * products / t / 1e9, "algorithmic_GBps": nbytes / t / 1e9,
                    "order_ms": t_order * 1e3, "order_first_call_ms": order_times[0] * 1e3, "rowsum_max_rel_err": rel,
                    "checks": {"rowsum_1e-12": rel <= 1e-12, "nnzC_le_products": nnzc <= products}})
        for h in (ha, hb, hc):
            MI.call("mi_sparse_destroy", h)
    else:
        per_row = args.per_row or 64
        dt = torch.float32
        a = make("uniform", args.rows_log2, args.cols, per_row, 3, dt)
        ha = handle("s", *a[:3], a[3], a[4])
        n = args.cols
        colsq = torch.zeros(n, device=dev, dtype=torch.float64)
        colsq.index_add_(0, a[1].long(), a[2].double() ** 2)
        lens = (a[0][1:] - a[0][:-1]).double()
        flops = float((lens * (lens + 1)).sum())  # 2 * sum r(r+1)/2
        if args.dense:
            C = torch.zeros((n, n), device=dev, dtype=dt)
            times = []
            for rep in range(args.reps + 1):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                _check_return_value(MI.call("mi_sparse_s_syrkd", 11, ha, 1.0, 0.0, C.data_ptr(), 101, n), "syrkd")
                torch.cuda.synchronize()
                if rep:
                    times.append(time.perf_counter() - t0)
            t = sorted(times)[len(times) // 2]
            diag_err = float(((torch.diagonal(C).double() - colsq).abs() / colsq.clamp(min=1e-30)).max())
            lower_zero = bool((torch.tril(C[:2048, :2048], -1) == 0).all())
            # total: sum of upper triangle == sum over rows of (sum_{p<=q} v_p v_q)
            out.update({"config": "uniform 2^%d x %d, %d/row fp32, dense out" % (args.rows_log2, n, per_row),
                        "nnzA": int(a[1].numel()), "ms": t * 1e3, "gflops": flops / t / 1e9,
                        "algorithmic_GBps": (a[1].numel() * 8 + n * n * 4 / 2) / t / 1e9,
                        "diag_max_rel_err": diag_err, "checks": {"diag_1e-5": diag_err <= 1e-5, "lower_zero": lower_zero}})
        else:
            times = []
            hc = None
            for rep in range(args.reps + 1):
                if hc is not None:
                    MI.call("mi_sparse_destroy", hc)
                hc = sparse_matrix_t()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                _check_return_value(MI.call("mi_sparse_syrk", 11, ha, ct.byref(hc)), "syrk")
                torch.cuda.synchronize()
                if rep:
                    times.append(time.perf_counter() - t0)
            t = sorted(times)[len(times) // 2]
            m, _, nnzc, _, _, _ = dev_csr(hc, dt)
            out.update({"config": "uniform 2^%d x %d, %d/row fp32, sparse out" % (args.rows_log2, n, per_row),
                        "nnzA": int(a[1].numel()), "nnzC": int(nnzc), "ms": t * 1e3, "gflops": flops / t / 1e9})
            MI.call("mi_sparse_destroy", hc)
        MI.call("mi_sparse_destroy", ha)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
