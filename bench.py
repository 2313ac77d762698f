#!/usr/bin/env python
"""
bench.py -- headline benchmark of the hot path: fp32 CSR x dense SpMM on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload rmat|uniform] [--no-cpu]

Workload (BASELINE.json configs[1]): R-MAT CSR 2^20 x 2^20 (scale 20, 32 edges/row drawn,
a,b,c,d = .57,.19,.19,.05, duplicates merged -> ~31.4 M nnz, values U[0.5,1.5)) times a dense
2^20 x 128 fp32 matrix, synthetic, generated on the GPU.  A "step" is one mi_sparse_s_mm call
(C := A @ B) through the C ABI with A, B, C resident in HBM (device pointers, zero copy).

One JSON line is printed by rank 0:
  value        = effective GFLOP/s = 2 * nnz * N * n_gpus / step time   (whole job)
  roofline     = algorithmic bytes of one launch (nnz*(4+4) + (M+1)*8 + K*N*4 + M*N*4;
                 the row pointer is 8 bytes per row in this build) / mean duration of the dominant
                 kernel (k_spmm) measured with hipEvents on the launch stream, vs the 8 TB/s HBM peak
  cpu_baseline = the same SpMM on the host: MKL's mkl_sparse_s_mm through oracle/mkl_shim.py when a
                 libmkl_rt is discoverable (kind "reference": MKL is the reference's arithmetic
                 engine), else the oracle's OpenMP port (kind "port").

Multi-GPU (--gpus N, launched by torch.distributed.run, one rank per GPU): weak scaling by 1-D row
blocks -- rank r owns its own 2^20-row block of a (N * 2^20) x 2^20 matrix (different R-MAT
seed); B is replicated with one RCCL broadcast at set-up; every rank's output row block stays on
the rank that produced it.  The timed step is the local SpMM (no data-path collective).  The cost
of the two collectives the north_star names -- broadcast(B) and all-gatherv(C) -- is measured
separately and reported under "collectives" (they are bandwidth-bound on xGMI and ~10x the kernel
time at this size; DESIGN.md section "Multi-GPU").
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def rmat_csr(torch, scale, edges_per_row, seed, device, abcd=(0.57, 0.19, 0.19, 0.05)):
    """R-MAT edge list -> de-duplicated, sorted CSR (int32 indptr / indices, fp32 values U[0.5,1.5))."""
    n = 1 << scale
    ne = n * edges_per_row
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    a, b, c, _ = abcd
    rows = torch.zeros(ne, dtype=torch.int64, device=device)
    cols = torch.zeros(ne, dtype=torch.int64, device=device)
    for _level in range(scale):
        r = torch.rand(ne, generator=g, device=device)
        row_bit = r >= (a + b)
        col_bit = ((r >= a) & (r < a + b)) | (r >= a + b + c)
        rows = rows * 2 + row_bit.to(torch.int64)
        cols = cols * 2 + col_bit.to(torch.int64)
        del r, row_bit, col_bit
    key = torch.unique(rows * n + cols)  # sorted, duplicates merged
    del rows, cols
    r = key // n
    indices = (key % n).to(torch.int32)
    del key
    counts = torch.bincount(r, minlength=n)
    indptr = torch.zeros(n + 1, dtype=torch.int64, device=device)
    indptr[1:] = torch.cumsum(counts, 0)
    vals = torch.rand(indices.numel(), generator=g, device=device, dtype=torch.float32) + 0.5
    return indptr.to(torch.int32), indices, vals, n


def uniform_csr(torch, n, per_row, seed, device):
    """`per_row` random distinct-ish columns per row (duplicates merged), sorted."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    rows = torch.arange(n, device=device, dtype=torch.int64).repeat_interleave(per_row)
    cols = torch.randint(0, n, (n * per_row,), generator=g, device=device, dtype=torch.int64)
    key = torch.unique(rows * n + cols)
    r = key // n
    indices = (key % n).to(torch.int32)
    counts = torch.bincount(r, minlength=n)
    indptr = torch.zeros(n + 1, dtype=torch.int64, device=device)
    indptr[1:] = torch.cumsum(counts, 0)
    vals = torch.rand(indices.numel(), generator=g, device=device, dtype=torch.float32) + 0.5
    return indptr.to(torch.int32), indices, vals, n


def cpu_baseline(indptr, indices, vals, n, bmat, nrep=3):
    """Time the same SpMM on the host cores.  Returns the cpu_baseline JSON object."""
    import numpy as np
    import scipy.sparse as sps

    a = sps.csr_matrix((vals, indices, indptr), shape=(n, n))
    flops = 2.0 * a.nnz * bmat.shape[1]
    out = np.zeros((n, bmat.shape[1]), dtype=np.float32)  # preallocated: no first-touch faults in the timing
    try:
        from oracle import mkl_shim
        mkl = mkl_shim.MklSpmm()
        h = mkl.make(a)
        mkl.mm(h, bmat, out)  # warm-up (MKL's first call is slow)
        ts = []
        for _ in range(nrep):
            t0 = time.perf_counter()
            mkl.mm(h, bmat, out)
            ts.append(time.perf_counter() - t0)
        mkl.destroy(h)
        t = sorted(ts)[len(ts) // 2]
        return {"value": flops / t / 1e9, "unit": "GFLOP/s", "cores": mkl.threads(), "kind": "reference",
                "sample": "full workload (%d nnz x N=%d), median of %d mkl_sparse_s_mm calls via oracle/mkl_shim.py, "
                          "preallocated output; %s; host has %d logical cpus"
                          % (a.nnz, bmat.shape[1], nrep, mkl.version(), os.cpu_count()),
                "ms": t * 1e3}
    except Exception as e:  # no MKL on this box: fall back to the oracle's OpenMP port
        note = "libmkl_rt unavailable (%s)" % (str(e)[:80],)
    from oracle import cpu_oracle
    cpu_oracle.spmm(a[:1024], bmat)  # build + warm
    ts = []
    for _ in range(nrep):
        t0 = time.perf_counter()
        cpu_oracle.spmm(a, bmat)
        ts.append(time.perf_counter() - t0)
    t = sorted(ts)[len(ts) // 2]
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count()
    return {"value": flops / t / 1e9, "unit": "GFLOP/s", "cores": cores, "kind": "port",
            "sample": "full workload (%d nnz x N=%d), median of %d runs of the oracle's OpenMP csr_mm "
                      "(includes output allocation); %s" % (a.nnz, bmat.shape[1], nrep, note),
            "ms": t * 1e3}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="rmat", choices=["rmat", "uniform"])
    ap.add_argument("--scale", type=int, default=20)
    ap.add_argument("--ncols", type=int, default=128)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-secondary", action="store_true", help="skip the uniform-random secondary measurement")
    ap.add_argument("--chunk", type=int, default=0, help="override the SpMM work-item chunk (tuning)")
    ap.add_argument("--unroll", type=int, default=0, help="override the SpMM load unroll 4|8 (tuning)")
    ap.add_argument("--hot-kb", type=int, default=-1, help="override the hot-set budget in KiB, 0 = no tagging (tuning)")
    args = ap.parse_args()

    import numpy as np
    import torch

    import sparse_dot_amd as sda
    from sparse_dot_amd._mi_interface import MI, SparseHandle, matrix_descr, sparse_matrix_t, _check_return_value
    import ctypes as ct

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch multi-GPU runs with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    ndev = torch.cuda.device_count()
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")  # "gloo": dry run of the N > 1 control flow on one GPU
    dev_index = local_rank if backend == "nccl" else local_rank % ndev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI
        else:
            dist.init_process_group(backend)

    def bcast(t, src=0):
        if backend == "nccl":
            dist.broadcast(t, src=src)
        else:  # gloo dry run: stage through the host
            h = t.cpu()
            dist.broadcast(h, src=src)
            t.copy_(h)

    def allreduce(t, op):
        if backend == "nccl":
            dist.all_reduce(t, op=op)
            return t
        h = t.cpu()
        dist.all_reduce(h, op=op)
        return h.to(t.device)

    def allgather_rows(dst, src_block):
        if backend == "nccl":
            dist.all_gather_into_tensor(dst, src_block)
        else:
            parts = [torch.empty_like(src_block, device="cpu") for _ in range(world)]
            dist.all_gather(parts, src_block.cpu())
            dst.copy_(torch.cat(parts, 0))

    sda.mi_set_device(dev_index)
    stream = torch.cuda.current_stream()
    sda.mi_set_stream(stream.cuda_stream)
    if args.chunk:
        sda.mi_set_option("spmm_chunk", args.chunk)
    if args.unroll:
        sda.mi_set_option("spmm_unroll", args.unroll)
    if args.hot_kb >= 0:
        sda.mi_set_option("spmm_hot_kb", args.hot_kb)
    for kv in os.environ.get("MI_BENCH_OPTS", "").split(","):  # tuning hook, e.g. MI_BENCH_OPTS=pool_enable=0
        if "=" in kv:
            sda.mi_set_option(kv.split("=")[0], int(kv.split("=")[1]))

    # ---- synthetic inputs, generated on the device ------------------------------------------------
    N = args.ncols
    if args.workload == "rmat":
        indptr, indices, vals, n = rmat_csr(torch, args.scale, 32, 7 + rank, dev)
    else:
        indptr, indices, vals, n = uniform_csr(torch, 1 << args.scale, 32, 7 + rank, dev)
    nnz = int(indices.numel())
    gb = torch.Generator(device=dev)
    gb.manual_seed(9)
    B = torch.rand((n, N), generator=gb, device=dev, dtype=torch.float32)
    if world > 1:
        bcast(B)  # replicate B (set-up; cost reported under "collectives")
    C = torch.empty((n, N), device=dev, dtype=torch.float32)
    torch.cuda.synchronize()

    # ---- handle over DEVICE pointers (zero copy) + executor call through the C ABI ---------------
    ref = sparse_matrix_t()
    ret = MI.call("mi_sparse_s_create_csr", ct.byref(ref), 0, n, n, indptr.data_ptr(), indptr.data_ptr() + 4,
                  indices.data_ptr(), vals.data_ptr())
    _check_return_value(ret, "mi_sparse_s_create_csr")
    handle = SparseHandle(ref, "s", keepalive=(indptr, indices, vals))

    def step():
        r = MI.call("mi_sparse_s_mm", 10, 1.0, handle.ptr, matrix_descr(), 101, B.data_ptr(), N, N, 0.0,
                    C.data_ptr(), N)
        if r:
            _check_return_value(r, "mi_sparse_s_mm")

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist:
        t = allreduce(torch.tensor([elapsed], device=dev, dtype=torch.float64), dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tn = allreduce(torch.tensor([float(nnz)], device=dev, dtype=torch.float64), dist.ReduceOp.SUM)
        total_nnz = float(tn.item())
    else:
        total_nnz = float(nnz)
    ms_per_step = elapsed / args.steps * 1e3
    gflops = 2.0 * total_nnz * N / (elapsed / args.steps) / 1e9

    # ---- dominant-kernel duration with hipEvents on the launch stream (separate loop) ------------
    sda.mi_set_option("profile_events", 1)
    sda.mi_get_counter("reset")
    for _ in range(max(5, min(args.steps, 20))):
        step()
    torch.cuda.synchronize()
    k_ms = sda.mi_get_counter("spmm_kernel_ms") / max(1.0, sda.mi_get_counter("spmm_kernel_launches"))
    sda.mi_set_option("profile_events", 0)
    alg_bytes = nnz * 8 + (n + 1) * 8 + n * N * 4 + n * N * 4
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "spmm_traffic.json")
    if os.path.exists(tpath) and args.workload == "rmat" and N == 128 and args.scale == 20:
        try:
            traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    tagged = bool(sda.mi_get_counter("spmm_last_tagged"))
    hot_coverage = round(sda.mi_get_counter("spmm_hot_coverage"), 4)  # of the timed matrix (the counters are "last call")
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "kernel": "k_spmm<float,4,%d,%d,%s>" % (32 if N == 128 else 64 if N >= 256 else 16, args.unroll or 4,
                                                        "true" if tagged else "false"),
                "kernel_ms": round(k_ms, 4), "algorithmic_bytes": alg_bytes}

    # ---- collectives the row-partitioned path needs around the kernel (reported, not timed above) --
    collectives = None
    if dist:
        def timed(fn, reps=3):
            fn()
            torch.cuda.synchronize()
            dist.barrier()
            t1 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t1) / reps * 1e3
        try:  # informational: a failure here must not cost the headline line
            gathered = torch.empty((world * n, N), device=dev, dtype=torch.float32)
            collectives = {
                "broadcast_B_ms": round(timed(lambda: bcast(B)), 3),
                "allgather_C_ms": round(timed(lambda: allgather_rows(gathered, C)), 3),
                "backend": backend,
                "note": "RCCL over xGMI; not part of the timed step (weak scaling, outputs stay row-distributed)",
            }
            del gathered
        except Exception as exc:  # noqa: BLE001
            collectives = {"error": "%s: %s" % (type(exc).__name__, exc), "backend": backend}

    # ---- secondary workload the north_star asks to report alongside: uniform-random CSR, same shape ----
    secondary = None
    if not args.no_secondary and args.workload == "rmat" and world == 1 and not args.no_cpu:
        u_ptr, u_idx, u_val, _ = uniform_csr(torch, n, 32, 11, dev)
        uref = sparse_matrix_t()
        _check_return_value(MI.call("mi_sparse_s_create_csr", ct.byref(uref), 0, n, n, u_ptr.data_ptr(),
                                    u_ptr.data_ptr() + 4, u_idx.data_ptr(), u_val.data_ptr()), "mi_sparse_s_create_csr")
        C2 = torch.empty_like(C)

        def ustep():
            r = MI.call("mi_sparse_s_mm", 10, 1.0, uref, matrix_descr(), 101, B.data_ptr(), N, N, 0.0, C2.data_ptr(), N)
            if r:
                _check_return_value(r, "mi_sparse_s_mm")
        for _ in range(args.warmup):
            ustep()
        torch.cuda.synchronize()
        tu = time.perf_counter()
        for _ in range(args.steps):
            ustep()
        torch.cuda.synchronize()
        tu = (time.perf_counter() - tu) / args.steps
        u_nnz = int(u_idx.numel())
        u_bytes = u_nnz * 8 + (n + 1) * 8 + 2 * n * N * 4
        secondary = {"workload": "uniform-random CSR %dx%d, 32/row (%d nnz) x dense %dx%d fp32" % (n, n, u_nnz, n, N),
                     "value": round(2.0 * u_nnz * N / tu / 1e9, 2), "unit": "GFLOP/s", "ms_per_step": round(tu * 1e3, 4),
                     "algorithmic_GBps": round(u_bytes / tu / 1e9, 1),
                     "hot_cold_tagged_gather": bool(sda.mi_get_counter("spmm_last_tagged"))}
        MI.call("mi_sparse_destroy", uref)
        del C2, u_ptr, u_idx, u_val

    # ---- parity spot check of the timed configuration (a row sample vs fp64 on the GPU) -----------
    cpu = None
    if rank == 0:
        sel = torch.randint(0, n, (64,), device=dev)
        ip = indptr.to(torch.int64)
        worst = 0.0
        for r in sel.tolist():
            lo, hi = int(ip[r]), int(ip[r + 1])
            want = (vals[lo:hi].double()[:, None] * B[indices[lo:hi].long()].double()).sum(0)
            got = C[r].double()
            den = torch.clamp(want.abs(), min=1e-30)
            worst = max(worst, float(((got - want).abs() / den).max())) if hi > lo else max(worst, float(got.abs().max()))
        assert worst < 1e-5, "bench result fails the fp32 parity bar: %g" % worst
        if not args.no_cpu and world == 1:
            cpu = cpu_baseline(indptr.cpu().numpy(), indices.cpu().numpy(), vals.cpu().numpy(), n, B.cpu().numpy())

    if rank == 0:
        line = {
            "metric": "spmm_effective_gflops", "value": round(gflops, 2), "unit": "GFLOP/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s CSR %dx%d (%d nnz per rank, int32 indices) x dense %dx%d fp32, C := A @ B"
                                   % ("R-MAT(.57,.19,.19,.05) scale %d, 32 edges/row, dedup" % args.scale
                                      if args.workload == "rmat" else "uniform 32/row", n, n, nnz, n, N),
                       "partition": "1-D row blocks, one block per GPU" if world > 1 else "single GPU",
                       "spmm_chunk": args.chunk or 256,
                       "hot_cold_tagged_gather": tagged, "hot_column_coverage": hot_coverage},
            "roofline": roofline, "cpu_baseline": cpu, "parity_max_rel_err_sample": worst,
        }
        if collectives:
            line["collectives"] = collectives
        if secondary:
            line["secondary"] = secondary
        print(json.dumps(line), flush=True)
    handle.destroy()
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
