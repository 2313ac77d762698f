#!/usr/bin/env python
"""
bench.py -- headline benchmark of the hot path: fp32 CSR x dense SpMM on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload rmat|uniform] [--no-cpu] [--no-secondary]

Workload (BASELINE.json configs[1]): R-MAT CSR 2^20 x 2^20 (scale 20, 32 edges/row drawn,
a,b,c,d = .57,.19,.19,.05, duplicates merged -> ~31.4 M nnz, values U[0.5,1.5)) times a dense
2^20 x 128 fp32 matrix, synthetic, generated on the GPU.  A "step" is one mi_sparse_s_mm call
(C := A @ B) through the C ABI with A, B, C resident in HBM (device pointers, zero copy).

One JSON line is printed by rank 0:
  value        = effective GFLOP/s = 2 * nnz * N / step time   (whole job)
  roofline     = algorithmic bytes of one launch (nnz*(4+4) + (M+1)*8 + K*N*4 + M*N*4;
                 the row pointer is 8 bytes per row in this build) / mean duration of the dominant
                 kernel (k_spmm) measured with hipEvents on the launch stream, vs the 8 TB/s HBM peak
                 (and, as `frac_of_measured_copy`, vs a device copy measured in the same run)
  plan         = what the inspector costs: first call on a fresh handle vs the steady state, and when the
                 hot / cold tags are adopted (the analysis runs on the device behind the second product)
  host_api     = one dot_product_mkl(a_scipy, b_numpy) call -- the reference's calling convention: host
                 arrays in, host array out, handle created and destroyed inside the call
  cpu_baseline = the same SpMM on the host: MKL's mkl_sparse_s_mm through oracle/mkl_shim.py when a
                 libmkl_rt is discoverable (kind "reference": MKL is the reference's arithmetic
                 engine), else the oracle's OpenMP port (kind "port"); `variants` adds MKL at one thread
                 and scipy's `@`; medians of 5.
  secondary    = the other BASELINE configs on one GPU, each with its own roofline (+ cpu_baseline where a CPU
                 can hold it): uniform SpMM, SpGEMM cfg3 (uniform and the literal R-MAT), gram cfg4 (literal).

Multi-GPU (--gpus N, launched by torch.distributed.run, one rank per GPU): the SAME matrix is split into N
contiguous nnz-balanced row blocks (sparse_dot_amd.distributed.partition_rows); the timed step is the
north_star's pipeline end to end -- RCCL broadcast of B from rank 0, the local kernel on the resident block
writing straight into this rank's slice of C, all-gatherv of the output row blocks -- everything device
resident.  `value` is computed from that step (total work is fixed: "scaling": "strong"); `compute_only_ms`
(the kernels alone, max over ranks) and the single collectives are reported beside it.  At N = 1 the step is
the kernel alone, i.e. the single-GPU line.  The same step is also timed in its xGMI-shaped forms (`variants`: grouped
point-to-point all-gatherv + scatter / all-gather broadcast, and the column-panel pipeline that overlaps them with the
kernel); `value` is the best form's, every form is listed.  With 8 ranks BASELINE configs[4] (2^24 x 2^24 x 256) runs as
`secondary.spmm_config5_8gpu`.  A watchdog prints the contract line and exits if one of these never-before-run RCCL
paths does not come back.
"""
import argparse
import ctypes as ct
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the guide's measured copy ceiling


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


def uniform_csr(torch, n, per_row, seed, device, ncols=None, dtype=None):
    """`per_row` random columns per row (duplicates merged), sorted; n x ncols (square by default)."""
    ncols = n if ncols is None else ncols
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    rows = torch.arange(n, device=device, dtype=torch.int64).repeat_interleave(per_row)
    cols = torch.randint(0, ncols, (n * per_row,), generator=g, device=device, dtype=torch.int64)
    key = torch.unique(rows * ncols + cols)
    del rows, cols
    r = key // ncols
    indices = (key % ncols).to(torch.int32)
    del key
    counts = torch.bincount(r, minlength=n)
    indptr = torch.zeros(n + 1, dtype=torch.int64, device=device)
    indptr[1:] = torch.cumsum(counts, 0)
    vals = torch.rand(indices.numel(), generator=g, device=device, dtype=torch.float32) + 0.5
    if dtype is not None:
        vals = vals.to(dtype)
    return indptr.to(torch.int32), indices, vals, n


def _median(ts):
    return sorted(ts)[len(ts) // 2]


def _timed(fn, reps, warm=1):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return ts


def _host_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count()


# ------------------------------------------------------------------------------------------------
# CPU baselines (rank 0, N = 1 only): the reference's engine (MKL through the build's own shim) when the box
# has a libmkl_rt, else the oracle's OpenMP port; bounded samples, medians of 5
# ------------------------------------------------------------------------------------------------
def cpu_baseline_spmm(indptr, indices, vals, n, bmat, nrep=5):
    import numpy as np
    import scipy.sparse as sps

    a = sps.csr_matrix((vals, indices, indptr), shape=(n, n))
    flops = 2.0 * a.nnz * bmat.shape[1]
    out = np.zeros((n, bmat.shape[1]), dtype=np.float32)  # preallocated: no first-touch faults in the timing
    variants = {}
    base = None
    note = ""
    try:
        from oracle import mkl_shim
        mkl = mkl_shim.MklSpmm()
        h = mkl.make(a)
        nthr = mkl.threads()
        t = _median(_timed(lambda: mkl.mm(h, bmat, out), nrep))
        base = {"value": round(flops / t / 1e9, 2), "unit": "GFLOP/s", "cores": nthr, "kind": "reference",
                "sample": "full workload (%d nnz x N=%d), median of %d mkl_sparse_s_mm calls via oracle/mkl_shim.py after 1 "
                          "warm-up, preallocated output; %s; host has %d logical cpus"
                          % (a.nnz, bmat.shape[1], nrep, mkl.version(), os.cpu_count()),
                "ms": round(t * 1e3, 2)}
        one = mkl.set_threads(1)
        t1 = _median(_timed(lambda: mkl.mm(h, bmat, out), nrep))
        variants["mkl_1_thread"] = {"value": round(flops / t1 / 1e9, 2), "unit": "GFLOP/s", "cores": one,
                                    "ms": round(t1 * 1e3, 2), "sample": "same call, MKL_Set_Num_Threads(1), median of %d" % nrep}
        mkl.set_threads(nthr)
        mkl.destroy(h)
    except Exception as e:  # no MKL on this box: fall back to the oracle's OpenMP port
        note = "libmkl_rt unavailable (%s)" % (str(e)[:80],)
    if base is None:
        from oracle import cpu_oracle
        cpu_oracle.spmm(a[:1024], bmat)  # build + warm
        t = _median(_timed(lambda: cpu_oracle.spmm(a, bmat), 3, warm=0))
        base = {"value": round(flops / t / 1e9, 2), "unit": "GFLOP/s", "cores": _host_cores(), "kind": "port",
                "sample": "full workload (%d nnz x N=%d), median of 3 runs of the oracle's OpenMP csr_mm "
                          "(includes output allocation); %s" % (a.nnz, bmat.shape[1], note),
                "ms": round(t * 1e3, 2)}
    if base.get("kind") == "reference":
        # the build's own OpenMP restatement (oracle/sparse_oracle.c, the parity checker) at all cores,
        # reported beside MKL -- a checker being timed, not a product path
        try:
            from oracle import cpu_oracle
            cpu_oracle.spmm(a[:1024], bmat)  # build + warm
            to = _median(_timed(lambda: cpu_oracle.spmm(a, bmat), 3, warm=0))
            variants["oracle_openmp_port"] = {"value": round(flops / to / 1e9, 2), "unit": "GFLOP/s", "cores": _host_cores(),
                                              "ms": round(to * 1e3, 2), "kind": "port",
                                              "sample": "the oracle's OpenMP csr_mm on the full workload (includes its output "
                                                        "allocation), median of 3"}
        except Exception as e:  # noqa: BLE001
            variants["oracle_openmp_port"] = {"error": str(e)[:120]}
    ts = _timed(lambda: a @ bmat, nrep)
    variants["scipy_1_thread"] = {"value": round(flops / _median(ts) / 1e9, 2), "unit": "GFLOP/s", "cores": 1,
                                  "ms": round(_median(ts) * 1e3, 2),
                                  "sample": "scipy.sparse `a @ b` (allocates its output), median of %d" % nrep}
    base["variants"] = variants
    return base


def cpu_baseline_spgemm(a, b, products, nrep=5):
    """MKL mkl_sparse_spmm on host copies of the operands (LP64: the result must stay below 2^31 entries)."""
    flops = 2.0 * products
    try:
        from oracle import mkl_shim
        mkl = mkl_shim.MklSpmm()
        ha, hb = mkl.make(a), mkl.make(b)
        t = _median(_timed(lambda: mkl.spmm(ha, hb), nrep))
        mkl.destroy(ha)
        mkl.destroy(hb)
        return {"value": round(flops / t / 1e9, 3), "unit": "GFLOP/s", "cores": mkl.threads(), "kind": "reference",
                "ms": round(t * 1e3, 1),
                "sample": "full workload, median of %d mkl_sparse_spmm calls (multiply only: no export, no ordering) via "
                          "oracle/mkl_shim.py; %s" % (nrep, mkl.version())}
    except Exception as e:
        ts = _timed(lambda: a @ b, 3, warm=0)
        return {"value": round(flops / _median(ts) / 1e9, 3), "unit": "GFLOP/s", "cores": 1, "kind": "port",
                "ms": round(_median(ts) * 1e3, 1),
                "sample": "full workload, scipy.sparse `a @ b` (the north_star's parity oracle), median of 3; libmkl_rt "
                          "unavailable (%s)" % (str(e)[:60],)}


def cpu_baseline_gram(a, nrep=3):
    """MKL mkl_sparse_s_syrkd on a bounded sample (the CPU scatter runs at ~0.2 GFLOP/s on 8 threads)."""
    import numpy as np
    lens = np.diff(a.indptr).astype(np.float64)
    flops = float((lens * (lens + 1)).sum())
    n = a.shape[1]
    try:
        from oracle import mkl_shim
        mkl = mkl_shim.MklSpmm()
        h = mkl.make(a)
        out = np.zeros((n, n), dtype=np.float32)
        t = _median(_timed(lambda: mkl.syrkd(h, out), nrep))
        mkl.destroy(h)
        return {"value": round(flops / t / 1e9, 3), "unit": "GFLOP/s", "cores": mkl.threads(), "kind": "reference",
                "ms": round(t * 1e3, 1),
                "sample": "uniform %d x %d, %d nnz fp32 (a row sample of the workload at 1/16 of its width), median of %d "
                          "mkl_sparse_s_syrkd calls via oracle/mkl_shim.py; %s" % (a.shape[0], n, a.nnz, nrep, mkl.version())}
    except Exception as e:
        from oracle import cpu_oracle
        ts = _timed(lambda: cpu_oracle.syrkd(a), 1, warm=0)
        return {"value": round(flops / _median(ts) / 1e9, 3), "unit": "GFLOP/s", "cores": _host_cores(), "kind": "port",
                "ms": round(_median(ts) * 1e3, 1),
                "sample": "uniform %d x %d, %d nnz fp32, one run of the oracle's syrkd; libmkl_rt unavailable (%s)"
                          % (a.shape[0], n, a.nnz, str(e)[:60])}


def cpu_baseline_mkl_child(op, operands, flops, sample, nrep=5, unit="GFLOP/s"):
    """MKL through the build's own shim for one more entry point, timed in a CHILD process (tools/mkl_child.py): oneMKL 2021.4's
    mkl_sparse_syrk aborted with heap corruption inside bench.py's own process (next to torch's OpenMP runtime) and ran cleanly on
    the same operand in a process of its own; whatever happens to the child, the line survives.  operands: scipy CSR matrices."""
    import subprocess
    import tempfile
    import numpy as np
    try:
        with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
            paths = []
            for k, m in enumerate(operands):
                path = os.path.join(tmp, "m%d.npz" % k)
                np.savez(path, data=m.data, indices=m.indices, indptr=m.indptr, shape=np.array(m.shape))
                paths.append(path)
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "mkl_child.py"), op] + paths + [str(nrep)],
                               stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=900)
        lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
        if not lines:
            return {"value": None, "note": "%s: the MKL child died (rc %d) or libmkl_rt is unavailable" % (sample, r.returncode)}
        d = json.loads(lines[-1])
        time.sleep(0.4)
        return {"value": round(flops / (d["ms"] / 1e3) / 1e9, 3), "unit": unit, "cores": d["cores"], "kind": "reference", "ms": round(d["ms"], 3),
                "sample": "%s, median of %d calls after 1 warm-up in a child process via oracle/mkl_shim.py; %s; host has %d logical cpus"
                          % (sample, nrep, d["version"], os.cpu_count())}
    except Exception as e:  # noqa: BLE001
        return {"value": None, "note": "%s: %s" % (sample, str(e)[:120])}


# ------------------------------------------------------------------------------------------------
# C-ABI helpers over torch device tensors
# ------------------------------------------------------------------------------------------------
class Abi:
    def __init__(self):
        import sparse_dot_amd as sda
        from sparse_dot_amd._mi_interface import MI, matrix_descr, sparse_matrix_t, _check_return_value
        self.sda, self.MI, self.descr, self.handle_t, self.check = sda, MI, matrix_descr, sparse_matrix_t, _check_return_value

    def create(self, letter, indptr, indices, vals, rows, cols):
        h = self.handle_t()
        self.check(self.MI.call("mi_sparse_%s_create_csr" % letter, ct.byref(h), 0, rows, cols, indptr.data_ptr(),
                                indptr.data_ptr() + indptr.element_size(), indices.data_ptr(), vals.data_ptr()), "create_csr")
        return h

    def destroy(self, h):
        self.MI.call("mi_sparse_destroy", h)

    def mm(self, letter, h, B, C, N, beta=0.0):
        r = self.MI.call("mi_sparse_%s_mm" % letter, 10, 1.0, h, self.descr(), 101, B.data_ptr(), N, N, beta, C.data_ptr(), N)
        if r:
            self.check(r, "mi_sparse_%s_mm" % letter)

    def mv(self, letter, h, x, y):
        self.check(self.MI.call("mi_sparse_%s_mv" % letter, 10, 1.0, h, self.descr(), x.data_ptr(), 0.0, y.data_ptr()), "mv")

    def info(self, h):
        rows, cols, nnz = ct.c_int64(), ct.c_int64(), ct.c_int64()
        self.check(self.MI.call("mi_sparse_get_info", h, ct.byref(rows), ct.byref(cols), ct.byref(nnz), None, None), "info")
        return rows.value, cols.value, nnz.value


def measured_copy_gbs(torch, dev):
    """Device copy bandwidth (read + write) of a 1 GiB fp32 tensor, best of 5 -- SURVEY section 8d's 'measured' peak."""
    src = torch.empty(1 << 28, dtype=torch.float32, device=dev)
    dst = torch.empty_like(src)
    src.fill_(1.0)
    best = 0.0
    for _ in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        dst.copy_(src)
        e1.record()
        torch.cuda.synchronize()
        best = max(best, 2.0 * src.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    del src, dst
    return best


def measure_traffic_live(timeout_s=150):
    """HBM bytes per PRODUCT of the headline SpMM from rocprofv3 PMC counters, collected NOW: two separate
    `rocprofv3 --kernel-trace --pmc <counter>` passes (FETCH_SIZE, WRITE_SIZE -- MI355X_MICROARCH.md, HBM section) over
    tools/gpu_kpart.py, which runs this very configuration (same generator, seeds, library defaults, plans adopted by
    untimed calls first) in a child process.  One product = every SpMM kernel the library launches for it (round 5: short
    rows row-owned + long rows by column partition + their carry fix-ups + the combine); the bytes of all of them are
    summed.  gfx950 corrections as the guide prescribes: FETCH_SIZE (KB) x 2 for wide coalesced reads, WRITE_SIZE (KB)
    as is.  Returns (bytes, description) or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    from collections import defaultdict
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not on PATH"
    launches = 3
    out = {}
    names = set()
    tmp = tempfile.mkdtemp(prefix="mi_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = [exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "p", "--",
                   sys.executable, os.path.join(ROOT, "tools", "gpu_kpart.py"), "--launches", str(launches), "--variants", "default"]
            env = dict(os.environ, TMPDIR="/tmp")
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout_s)
            if r.returncode != 0:
                return None, "rocprofv3 --pmc %s failed (rc %d)" % (counter, r.returncode)
            acc = defaultdict(float)
            kname = {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row["Counter_Name"] == counter:
                        acc[int(row["Dispatch_Id"])] += float(row["Counter_Value"])
                        kname[int(row["Dispatch_Id"])] = row["Kernel_Name"]
            marks = [k for k in kname if "FillFunctor" in kname[k]]
            if not marks:
                return None, "no marker dispatch in the %s pass" % counter
            timed = [k for k in sorted(kname) if k > max(marks) and
                     any(t in kname[k] for t in ("mi::k_spmm", "mi::k_kp_combine"))]
            if len(timed) < launches:
                return None, "no SpMM dispatches behind the marker in the %s pass" % counter
            names.update(kname[k].split("(")[0].replace("void ", "") for k in timed)
            out[counter] = sum(acc[k] for k in timed) / launches
        nbytes = out["FETCH_SIZE"] * 1024.0 * 2.0 + out["WRITE_SIZE"] * 1024.0
        return nbytes, ("rocprofv3 --pmc, two passes in this run: FETCH_SIZE %.0f KB x 2 (gfx950 correction) + WRITE_SIZE %.0f KB per "
                        "product, mean of %d products, summed over its kernels: %s" % (out["FETCH_SIZE"], out["WRITE_SIZE"], launches,
                                                                                        "; ".join(sorted(names))))
    except Exception as exc:  # noqa: BLE001
        return None, "%s: %s" % (type(exc).__name__, str(exc)[:200])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)



def measure_traffic_child(child_args, calls, include, exclude=("k_spmv", "k_spmm"), timeout_s=240, need_free_bytes=0):
    """HBM-side bytes per CALL of a secondary workload from rocprofv3 PMC counters collected NOW: two separate
    `rocprofv3 --kernel-trace --pmc <counter>` passes (FETCH_SIZE, WRITE_SIZE) over tools/bench_ops.py, which runs the very
    workload of the secondary (same generator, seeds, library defaults) `calls` times in a child process.  Bytes = sum over
    every library kernel whose name contains one of `include` (and none of `exclude`: the row-sum checks of the child)
    of FETCH_SIZE KB x 2 (gfx950 correction for wide reads, MI355X_MICROARCH.md; RDREQ x 128 B for gathers by the
    calibration of profiles/r03_fetch_size_calibration.log -- the same figure) + WRITE_SIZE KB, divided by `calls`.
    Returns ({"traffic": bytes, "by_kernel_GB": {...}}, description) or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    from collections import defaultdict
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not on PATH"
    tmp = tempfile.mkdtemp(prefix="mi_pmc_", dir="/tmp")
    per_kernel = defaultdict(lambda: defaultdict(float))

    def wait_for_memory():  # what this process (and the previous child) released comes back to the driver asynchronously
        if not need_free_bytes:
            return
        import torch
        import sparse_dot_amd as sda
        sda.mi_set_option("pool_trim", 1)  # this process's cached device blocks and scratch arena back to the driver: the child needs them
        torch.cuda.empty_cache()
        for _ in range(80):
            if torch.cuda.mem_get_info()[0] >= need_free_bytes:
                return
            time.sleep(0.5)
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            wait_for_memory()
            cmd = [exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "p", "--",
                   sys.executable, os.path.join(ROOT, "tools", "bench_ops.py")] + list(child_args)
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE,
                               stderr=subprocess.STDOUT, timeout=timeout_s)
            if r.returncode != 0:
                return None, "rocprofv3 --pmc %s failed (rc %d): %s" % (counter, r.returncode, r.stdout.decode(errors="replace")[-300:].replace("\n", " | "))
            seen = 0
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    name = row["Kernel_Name"]
                    if row["Counter_Name"] != counter or not any(k in name for k in include) or any(k in name for k in exclude):
                        continue
                    base = name.split("<")[0].split("(")[0].replace("void ", "").replace("mi::", "")
                    per_kernel[base][counter] += float(row["Counter_Value"])
                    seen += 1
            if not seen:
                return None, "no matching dispatches in the %s pass" % counter
        by = {k: (v.get("FETCH_SIZE", 0.0) * 2048.0 + v.get("WRITE_SIZE", 0.0) * 1024.0) / calls for k, v in per_kernel.items()}
        total = sum(by.values())
        top = dict(sorted(((k, round(b / 1e9, 3)) for k, b in by.items()), key=lambda kv: -kv[1])[:6])
        return ({"traffic": total, "by_kernel_GB": top},
                "rocprofv3 --pmc, two passes in this run over `tools/bench_ops.py %s`: sum over the library's kernels of FETCH_SIZE KB x 2 "
                "+ WRITE_SIZE KB, per call (%d calls)" % (" ".join(child_args), calls))
    except Exception as exc:  # noqa: BLE001
        return None, "%s: %s" % (type(exc).__name__, str(exc)[:200])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def attach_traffic(entry, child_args, calls, include, enabled, exclude=("k_spmv", "k_spmm"), need_free_bytes=0):
    """roofline.traffic (+ its source and the per-kernel split) for a secondary entry."""
    if not enabled or "roofline" not in entry:
        return
    res, src = measure_traffic_child(child_args, calls, include, exclude=exclude, need_free_bytes=need_free_bytes)
    rf = entry["roofline"]
    if res is None:
        rf["traffic"] = None
        rf["traffic_source"] = "live collection unavailable: %s" % src
        return
    rf["traffic"] = res["traffic"]
    rf["traffic_over_algorithmic"] = round(res["traffic"] / rf["algorithmic_bytes"], 2)
    rf["traffic_by_kernel_GB"] = res["by_kernel_GB"]
    rf["traffic_source"] = src


def load_traffic(name):
    """HBM bytes per launch from this round's committed PMC summary (separate rocprofv3 --pmc passes; the counters
    cannot be collected from inside this process)."""
    path = os.path.join(ROOT, "profiles", name)
    try:
        d = json.load(open(path))
        return d.get("hbm_bytes_per_launch"), "profiles/" + name
    except Exception:
        return None, None


# ------------------------------------------------------------------------------------------------
# secondary workloads (single GPU)
# ------------------------------------------------------------------------------------------------
def secondary_uniform_spmm(torch, abi, dev, n, N, B, steps, warmup):
    u_ptr, u_idx, u_val, _ = uniform_csr(torch, n, 32, 11, dev)
    h = abi.create("s", u_ptr, u_idx, u_val, n, n)
    C2 = torch.empty((n, N), device=dev, dtype=torch.float32)
    for _ in range(max(warmup, 3)):
        abi.mm("s", h, B, C2, N)
    torch.cuda.synchronize()
    tu = time.perf_counter()
    for _ in range(steps):
        abi.mm("s", h, B, C2, N)
    torch.cuda.synchronize()
    tu = (time.perf_counter() - tu) / steps
    u_nnz = int(u_idx.numel())
    u_bytes = u_nnz * 8 + (n + 1) * 8 + 2 * n * N * 4
    out = {"workload": "uniform-random CSR %dx%d, 32/row (%d nnz) x dense %dx%d fp32" % (n, n, u_nnz, n, N),
           "value": round(2.0 * u_nnz * N / tu / 1e9, 2), "unit": "GFLOP/s", "ms_per_step": round(tu * 1e3, 4),
           "roofline": {"bound": "hbm", "achieved": round(u_bytes / tu / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(u_bytes / tu / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes": u_bytes,
                        "note": "step time (kernel + fix-up); no hot set exists, every B row is an L2 miss"},
           "hot_cold_tagged_gather": bool(abi.sda.mi_get_counter("spmm_last_tagged")),
           "slices": int(abi.sda.mi_get_counter("spmm_last_slices"))}
    abi.destroy(h)
    return out


def secondary_spgemm(torch, abi, dev, kind, with_cpu):
    """BASELINE configs[2]: two CSR 2^20 x 2^20, 16/row fp64 (uniform, or the literal R-MAT) -> sparse C."""
    n = 1 << 20
    if kind == "rmat":
        a = rmat_csr(torch, 20, 16, 21, dev)
        b = rmat_csr(torch, 20, 16, 23, dev)
    else:
        a = uniform_csr(torch, n, 16, 1, dev)
        b = uniform_csr(torch, n, 16, 2, dev)
    av, bv = a[2].double(), b[2].double()
    ha = abi.create("d", a[0], a[1], av, n, n)
    hb = abi.create("d", b[0], b[1], bv, n, n)
    times = []
    hc = None
    reps = 2 if kind == "rmat" else 5
    for rep in range(reps + 1):
        if hc is not None:
            abi.destroy(hc)
        hc = abi.handle_t()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        abi.check(abi.MI.call("mi_sparse_spmm", 10, ha, hb, ct.byref(hc)), "spmm")
        torch.cuda.synchronize()
        if rep:
            times.append(time.perf_counter() - t0)
        else:
            first = time.perf_counter() - t0
    t = _median(times)
    _, _, nnzc = abi.info(hc)
    colnnz_a = torch.bincount(a[1].long(), minlength=n).double()
    rownnz_b = (b[0][1:] - b[0][:-1]).double()
    products = float((colnnz_a * rownnz_b).sum())
    # parity at scale: C 1 == A (B 1)
    ones = torch.ones(n, device=dev, dtype=torch.float64)
    b1, ab1, c1 = (torch.empty(n, device=dev, dtype=torch.float64) for _ in range(3))
    abi.mv("d", hb, ones, b1)
    abi.mv("d", ha, b1, ab1)
    abi.mv("d", hc, ones, c1)
    torch.cuda.synchronize()
    rel = float(((c1 - ab1).abs() / ab1.abs().clamp(min=1e-300)).max())
    nbytes = (a[1].numel() + b[1].numel() + nnzc) * 12 + 3 * (n + 1) * 8
    out = {"workload": "SpGEMM (BASELINE configs[2]%s): %s CSR 2^20 x 2^20, 16/row fp64, squared; sparse C with %d entries"
                       % (" as literally stated" if kind == "rmat" else ", uniform variant", kind, nnzc),
           "nnzA": int(a[1].numel()), "nnzB": int(b[1].numel()), "nnzC": int(nnzc), "products": products,
           "ms": round(t * 1e3, 3), "first_call_ms": round(first * 1e3, 1), "value": round(2 * products / t / 1e9, 2),
           "unit": "GFLOP/s", "dtype": "f64",
           "roofline": {"bound": "hbm", "achieved": round(nbytes / t / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(nbytes / t / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes": nbytes,
                        "note": "(nnzA + nnzB + nnzC) * 12 + 3 (M + 1) * 8 over the whole mi_sparse_spmm call (median of %d, "
                                "all phases: upper bounds, binning, symbolic, scan, numeric)" % reps},
           "parity_rowsum_max_rel_err": rel}
    assert rel <= 1e-12, "SpGEMM row-sum parity check failed: %g" % rel
    # the same product with ordered rows (the reference's reorder_output=True: mkl_sparse_spmm + mkl_sparse_order,
    # _sparse_sparse.py:226-230): one mi_sparse_spmm_ordered call, median of 2 after 1
    try:
        abi.destroy(hc)
        hc = None
        ot = []
        for rep in range(3):
            ho = abi.handle_t()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            abi.check(abi.MI.call("mi_sparse_spmm_ordered", 10, ha, hb, ct.byref(ho)), "spmm_ordered")
            torch.cuda.synchronize()
            if rep:
                ot.append(time.perf_counter() - t0)
            if rep < 2:
                abi.destroy(ho)
        hc = ho
        out["ordered_ms"] = round(_median(ot) * 1e3, 3)
        out["ordered_note"] = ("mi_sparse_spmm_ordered: the product with every row's columns in increasing order, whole call "
                               "(product + ordering; long rows by rank when their bitmaps fit in 8 GiB, else sorted run by run in place)")
    except Exception as exc:  # noqa: BLE001
        out["ordered_ms"] = None
        out["ordered_note"] = "%s: %s" % (type(exc).__name__, str(exc)[:200])
    out["first_call_note"] = ("first_call_ms is the first product in THIS process; first_call_fresh_process the same product as the first "
                              "thing a new process does.  Both contain one hipMalloc of the 117 GB result, and device memory that has been "
                              "used before -- by this process or by one that has exited -- is scrubbed by the driver when it is handed out "
                              "again (tools/probes/alloc_probe.hip: 78 GiB in 0.25 ms from untouched memory, 5.8 s after a hipFree): "
                              "observed 0.18 s on untouched memory, 0.9 - 2.6 s (child) / 5.5 s (parent) on recycled memory, with the "
                              "same library and a steady state of 0.18 s")
    if kind == "rmat":
        for h in (ha, hb, hc):
            abi.destroy(h)
        ha = hb = hc = None
        del a, b, av, bv, b1, ab1, c1, ones, colnnz_a, rownnz_b
        abi.sda.mi_set_option("pool_trim", 1)
        torch.cuda.empty_cache()
        try:  # the same product as the first thing a fresh process does (tools/gpu_first_call.py)
            import subprocess
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_first_call.py")], stdout=subprocess.PIPE,
                               stderr=subprocess.DEVNULL, timeout=300)
            lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
            out["first_call_fresh_process"] = json.loads(lines[-1]) if lines else {"error": "no output (rc %d)" % r.returncode}
        except Exception as exc:  # noqa: BLE001
            out["first_call_fresh_process"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:200])}
        if with_cpu:
            out["cpu_baseline"] = {"value": None, "note": "nnz(C) = %d exceeds MKL's LP64 index range and the reference's "
                                                          "INT_MAX guard (_common.py:166-172): not runnable on the CPU path" % nnzc}
        return out
    if with_cpu and nnzc < 2**31 - 1:
        import scipy.sparse as sps
        ah = sps.csr_matrix((av.cpu().numpy(), a[1].cpu().numpy(), a[0].cpu().numpy()), shape=(n, n))
        bh = sps.csr_matrix((bv.cpu().numpy(), b[1].cpu().numpy(), b[0].cpu().numpy()), shape=(n, n))
        out["cpu_baseline"] = cpu_baseline_spgemm(ah, bh, products, nrep=3)
    elif with_cpu:
        out["cpu_baseline"] = {"value": None, "note": "nnz(C) = %d exceeds MKL's LP64 index range and the reference's "
                                                      "INT_MAX guard (_common.py:166-172): not runnable on the CPU path" % nnzc}
    for h in (ha, hb, hc):
        abi.destroy(h)
    return out


def secondary_spmv_gram_sparse(torch, abi, dev, indptr, indices, vals, n, steps, with_cpu=True):
    """SURVEY section 8 f1 (SpMV, mkl_sparse_?_mv, _sparse_vector.py:87-95) on the headline matrix and a5 (the SPARSE gram matrix,
    mkl_sparse_syrk, _gram_matrix.py:70-74) on a uniform 2^20 x 2^18, 16 / row fp64 operand -- the two rows of the scope table the
    line did not carry a measurement for."""
    out = {}
    h = abi.create("s", indptr, indices, vals, n, n)
    x = torch.rand(n, device=dev, dtype=torch.float32)
    y = torch.empty(n, device=dev, dtype=torch.float32)
    for _ in range(3):
        abi.mv("s", h, x, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        abi.mv("s", h, x, y)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / steps / 1e3
    ip = indptr.to(torch.int64)
    rows = torch.randint(0, n, (64,), device=dev)
    err = 0.0
    for r in rows.tolist():
        lo, hi = int(ip[r]), int(ip[r + 1])
        if hi > lo:
            w = float((vals[lo:hi].double() * x[indices[lo:hi].long()].double()).sum())
            err = max(err, abs(float(y[r]) - w) / max(abs(w), 1e-30))
    nnz = int(indices.numel())
    nbytes = nnz * 8 + (n + 1) * 8 + 2 * n * 4
    out["spmv"] = {"workload": "y = A x on the headline matrix (%d nnz), fp32" % nnz, "ms": round(t * 1e3, 4),
                   "value": round(2.0 * nnz / t / 1e9, 1), "unit": "GFLOP/s", "parity_max_rel_err_rows": err,
                   "roofline": {"bound": "hbm", "achieved": round(nbytes / t / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": round(nbytes / t / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes": nbytes,
                                "note": "nnz * 8 + (M + 1) * 8 + 2 M * 4; device time (events) of %d calls" % steps}}
    abi.destroy(h)
    if with_cpu:
        import numpy as np
        import scipy.sparse as sps
        ah = sps.csr_matrix((vals.cpu().numpy(), indices.cpu().numpy(), indptr.cpu().numpy()), shape=(n, n))
        out["spmv"]["cpu_baseline"] = cpu_baseline_mkl_child("mv", [ah], 2.0 * nnz, "full workload (mkl_sparse_s_mv, %d nnz)" % nnz)
        del ah
    m_, c_ = 1 << 20, 1 << 18
    u = uniform_csr(torch, m_, 16, 5, dev, ncols=c_)
    uv = u[2].double()
    hu = abi.create("d", u[0], u[1], uv, m_, c_)
    ts = []
    nnzc = 0
    for rep in range(4):
        hc = abi.handle_t()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        abi.check(abi.MI.call("mi_sparse_syrk", 11, hu, ct.byref(hc)), "syrk")
        torch.cuda.synchronize()
        if rep:
            ts.append(time.perf_counter() - t0)
        nnzc = abi.info(hc)[2]
        if rep == 3:
            # parity at scale: the row sums of triu(A^T A), from the result (C 1) and from the operand -- on the device, fp64:
            # (triu(A^T A) 1)_i = sum over rows r of A with a_ri != 0 of a_ri * (sum of a_rj over j >= i)
            onesc = torch.ones(c_, device=dev, dtype=torch.float64)
            c1 = torch.empty(c_, device=dev, dtype=torch.float64)
            abi.mv("d", hc, onesc, c1)
            torch.cuda.synchronize()
            ipu = u[0].to(torch.int64)
            lens_u = ipu[1:] - ipu[:-1]
            rows_of = torch.repeat_interleave(torch.arange(m_, device=dev), lens_u)
            pos = torch.arange(int(u[1].numel()), device=dev) - ipu[:-1][rows_of]  # position of the entry inside its (sorted) row
            wmax = int(lens_u.max())
            dense_rows = torch.zeros((m_, wmax), device=dev, dtype=torch.float64)
            dense_rows[rows_of, pos] = uv
            suffix = torch.flip(torch.cumsum(torch.flip(dense_rows, [1]), 1), [1])[rows_of, pos]  # sum of a_rj over j >= this column
            want = torch.zeros(c_, device=dev, dtype=torch.float64)
            want.index_add_(0, u[1].long(), uv * suffix)
            gram_parity = float(((c1 - want).abs() / want.abs().clamp(min=1e-300)).max())
            del onesc, c1, rows_of, pos, dense_rows, suffix, want
        abi.destroy(hc)
    tg = _median(ts)
    lens = (u[0][1:] - u[0][:-1]).double()
    prod = float((lens * (lens + 1) / 2).sum())
    gbytes = (int(u[1].numel()) * 2 + nnzc) * 12 + 3 * (c_ + 1) * 8
    out["gram_sparse"] = {"workload": "upper triangle of A^T A as CSR (mkl_sparse_syrk): uniform %d x %d, 16 / row fp64; %d entries" % (m_, c_, nnzc),
                          "ms": round(tg * 1e3, 3), "value": round(2 * prod / tg / 1e9, 2), "unit": "GFLOP/s",
                          "parity_rowsum_max_rel_err": gram_parity,
                          "roofline": {"bound": "hbm", "achieved": round(gbytes / tg / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": round(gbytes / tg / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes": gbytes,
                                       "note": "(nnz(A) + nnz(A^T) + nnz(C)) * 12 + 3 (n + 1) * 8 over the whole call (cached transpose + SpGEMM "
                                               "restricted to col >= row), median of 3 after 1"}}
    assert gram_parity <= 1e-12, "sparse gram row-sum parity check failed: %g" % gram_parity
    abi.destroy(hu)
    if with_cpu:
        # full workload first; a row sample (the first 2^18 rows: a quarter of the products) if MKL does not survive it
        import scipy.sparse as sps
        up, ui, ud = u[0].cpu().numpy(), u[1].cpu().numpy(), uv.cpu().numpy()
        base = None
        for rows_s, what in ((m_, "full workload"), (m_ // 4, "row sample: the first 2^18 of the 2^20 rows")):
            e = int(up[rows_s])
            us = sps.csr_matrix((ud[:e], ui[:e], up[:rows_s + 1]), shape=(rows_s, c_))
            lens_s = (u[0][1:rows_s + 1] - u[0][:rows_s]).double()
            prod_s = float((lens_s * (lens_s + 1) / 2).sum())
            base = cpu_baseline_mkl_child("syrk", [us], 2 * prod_s, "%s (mkl_sparse_syrk, multiply only: no export)" % what, nrep=3)
            if base.get("value") is not None:
                break
        out["gram_sparse"]["cpu_baseline"] = base
    return out


def secondary_rows_f3_f4_a4(torch, abi, dev, with_cpu=True):
    """The remaining rows of the scope table: f3 (BSR x dense through the block kernel, _common.py:327-384) and f4 (staged product,
    pattern reuse) measured by tools/bench_ops.py in a child process (its last JSON line, trimmed); a4 (mkl_sparse_?_spmmd,
    _sparse_sparse.py:94-101: sparse x sparse into a DENSE row-major result) here."""
    import subprocess
    out = {}
    for key, argv in (("bsr_spmm", ["bsr", "--rows-log2", "18", "--block", "4", "--ncols", "128", "--reps", "2"]),
                      ("sp2m_pattern_reuse", ["sp2m", "--kind", "rmat", "--scale", "18", "--reps", "2"])):
        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_ops.py")] + argv, stdout=subprocess.PIPE,
                               stderr=subprocess.DEVNULL, timeout=300)
            lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
            d = json.loads(lines[-1]) if lines else {"error": "no output (rc %d)" % r.returncode}
            out[key] = {k: v for k, v in d.items() if not isinstance(v, (list, dict)) or k in ("block_kernel", "csr_expansion", "checks")}
        except Exception as exc:  # noqa: BLE001
            out[key] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:200])}
    try:
        n = 1 << 14
        a = uniform_csr(torch, n, 32, 11, dev)
        b = uniform_csr(torch, n, 32, 12, dev)
        av, bv = a[2].double(), b[2].double()
        ha = abi.create("d", a[0], a[1], av, n, n)
        hb = abi.create("d", b[0], b[1], bv, n, n)
        C = torch.empty((n, n), device=dev, dtype=torch.float64)
        ts = []
        for rep in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            abi.check(abi.MI.call("mi_sparse_d_spmmd", 10, ha, hb, 101, C.data_ptr(), n), "spmmd")
            torch.cuda.synchronize()
            if rep:
                ts.append(time.perf_counter() - t0)
        t = _median(ts)
        ones = torch.ones(n, device=dev, dtype=torch.float64)
        b1, ab1 = torch.empty(n, device=dev, dtype=torch.float64), torch.empty(n, device=dev, dtype=torch.float64)
        abi.mv("d", hb, ones, b1)
        abi.mv("d", ha, b1, ab1)
        torch.cuda.synchronize()
        err = float(((C.sum(1) - ab1).abs() / ab1.abs().clamp(min=1e-300)).max())
        nbytes = (int(a[1].numel()) + int(b[1].numel())) * 12 + n * n * 8
        out["spmmd"] = {"workload": "sparse x sparse -> dense (mkl_sparse_d_spmmd): two uniform CSR 2^14 x 2^14, 32 / row fp64, 2 GiB row-major result",
                        "ms": round(t * 1e3, 3), "rowsum_max_rel_err": err,
                        "roofline": {"bound": "hbm", "achieved": round(nbytes / t / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": round(nbytes / t / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes": nbytes,
                                     "note": "(nnz(A) + nnz(B)) * 12 + M N * 8 (the dense result written once), median of 3 after 1"}}
        for h in (ha, hb):
            abi.destroy(h)
        if with_cpu:
            import numpy as np
            import scipy.sparse as sps
            ah = sps.csr_matrix((av.cpu().numpy(), a[1].cpu().numpy(), a[0].cpu().numpy()), shape=(n, n))
            bh = sps.csr_matrix((bv.cpu().numpy(), b[1].cpu().numpy(), b[0].cpu().numpy()), shape=(n, n))
            del C
            prods = float((np.bincount(ah.indices, minlength=n).astype(np.float64) * np.diff(bh.indptr)).sum())

            out["spmmd"]["cpu_baseline"] = cpu_baseline_mkl_child("spmmd", [ah, bh], 2 * prods,
                                                                   "full workload (mkl_sparse_d_spmmd into a preallocated 2 GiB array)", nrep=3)
            out["spmmd"]["value"] = round(2 * prods / t / 1e9, 2)
            out["spmmd"]["unit"] = "GFLOP/s"
    except Exception as exc:  # noqa: BLE001
        out["spmmd"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:200])}
    return out


def secondary_gemm(torch, abi, dev, with_cpu):
    """SURVEY section 8 a7 / a8 (the dense x dense fallback, the only MFMA consumer): mi_cblas_sgemm / mi_cblas_dgemm on
    device-resident 4096^3 operands against the dense MFMA peak; numpy's BLAS on the host beside it (timed AFTER both device
    measurements: its threads keep spinning on the host cores for a while and slow the launches of whatever follows)."""
    m = n = k = 4096
    out = {"workload": "dense x dense fallback (cblas_?gemm, reference _dense_dense.py:53-66): 4096 x 4096 x 4096, row-major, device-resident"}
    host = {}
    for letter, tdt, peak in (("s", torch.float32, 157.3), ("d", torch.float64, 78.6)):
        A = torch.rand((m, k), device=dev, dtype=tdt)
        Bm = torch.rand((k, n), device=dev, dtype=tdt)
        C = torch.empty((m, n), device=dev, dtype=tdt)

        def step():
            abi.check(abi.MI.call("mi_cblas_%sgemm" % letter, 101, 111, 111, m, n, k, 1.0, A.data_ptr(), k, Bm.data_ptr(), n, 0.0,
                                  C.data_ptr(), n), "gemm")
        step()
        torch.cuda.synchronize()
        rows = torch.randint(0, m, (16,), device=dev)
        want = A[rows].double() @ Bm.double()
        err = float(((C[rows].double() - want).abs() / want.abs().clamp(min=1e-300)).max())
        ok = err <= (1e-5 if letter == "s" else 1e-12)  # (recorded, not asserted: 4096-term fp32 sums sit within a factor 3 of the bar)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            step()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 5 / 1e3
        tf = 2.0 * m * n * k / t / 1e12
        key = "f32" if letter == "s" else "f64"
        out[key] = {"ms": round(t * 1e3, 3), "value": round(tf, 2), "unit": "TFLOP/s", "max_rel_err_rows": err, "within_tolerance": bool(ok),
                    "roofline": {"bound": "mfma", "achieved": round(tf, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(tf / peak, 4),
                                 "note": "dense %s MFMA peak (MI355X_MICROARCH.md); device time of 5 calls (events on the launch stream)" % key}}
        if with_cpu:
            host[key] = (A.cpu().numpy(), Bm.cpu().numpy())
        del A, Bm, C
    for key, (ah, bh) in host.items():
        import numpy as np
        np.dot(ah[:512], bh)
        t0 = time.perf_counter()
        np.dot(ah, bh)
        tc = time.perf_counter() - t0
        out[key]["cpu_baseline"] = {"value": round(2.0 * m * n * k / tc / 1e12, 3), "unit": "TFLOP/s", "cores": _host_cores(), "kind": "port",
                                    "sample": "numpy.dot of the same operands on the host (its BLAS, all threads), one call after a warm-up"}
    return out


def secondary_gram(torch, abi, dev, with_cpu):
    """BASELINE configs[3] as literally stated: A^T A of a uniform CSR 4 M x 262144, 64/row fp32, dense output
    (256 GiB, upper triangle written).  Falls back to half the width if the output cannot be allocated."""
    m = 1 << 22
    note = ""
    for ncols in (262144, 131072, 65536):
        try:
            need = ncols * ncols * 4 + (8 << 30)
            for _ in range(40):  # memory released by the previous secondary (and its child process) comes back asynchronously
                if torch.cuda.mem_get_info(dev)[0] >= need:
                    break
                time.sleep(0.5)
            C = torch.zeros((ncols, ncols), device=dev, dtype=torch.float32)
            break
        except Exception as e:  # noqa: BLE001
            note += "n=%d: %s; " % (ncols, str(e)[:100])
            C = None
    if C is None:
        return {"workload": "gram cfg4", "error": note}
    ip, idx, val, _ = uniform_csr(torch, m, 64, 3, dev, ncols=ncols)
    torch.cuda.empty_cache()  # the generator's temporaries back to the driver: the library caches ~9 GB of tables next to the 256 GiB output
    h = abi.create("s", ip, idx, val, m, ncols)
    times = []
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        abi.check(abi.MI.call("mi_sparse_s_syrkd", 11, h, 1.0, 0.0, C.data_ptr(), 101, ncols), "syrkd")
        torch.cuda.synchronize()
        if rep:
            times.append(time.perf_counter() - t0)
        else:
            first_call = time.perf_counter() - t0
    t = min(times)
    # a FRESH handle in the now-warm process (what every later gram_matrix_mkl call pays: the released tables of the previous
    # handle come back from the library's block cache instead of hipMalloc, the scratch arena is grown)
    abi.destroy(h)
    h = abi.create("s", ip, idx, val, m, ncols)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    abi.check(abi.MI.call("mi_sparse_s_syrkd", 11, h, 1.0, 0.0, C.data_ptr(), 101, ncols), "syrkd")
    torch.cuda.synchronize()
    fresh_warm = time.perf_counter() - t0
    colsq = torch.zeros(ncols, device=dev, dtype=torch.float64)
    for lo in range(0, int(idx.numel()), 1 << 24):  # in pieces: next to the 256 GiB output and the library's tables ~1 GiB is free
        colsq.index_add_(0, idx[lo:lo + (1 << 24)].long(), val[lo:lo + (1 << 24)].double() ** 2)
    diag_err = float(((torch.diagonal(C).double() - colsq).abs() / colsq.clamp(min=1e-30)).max())
    lower_zero = bool((torch.tril(C[-4096:, -4096:], -1) == 0).all()) and bool((C[-4096:, :4096] == 0).all())
    lens = (ip[1:] - ip[:-1]).double()
    flops = float((lens * (lens + 1)).sum())  # 2 * sum r (r + 1) / 2
    nnz = int(idx.numel())
    nbytes = nnz * 8 + (m + 1) * 8 + ncols * (ncols + 1) // 2 * 4
    out = {"workload": "gram (BASELINE configs[3]%s): A^T A, uniform CSR 2^22 x %d, 64/row (%d nnz) fp32, dense output "
                       "(%.0f GiB array, upper triangle written)" % (" as literally stated" if ncols == 262144 else
                                                                     ", NARROWER than stated: " + note, ncols, nnz,
                                                                     ncols * ncols * 4 / 2**30),
           "ms": round(t * 1e3, 2), "value": round(flops / t / 1e9, 2), "unit": "GFLOP/s", "dtype": "f32",
           "first_call_ms": round(first_call * 1e3, 2),
           "fresh_handle_warm_process_ms": round(fresh_warm * 1e3, 2),
           "first_call_note": "first_call_ms: the FIRST mi_sparse_s_syrkd of the process (cold block cache: its tables are hipMalloc'ed); "
                              "fresh_handle_warm_process_ms: the first call of a second fresh handle after the first was destroyed -- the "
                              "tables come back from the block cache, the scratch arena is grown: transpose + tables + product.  "
                              "mi_sparse_s_syrkd on a FRESH handle is what the public gram_matrix_mkl pays on every call, as it "
                              "creates its handle per call like the reference (_gram_matrix.py:121-127): the transpose of X "
                              "(round 5: a stable radix sort of the entries, 2 passes here), the tile tables and the packed "
                              "records, then the product; `ms` is the product alone on the same handle.  Device kernels of the first call: ~68 ms "
                              "(profiles/r05_gram_first_call_kernel_stats.log); the rest is hipMalloc of ~16 GB of transpose scratch and "
                              "tables, whose cost depends on whether the driver hands out recycled pages (NOTEBOOK 3.2, alloc probe)",
           "roofline": {"bound": "hbm", "achieved": round(nbytes / t / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(nbytes / t / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes": nbytes,
                        "note": "nnz * 8 + (M + 1) * 8 + n (n + 1) / 2 * 4 (the triangle written once) over the "
                                "mi_sparse_s_syrkd call (best of 2 after 1 warm-up)"},
           "parity_diag_max_rel_err": diag_err, "lower_triangle_untouched_sample": lower_zero,
           "kernel": abi.sda.mi_get_last_kernel()}
    assert diag_err <= 1e-5 and lower_zero, "gram parity check failed (%g, %s)" % (diag_err, lower_zero)
    abi.destroy(h)
    del C
    if with_cpu:
        import scipy.sparse as sps
        sip, sidx, sval, _ = uniform_csr(torch, 1 << 18, 64, 3, dev, ncols=16384)
        a = sps.csr_matrix((sval.cpu().numpy(), sidx.cpu().numpy(), sip.cpu().numpy()), shape=(1 << 18, 16384))
        out["cpu_baseline"] = cpu_baseline_gram(a, nrep=3)
        # the like-for-like partner of that CPU number: the SAME sample (2^18 x 16384, 64/row, dense 1 GiB output) on the GPU
        hs = abi.create("s", sip, sidx, sval, 1 << 18, 16384)
        Cs = torch.zeros((16384, 16384), device=dev, dtype=torch.float32)
        ts = []
        for rep in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            abi.check(abi.MI.call("mi_sparse_s_syrkd", 11, hs, 1.0, 0.0, Cs.data_ptr(), 101, 16384), "syrkd")
            torch.cuda.synchronize()
            if rep:
                ts.append(time.perf_counter() - t0)
        slens = (sip[1:] - sip[:-1]).double()
        sflops = float((slens * (slens + 1)).sum())
        out["gpu_on_cpu_sample_shape"] = {"workload": "A^T A, uniform CSR 2^18 x 16384, 64/row fp32, dense 1 GiB output (the cpu_baseline sample)",
                                          "ms": round(min(ts) * 1e3, 3), "value": round(sflops / min(ts) / 1e9, 2), "unit": "GFLOP/s"}
        abi.destroy(hs)
        del Cs
    return out


def secondary_config5(torch, dist, dev, abi, rank, world, allreduce_max, steps=3):
    """BASELINE configs[4]: R-MAT CSR 2^24 x 2^24 (scale 24, 32 edges/row, ~5.2e8 nnz) x dense 2^24 x 256 fp32, the left matrix
    row-partitioned over the 8 ranks (every rank generates the same matrix and keeps its nnz-balanced block), B broadcast
    from rank 0, C all-gathered -- timed end to end like the headline step, plus the kernels alone."""
    from sparse_dot_amd import distributed as D
    scale, N = int(os.environ.get("BENCH_CFG5_SCALE", "24")), 256
    indptr, indices, vals, n = rmat_csr(torch, scale, 32, 7, dev)
    nnz = int(indices.numel())
    ip64 = indptr.to(torch.int64)
    del indptr
    bounds = D.partition_rows(ip64.cpu().numpy(), world, dense_bytes=n * N * 4)
    r0, r1 = int(bounds[rank]), int(bounds[rank + 1])
    lo, hi = int(ip64[r0]), int(ip64[r1])
    blk_ptr = (ip64[r0:r1 + 1] - lo).to(torch.int32).contiguous()
    blk_idx, blk_val = indices[lo:hi].clone(), vals[lo:hi].clone()
    sample = []
    if rank == 0:  # rows of every block, checked against fp64 after the timed steps
        g = torch.Generator(device="cpu")
        g.manual_seed(3)
        for r in torch.randint(0, n, (24,), generator=g).tolist() + [int(b) for b in bounds[:-1]]:
            a, b = int(ip64[r]), int(ip64[r + 1])
            sample.append((r, indices[a:b].clone(), vals[a:b].clone()))
    del indices, vals, ip64
    torch.cuda.empty_cache()
    h = abi.create("s", blk_ptr, blk_idx, blk_val, r1 - r0, n)
    gb = torch.Generator(device=dev)
    gb.manual_seed(9)
    B = torch.rand((n, N), generator=gb, device=dev, dtype=torch.float32) if rank == 0 else torch.zeros((n, N), device=dev)
    C = torch.empty((n, N), device=dev, dtype=torch.float32)

    def barrier():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    def compute():
        if r1 > r0:
            abi.mm("s", h, B, C[r0:r1], N)

    def step():
        D.broadcast_rows(B, 0, None, "scatter_allgather").wait()
        compute()
        D.gather_rows(C, bounds, None, "p2p")

    def timed(fn, k):
        barrier()
        t0 = time.perf_counter()
        for _ in range(k):
            fn()
        barrier()
        return (time.perf_counter() - t0) / k
    step()
    step()
    t_e2e = allreduce_max(timed(step, steps))
    compute()
    t_cmp = allreduce_max(timed(compute, steps))
    t_b = allreduce_max(timed(lambda: D.broadcast_rows(B, 0, None, "scatter_allgather").wait(), 2))
    t_g = allreduce_max(timed(lambda: D.gather_rows(C, bounds, None, "p2p"), 2))
    worst = 0.0
    for r, cols, v in sample:
        if cols.numel() == 0:
            worst = max(worst, float(C[r].abs().max()))
            continue
        want = (v.double()[:, None] * B[cols.long()].double()).sum(0)
        worst = max(worst, float(((C[r].double() - want).abs() / want.abs().clamp(min=1e-30)).max()))
    assert worst < 1e-5, "configs[4] fails the fp32 parity bar: %g" % worst
    abi.destroy(h)
    return {"workload": "%sR-MAT CSR %dx%d (%d nnz) x dense %dx%d fp32, %d nnz-balanced row blocks"
                        % ("BASELINE configs[4]: " if scale == 24 else "DRY RUN at scale %d of BASELINE configs[4]: " % scale, n, n, nnz, n, N, world),
            "value": round(2.0 * nnz * N / t_e2e / 1e9, 2), "unit": "GFLOP/s", "end_to_end_ms": round(t_e2e * 1e3, 3),
            "compute_only_ms": round(t_cmp * 1e3, 3), "compute_only_value": round(2.0 * nnz * N / t_cmp / 1e9, 2),
            "bcast_scatter_allgather_ms": round(t_b * 1e3, 3), "gather_p2p_ms": round(t_g * 1e3, 3),
            "parity_max_rel_err_sample": worst, "block_rows_rank0": r1 - r0,
            "note": "step = scatter + all-gather broadcast of B (17.2 GB) -> local kernel -> point-to-point all-gatherv of C "
                    "(17.2 GB), max over ranks, %d steps; compute_only = the kernels alone" % steps}


def host_api_figure(sda):
    """dot_product_mkl on HOST arrays (the reference's calling convention), one call each: handle created and
    destroyed inside, operands cross PCIe both ways."""
    import numpy as np
    import scipy.sparse as sps
    rng = np.random.default_rng(0)
    n2 = 1 << 18
    ind = np.sort(rng.integers(0, n2, (n2, 32)), axis=1).astype(np.int32).ravel()
    a2 = sps.csr_matrix((rng.standard_normal(ind.size).astype(np.float32), ind, np.arange(0, ind.size + 1, 32)), shape=(n2, n2))
    b2 = rng.standard_normal((n2, 128)).astype(np.float32)
    a1 = sps.random(10000, 10000, density=0.001, format="csr", random_state=1, dtype=np.float64)
    b1 = rng.standard_normal((10000, 64))
    out = {}
    for label, a, b in (("cfg1_10k_fp64_x64_ms", a1, b1), ("rows2e18_8.4Mnnz_fp32_x128_ms", a2, b2)):
        ts = _timed(lambda: sda.dot_product_mkl(a, b), 5, warm=1)
        out[label] = round(_median(ts) * 1e3, 3)
    out["note"] = "median of 5 single calls of dot_product_mkl(a_scipy, b_numpy): H2D of A and B, plan, kernel, D2H of C"
    return out


# ------------------------------------------------------------------------------------------------
# the north_star's partitioned SpMM (N >= 1 ranks); device agnostic so that the gloo test can drive it on CPU
# ------------------------------------------------------------------------------------------------
def run_partitioned(torch, dist, dev, indptr, indices, vals, n, B, steps, warmup, make_local, gather_mode="bcast",
                    sync=None, group=None, bcast_mode="bcast", variants=True, gather_group=None, panels=4):
    """Split (indptr, indices, vals) into world row blocks, keep this rank's block resident, and time
    (a) the local kernels alone and (b) the end-to-end step  bcast(B) -> kernel -> all-gatherv(C).

    make_local(indptr_blk, indices_blk, vals_blk, rows, cols) -> (mm(B, C_slice), free())."""
    from sparse_dot_amd import distributed as D
    rank = dist.get_rank(group) if dist else 0
    world = dist.get_world_size(group) if dist else 1
    sync = sync or (lambda: None)
    ip64 = indptr.to(torch.int64)
    bounds = D.partition_rows(ip64.cpu().numpy(), world, dense_bytes=int(B.numel()) * B.element_size())
    r0, r1 = int(bounds[rank]), int(bounds[rank + 1])
    lo, hi = int(ip64[r0]), int(ip64[r1])
    blk_ptr = (ip64[r0:r1 + 1] - lo).to(torch.int32).contiguous()
    blk_idx, blk_val = indices[lo:hi].contiguous(), vals[lo:hi].contiguous()
    mm, free = make_local(blk_ptr, blk_idx, blk_val, r1 - r0, n)
    N = B.shape[1]
    C = torch.zeros((n, N), dtype=B.dtype, device=dev)
    mine = C[r0:r1]

    def barrier():
        sync()
        if dist:
            dist.barrier(group=group)
        sync()

    def step_compute():
        if r1 > r0:
            mm(B, mine)

    def make_step(bmode, gmode):
        def step():
            if dist and world > 1:
                D.broadcast_rows(B, 0, group, bmode).wait()
            step_compute()
            if dist and world > 1:
                D.gather_rows(C, bounds, group, gmode)
        return step
    step_end_to_end = make_step(bcast_mode, gather_mode)

    def timed_loop(fn, k):
        barrier()
        t0 = time.perf_counter()
        for _ in range(k):
            fn()
        barrier()
        return (time.perf_counter() - t0) / k

    for _ in range(warmup):
        step_end_to_end()
    t_e2e = timed_loop(step_end_to_end, steps)  # THE timed region of the contract
    for _ in range(max(1, warmup)):
        step_compute()
    t_cmp = timed_loop(step_compute, steps)
    res = {"bounds": bounds, "block_rows": r1 - r0, "block_nnz": hi - lo, "t_end_to_end": t_e2e, "t_compute": t_cmp, "C": C,
           "free": free, "mm": mm, "timed_loop": timed_loop, "make_step": make_step}
    if dist and world > 1:
        # B already replicated (an iterative caller re-using one dense operand, or B produced in place on every rank):
        # step = local kernel -> all-gatherv(C)
        def step_resident():
            step_compute()
            D.gather_rows(C, bounds, group, gather_mode)
        for _ in range(max(1, warmup)):
            step_resident()
        res["t_resident_B"] = timed_loop(step_resident, steps)
        k4 = max(2, steps // 4)
        res["t_bcast"] = timed_loop(lambda: D.broadcast_rows(B, 0, group, "bcast").wait(), k4)
        res["t_gather_bcast"] = timed_loop(lambda: D.gather_rows(C, bounds, group, "bcast"), k4)
        res["t_gather_padded"] = timed_loop(lambda: D.gather_rows(C, bounds, group, "padded"), k4)
    return res


def run_variants(torch, dist, dev, res, n, B, steps, warmup, group=None, gather_group=None, panels=4, check=None):
    """The xGMI-shaped forms of the same step (sparse_dot_amd/distributed.py), each timed like the contract's loop
    (barrier + synchronise on both sides of exactly `steps` steps):
      p2p        scatter + all-gather broadcast of B, kernel, all-gatherv of C as one grouped send / receive batch
      pipelined  B and C held as `panels` column panels: broadcast(p + 2) | kernel(p) | all-gatherv(p - 1) overlap
    Returns {name: seconds per step} plus the single collectives; every variant's C is checked by `check`."""
    from sparse_dot_amd import distributed as D
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    bounds, C, mm, timed_loop = res["bounds"], res["C"], res["mm"], res["timed_loop"]
    out = {}
    k4 = max(2, steps // 4)
    if rank != 0:
        B.zero_()  # only the root holds B: the checks below then also prove the broadcasts
    out["t_bcast_scatter_allgather"] = timed_loop(lambda: D.broadcast_rows(B, 0, group, "scatter_allgather").wait(), k4)
    out["t_gather_p2p"] = timed_loop(lambda: D.gather_rows(C, bounds, group, "p2p"), k4)
    step = res["make_step"]("scatter_allgather", "p2p")
    C.zero_()
    for _ in range(max(1, warmup)):
        step()
    out["p2p"] = timed_loop(step, steps)
    if check:
        check(C, "p2p")
    N = B.shape[1]
    P = panels if (N % panels == 0 and (N // panels) % 4 == 0) else 1
    if P > 1:
        w = N // P
        Bp = B.view(n, P, w).permute(1, 0, 2).contiguous()  # panel-major copy of B (valid on rank 0, zeros elsewhere)
        Cp = torch.zeros((P, n, w), dtype=B.dtype, device=dev)
        if rank != 0:
            Bp.zero_()

        def pstep():
            D.pipelined_panels(mm, Bp, Cp, bounds, rank, src=0, group=group, bcast_mode="scatter_allgather",
                               gather_mode="p2p", gather_group=gather_group, depth=2)
        for _ in range(max(1, warmup)):
            pstep()
        out["pipelined"] = timed_loop(pstep, steps)
        out["pipelined_panels"] = P
        if check:
            check(Cp.permute(1, 0, 2).reshape(n, N), "pipelined")
        del Bp, Cp
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="rmat", choices=["rmat", "uniform"])
    ap.add_argument("--scale", type=int, default=20)
    ap.add_argument("--ncols", type=int, default=128)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline legs")
    ap.add_argument("--no-pmc", action="store_true", help="do not collect roofline.traffic with rocprofv3 in this run (use the committed summary)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary workloads")
    ap.add_argument("--secondary", default="all", help="comma list of: uniform,spgemm,spgemm_rmat,gram,host_api,gemm,spmv,rows (default all)")
    ap.add_argument("--gather-mode", default="bcast", choices=["bcast", "padded", "p2p"])
    ap.add_argument("--bcast-mode", default="bcast", choices=["bcast", "scatter_allgather"])
    ap.add_argument("--no-variants", action="store_true", help="N > 1: skip the p2p / pipelined forms and configs[4]")
    ap.add_argument("--panels", type=int, default=4, help="N > 1: column panels of the pipelined form")
    ap.add_argument("--variant-timeout", type=int, default=240)
    ap.add_argument("--config5-timeout", type=int, default=900)
    ap.add_argument("--chunk", type=int, default=0, help="override the SpMM work-item chunk (tuning)")
    ap.add_argument("--unroll", type=int, default=0, help="override the SpMM load unroll 4|8 (tuning)")
    ap.add_argument("--hot-kb", type=int, default=-1, help="override the hot-set budget in KiB, 0 = no tagging (tuning)")
    ap.add_argument("--slices", type=int, default=-1, help="override the column slices 0 (auto) | 1 | 2 | 4 | 8 (tuning)")
    args = ap.parse_args()

    import numpy as np  # noqa: F401
    import torch

    import sparse_dot_amd as sda

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
    devices_seen = None
    if dist is not None:
        # every rank's (hostname-independent) device identity: index + PCI bus id, gathered once
        try:
            props = torch.cuda.get_device_properties(dev_index)
            ident = "%d:%s" % (dev_index, getattr(props, "pci_bus_id", getattr(props, "uuid", "?")))
        except Exception:  # noqa: BLE001
            ident = str(dev_index)
        gathered = [None] * world
        dist.all_gather_object(gathered, ident)
        devices_seen = {"distinct": len(set(gathered)), "by_rank": gathered}

    sda.mi_set_device(dev_index)
    stream = torch.cuda.current_stream()
    sda.mi_set_stream(stream.cuda_stream)
    if args.chunk:
        sda.mi_set_option("spmm_chunk", args.chunk)
    if args.unroll:
        sda.mi_set_option("spmm_unroll", args.unroll)
    if args.hot_kb >= 0:
        sda.mi_set_option("spmm_hot_kb", args.hot_kb)
    if args.slices >= 0:
        sda.mi_set_option("spmm_slices", args.slices)
    for kv in os.environ.get("MI_BENCH_OPTS", "").split(","):  # tuning hook, e.g. MI_BENCH_OPTS=pool_enable=0
        if "=" in kv:
            sda.mi_set_option(kv.split("=")[0], int(kv.split("=")[1]))
    abi = Abi()

    # ---- synthetic inputs, generated on the device (every rank generates the SAME matrix: one problem, partitioned) ----
    N = args.ncols
    if args.workload == "rmat":
        indptr, indices, vals, n = rmat_csr(torch, args.scale, 32, 7, dev)
    else:
        indptr, indices, vals, n = uniform_csr(torch, 1 << args.scale, 32, 7, dev)
    nnz = int(indices.numel())
    gb = torch.Generator(device=dev)
    gb.manual_seed(9)
    B = torch.rand((n, N), generator=gb, device=dev, dtype=torch.float32)
    if rank != 0:
        B.zero_()  # only the root holds B; the others receive it in the timed step's broadcast
    torch.cuda.synchronize()

    handles = []

    def make_local(bp, bi, bv, rows, cols):
        h = abi.create("s", bp, bi, bv, rows, cols)
        handles.append((h, (bp, bi, bv)))
        return (lambda Bt, Ct: abi.mm("s", h, Bt, Ct, Bt.shape[1])), (lambda: abi.destroy(h))

    if dist and backend != "nccl":
        # gloo dry run (both ranks on one GPU): collectives staged through the host
        import sparse_dot_amd.distributed as D
        _orig_gather = D.gather_rows

        def _gather_host(full, bounds, group=None, mode="bcast"):
            hfull = full.cpu()
            _orig_gather(hfull, bounds, group, mode)
            full.copy_(hfull)
            return full
        D.gather_rows = _gather_host
        _orig_brows, _orig_p2p = D.broadcast_rows, D.gather_rows_p2p

        def _brows_host(t, src=0, group=None, mode="scatter_allgather"):
            hcopy = t.cpu()
            _orig_brows(hcopy, src, group, mode).wait()
            t.copy_(hcopy)
            return D._Pending()

        def _p2p_host(full, bounds, group=None):
            hfull = full.cpu()
            _orig_p2p(hfull, bounds, group).wait()
            full.copy_(hfull)
            return D._Pending()
        D.broadcast_rows, D.gather_rows_p2p = _brows_host, _p2p_host

    # ---- first call on a fresh handle vs steady state (single GPU only: the inspector's visible cost) ----
    plan = None
    if world == 1:
        C0 = torch.empty((n, N), device=dev, dtype=torch.float32)

        def first_calls():
            h0 = abi.create("s", indptr, indices, vals, n, n)
            torch.cuda.synchronize()
            ts = []
            for _ in range(6):
                t0 = time.perf_counter()
                abi.mm("s", h0, B, C0, N)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            abi.destroy(h0)
            return ts
        cold = first_calls()   # first handle of the process: also pays the scratch arena / block cache / pinned slab set-up
        warm = first_calls()   # a fresh handle in a warm process: what every later handle pays

        def single_call(optimize):
            """create (device pointers) [+ mi_sparse_optimize] + ONE product + destroy on a fresh handle: what a caller of the
            public dot_product_mkl(scipy_matrix, b) pays per call on the device side (before PCIe)."""
            out = []
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                h0 = abi.create("s", indptr, indices, vals, n, n)
                if optimize:
                    abi.check(abi.MI.call("mi_sparse_optimize", h0), "optimize")
                t1 = time.perf_counter()
                abi.mm("s", h0, B, C0, N)
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                abi.destroy(h0)
                out.append(((t2 - t0) * 1e3, (t1 - t0) * 1e3, (t2 - t1) * 1e3))
            out.sort()
            return out[1]
        single = single_call(False)
        single_opt = single_call(True)
        del C0
        plan = {"first_call_ms": round(warm[0], 3), "second_call_ms": round(warm[1], 3),
                "third_call_ms_incl_column_partition_build": round(warm[2], 3),
                "later_calls_ms": [round(x, 3) for x in warm[3:]],
                "plan_ms": round(max(0.0, warm[0] - warm[1]), 3),
                "column_partition_build_ms": round(max(0.0, warm[2] - min(warm[3:])), 3),
                "first_call_ms_cold_process": round(cold[0], 3),
                "single_call_ms": round(single[0], 3),
                "single_call_after_optimize": {"create_plus_optimize_ms": round(single_opt[1], 3), "product_ms": round(single_opt[2], 3),
                                               "total_ms": round(single_opt[0], 3)},
                "single_call_note": "single_call_ms = mi_sparse_s_create_csr (device pointers: aliased) + ONE mi_sparse_s_mm + synchronise "
                                    "on a fresh handle, median of 3: the device side of one public dot_product_mkl(scipy_matrix, b) -- "
                                    "the row-owned kernel plus its plan.  single_call_after_optimize = the same with mi_sparse_optimize "
                                    "(to_device(a, optimize=True)) between create and product: the inspector's cost up front, the "
                                    "product at the steady-state rate",
                "note": "one synchronised mi_sparse_s_mm per entry on a FRESH handle.  Calls 1-2 = the row-owned kernel (call 1 "
                        "also builds the row partition + fix-up schedule: plan_ms = call 1 - call 2); two products prove the "
                        "reuse, so call 3 builds the column-partitioned form of the long rows (three scans + two fill passes, "
                        "synchronous) and runs with it, as every later call does; the short-row half adopts its hot / cold "
                        "column tags two calls after that.  A single-use handle pays for none of it.  *_cold_process = the very "
                        "first handle of the process (one-time arena / cache growth included)."}

    res = run_partitioned(torch, dist, dev, indptr, indices, vals, n, B, args.steps, args.warmup, make_local,
                          gather_mode=args.gather_mode, sync=torch.cuda.synchronize, bcast_mode=args.bcast_mode)
    C = res["C"]

    def allreduce_max(x):
        if not dist:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    t_step = allreduce_max(res["t_end_to_end"])
    t_cmp = allreduce_max(res["t_compute"])
    t_res = allreduce_max(res["t_resident_B"]) if "t_resident_B" in res else None  # a collective: every rank takes part
    ms_per_step = t_step * 1e3
    gflops = 2.0 * nnz * N / t_step / 1e9

    # ---- device time of one product (ALL its kernels: round 5 splits the product into short rows row-owned + long rows by
    # column partition + carry fix-ups + combine) with hipEvents on the launch stream (separate loop, this rank's block) ----
    sda.mi_set_option("profile_events", 1)
    sda.mi_get_counter("reset")
    for _ in range(max(5, min(args.steps, 20))):
        if res["block_rows"]:
            res["mm"](B, C[int(res["bounds"][rank]):int(res["bounds"][rank + 1])])
    torch.cuda.synchronize()
    k_ms = sda.mi_get_counter("spmm_kernel_ms") / max(1.0, sda.mi_get_counter("spmm_kernel_launches"))
    sda.mi_set_option("profile_events", 0)
    tagged = bool(sda.mi_get_counter("spmm_last_tagged"))
    slices = int(sda.mi_get_counter("spmm_last_slices"))
    kpart = int(sda.mi_get_counter("spmm_last_kpart"))
    long_share = round(sda.mi_get_counter("spmm_kpart_long_share"), 4)
    hot_coverage = round(sda.mi_get_counter("spmm_hot_coverage"), 4)
    blk_nnz, blk_rows = res["block_nnz"], res["block_rows"]
    alg_bytes = blk_nnz * 8 + (blk_rows + 1) * 8 + n * N * 4 + blk_rows * N * 4  # this rank's launch
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    kernel_name = sda.mi_get_last_kernel()  # what the library launched (not what this script expects)
    traffic, traffic_src = (None, None)
    if world == 1 and args.workload == "rmat" and N == 128 and args.scale == 20:
        if not args.no_pmc:
            traffic, traffic_src = measure_traffic_live()
        if traffic is None:
            why = traffic_src
            traffic, traffic_src = load_traffic("spmm_traffic.json")
            if traffic_src and why:
                traffic_src += " (committed PMC summary; live collection unavailable: %s)" % why
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "kernel": kernel_name, "kernel_ms": round(k_ms, 4), "algorithmic_bytes": alg_bytes,
                "kernel_ms_is": "device time of ONE product = the sum of its kernels (hipEvents on the launch stream around all "
                                "of them): achieved = algorithmic bytes / that"}
    try:  # what the reference stream of this product costs at the fabric (tools/l2_bound.py: a CPU replay through one 4 MiB cache per XCD)
        with open(os.path.join(ROOT, "profiles", "r06_l2_bound.json")) as fh:
            lb = json.load(fh)
        pt = lb["partitioned"]
        roofline["formulation_floor_ms"] = pt["floor_ms_at_6.2TBps"]["lru"]
        roofline["formulation_floor"] = {
            "lru_ms_at_6.2TBps": pt["floor_ms_at_6.2TBps"]["lru"], "lru_ms_at_7.4TBps": pt["floor_ms_at_7.4TBps"]["lru"],
            "belady_ms_at_6.2TBps": pt["floor_ms_at_6.2TBps"]["opt"], "belady_ms_at_7.4TBps": pt["floor_ms_at_7.4TBps"]["opt"],
            "fabric_GB_lru": pt["fabric_GB_lru"], "fabric_GB_belady": pt["fabric_GB_opt"],
            "row_owned_fabric_GB_lru": lb["row_owned"]["fabric_GB_lru"], "row_owned_fabric_GB_belady": lb["row_owned"]["fabric_GB_opt"],
            "floor_over_achieved": round(pt["floor_ms_at_6.2TBps"]["lru"] / (k_ms if k_ms > 0 else float("nan")), 3),
            "note": "tools/l2_bound.py (committed result: profiles/r06_l2_bound.json): the B-row references of THIS schedule (rows of >= 64 "
                    "entries by 8 column partitions on one XCD each, the rest row-owned in 2 column slices; 128-item chunks dealt to the XCDs "
                    "as k_spmm maps them, 256 chunks in flight per XCD) replayed on the CPU through one fully associative 4 MiB cache per "
                    "XCD.  LRU: the bytes a real L2 must fetch -- the model reproduces the counters (row-owned 10.2 GB against 10.3 GB "
                    "measured in round 4; partitioned 6.8 GB against 7.5 GB measured) -- plus A once, C once and the partial rows twice, "
                    "divided by the fabric rate the product's own kernels reach (6.2 TB/s) = the floor of this schedule on an LRU cache; "
                    "Belady: the same with the optimal replacement policy (no hardware has it).  Matrix: the bench's generator on the CPU "
                    "(same distribution, other random numbers)"}
    except Exception as exc:  # noqa: BLE001
        roofline["formulation_floor_ms"] = None
        roofline["formulation_floor"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:120])}
    if world == 1:
        torch_copy = measured_copy_gbs(torch, dev)
        try:  # the library's own tuned copy kernel (16 B / lane, 4 - 8 in flight, plain and non-temporal, 4 - 32 workgroups per CU)
            copy_gbs = max(torch_copy, float(sda.mi_probe_copy_gbs(1 << 31, 3)))
        except Exception:  # noqa: BLE001
            copy_gbs = torch_copy
        roofline["measured_copy_GBps"] = round(copy_gbs, 1)
        roofline["measured_copy_note"] = ("best of the library's tuned copy kernel (mi_sparse_probe_copy: 2 GiB buffers, 10 variants, hipEvents) "
                                          "and torch's tensor copy (%.0f GB/s); read + write bytes per second; the guide quotes 6.29 TB/s "
                                          "for a float4 copy" % torch_copy)
        roofline["frac_of_measured_copy"] = round(achieved / copy_gbs, 4)

    # ---- parity spot check of the timed configuration (a row sample of the gathered C vs fp64 on the GPU) ----
    worst = 0.0
    if rank == 0:
        Bfull = B
        ip = indptr.to(torch.int64)
        lens = ip[1:] - ip[:-1]
        mid = torch.nonzero((lens >= 64) & (lens < 128)).flatten()
        # the 8 longest rows, 8 rows just past the column-partition threshold (64 .. 127 entries), 24 random ones: the long-row
        # kernels the timed steps ran are sampled on purpose (64 random rows of a power-law matrix contain ~3 long ones)
        sel = torch.cat([torch.topk(lens, min(8, n)).indices, mid[torch.randperm(mid.numel(), device=dev)[:8]],
                         torch.randint(0, n, (24,), device=dev)])
        for r in sel.tolist():
            lo, hi = int(ip[r]), int(ip[r + 1])
            want = (vals[lo:hi].double()[:, None] * Bfull[indices[lo:hi].long()].double()).sum(0)
            got = C[r].double()
            den = torch.clamp(want.abs(), min=1e-30)
            worst = max(worst, float(((got - want).abs() / den).max())) if hi > lo else max(worst, float(got.abs().max()))
        assert worst < 1e-5, "bench result fails the fp32 parity bar: %g" % worst

    line = None
    if rank == 0:
        line = {
            "metric": "spmm_effective_gflops", "value": round(gflops, 2), "unit": "GFLOP/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s CSR %dx%d (%d nnz, int32 indices) x dense %dx%d fp32, C := A @ B"
                                   % ("R-MAT(.57,.19,.19,.05) scale %d, 32 edges/row, dedup" % args.scale
                                      if args.workload == "rmat" else "uniform 32/row", n, n, nnz, n, N),
                       "partition": ("one matrix, %d contiguous nnz-balanced row blocks (partition_rows), one per GPU; step = "
                                     "RCCL bcast(B) -> local kernel -> all-gatherv(C) [%s]" % (world, args.gather_mode))
                       if world > 1 else "single GPU (step = one product: its kernels + carry fix-ups)",
                       "spmm_chunk": args.chunk or 256, "column_slices": slices,
                       "hot_cold_tagged_gather": tagged, "hot_column_coverage": hot_coverage,
                       "column_partitions_of_long_rows": kpart, "nonzeros_in_partitioned_rows": long_share},
            "compute_only_ms": round(t_cmp * 1e3, 4), "end_to_end_ms": round(ms_per_step, 4),
            "roofline": roofline, "parity_max_rel_err_sample": worst,
        }
        if plan:
            line["plan"] = plan
        if world > 1:
            line["collectives"] = {k: round(res[k] * 1e3, 3) for k in ("t_bcast", "t_gather_bcast", "t_gather_padded") if k in res}
            line["collectives"]["backend"] = backend
            # what the process group itself reports (not what --gpus asked for): ranks, the backend torch resolved, and the
            # devices behind the ranks (gathered): a line from "8 ranks on 1 GPU" or from gloo cannot pass as an 8-GPU RCCL run
            line["collectives"]["ranks_seen"] = int(dist.get_world_size())
            line["collectives"]["backend_seen"] = str(dist.get_backend())
            line["collectives"]["devices_seen"] = devices_seen
            line["collectives"]["note"] = ("ms, this rank; bcast = B (%.0f MB) from rank 0; gather_bcast = all-gatherv as one "
                                           "broadcast per rank into its slice of C (no padding); gather_padded = pad to the "
                                           "tallest block + all_gather_into_tensor" % (n * N * 4 / 1e6))
            line["compute_only_value"] = round(2.0 * nnz * N / t_cmp / 1e9, 2)
            line["block_rows_rank0"] = blk_rows
            line["resident_B_ms"] = round(t_res * 1e3, 4)
            line["resident_B_value"] = round(2.0 * nnz * N / t_res / 1e9, 2)
            line["scaling_basis"] = "end_to_end"
            line["scaling_note"] = ("`value` times bcast(B, %.0f MB) -> kernel -> all-gatherv(C, %.0f MB) per step on ONE partitioned "
                                    "matrix: at configs[1] the two collectives outweigh a kernel of ~%.2f ms per rank, so `value` "
                                    "measures xGMI, not the kernels.  Kernel scaling = compute_only_value (max over ranks of the "
                                    "local kernels); resident_B_value = kernel + all-gatherv with B already on every rank; the "
                                    "workload where 1-D row blocks scale end to end is secondary.spmm_config5_8gpu."
                                    % (n * N * 4 / 1e6, n * N * 4 / 1e6, t_cmp * 1e3))

    # ---- N > 1: the xGMI-shaped forms of the same step, and BASELINE configs[4] on 8 ranks ----
    # The contract line above is complete at this point.  The forms below drive RCCL paths that no build session could
    # run (one GPU there): a watchdog prints the line as it stands and ends the process if they do not come back.
    if dist and world > 1 and not args.no_variants:
        import threading

        def _bail():
            if rank == 0:
                line.setdefault("variants", {})["error"] = "timed out after %d s; the contract line is the basic form" % args.variant_timeout
                print(json.dumps(line), flush=True)
            os._exit(0)
        dog = threading.Timer(args.variant_timeout, _bail)
        dog.daemon = True
        dog.start()
        try:
            def check(Cx, name):
                if rank != 0:
                    return
                sel = torch.randint(0, n, (32,), device=dev)
                ip = indptr.to(torch.int64)
                for r in sel.tolist():
                    lo, hi = int(ip[r]), int(ip[r + 1])
                    if hi == lo:
                        continue
                    want = (vals[lo:hi].double()[:, None] * Bref[indices[lo:hi].long()].double()).sum(0)
                    err = float(((Cx[r].double() - want).abs() / want.abs().clamp(min=1e-30)).max())
                    assert err < 1e-5, "variant %s fails the fp32 parity bar: %g" % (name, err)
            Bref = B.clone() if rank == 0 else None
            ggroup = dist.new_group(list(range(world))) if backend == "nccl" else None  # second communicator: the all-gathervs get their own stream
            var = run_variants(torch, dist, dev, res, n, B, args.steps, args.warmup, gather_group=ggroup, panels=args.panels,
                               check=check)
            tv = {k: allreduce_max(v) for k, v in sorted(var.items()) if k != "pipelined_panels"}
            if rank == 0:
                forms = {"basic": t_step}
                forms.update({k: tv[k] for k in ("p2p", "pipelined") if k in tv})
                # only a form with the basic form's interface may replace `value`: p2p takes and returns the same row-major
                # (n, N) arrays; pipelined keeps B and C as column panels (the re-layout sits outside its timed loop) and is
                # reported here only
                same_layout = {k: v for k, v in forms.items() if k in ("basic", "p2p")}
                best = min(same_layout, key=same_layout.get)
                line["variants"] = {
                    "end_to_end_ms": {k: round(v * 1e3, 4) for k, v in forms.items()},
                    "value_GFLOPs": {k: round(2.0 * nnz * N / v / 1e9, 2) for k, v in forms.items()},
                    "best": best, "pipelined_panels": var.get("pipelined_panels"),
                    "note": "basic = dist.broadcast(B) + kernel + all-gatherv [%s]; p2p = scatter + all-gather broadcast and "
                            "all-gatherv as grouped point-to-point batches (all xGMI links of a GPU at once); pipelined = column "
                            "panels, broadcast(p+2) | kernel(p) | all-gatherv(p-1) overlapped, B and C HELD panel-major (layout "
                            "conversion not timed: never sets `value`).  `value` is the better of basic / p2p." % args.gather_mode}
                line["collectives"].update({k: round(tv[k] * 1e3, 3) for k in ("t_bcast_scatter_allgather", "t_gather_p2p") if k in tv})
                line["value"] = line["variants"]["value_GFLOPs"][best]
                line["ms_per_step"] = line["end_to_end_ms"] = line["variants"]["end_to_end_ms"][best]
        except AssertionError:
            raise
        except Exception as exc:  # noqa: BLE001 -- an experimental form must not cost the contract line
            if rank == 0:
                line.setdefault("variants", {})["error"] = "%s: %s" % (type(exc).__name__, str(exc)[:300])
        dog.cancel()

    # ---- release the primary workload before the secondaries (they need most of the HBM) ----
    res["free"]()
    del res, C
    handles.clear()
    cfg5_dry = os.environ.get("BENCH_CFG5_SCALE")  # dry run of the 8-rank control flow on a small matrix (gloo, ranks sharing a GPU)
    if dist and world == 8 and (backend == "nccl" or cfg5_dry) and not args.no_variants and not args.no_secondary and args.workload == "rmat":
        import threading

        def _bail5():
            if rank == 0:
                line.setdefault("secondary", {})["spmm_config5_8gpu"] = {"error": "timed out after %d s" % args.config5_timeout}
                print(json.dumps(line), flush=True)
            os._exit(0)
        dog = threading.Timer(args.config5_timeout, _bail5)
        dog.daemon = True
        dog.start()
        try:
            del B, indptr, indices, vals
            torch.cuda.empty_cache()
            entry = secondary_config5(torch, dist, dev, abi, rank, world, allreduce_max)
            if rank == 0:
                line.setdefault("secondary", {})["spmm_config5_8gpu"] = entry
        except AssertionError:
            raise
        except Exception as exc:  # noqa: BLE001
            if rank == 0:
                line.setdefault("secondary", {})["spmm_config5_8gpu"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}
        dog.cancel()
        if rank == 0:
            print(json.dumps(line), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return
    want_sec = set(args.secondary.split(",")) if args.secondary != "all" else {"uniform", "spgemm", "spgemm_rmat", "gram", "host_api", "gemm", "spmv", "rows"}
    with_cpu = not args.no_cpu
    if rank == 0 and world == 1 and not args.no_secondary and args.workload == "rmat":
        secondary = {}
        if "host_api" in want_sec:  # before the CPU baselines: MKL's 128 spinning OpenMP threads would fight the stager's
            try:
                secondary["host_api"] = host_api_figure(sda)
            except Exception as exc:  # noqa: BLE001
                secondary["host_api"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}
        if "uniform" in want_sec:
            secondary["spmm_uniform"] = secondary_uniform_spmm(torch, abi, dev, n, N, B, args.steps, args.warmup)
        if "spmv" in want_sec:
            try:
                secondary.update(secondary_spmv_gram_sparse(torch, abi, dev, indptr, indices, vals, n, args.steps, with_cpu))
            except AssertionError:
                raise
            except Exception as exc:  # noqa: BLE001
                secondary["spmv"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}
        if "rows" in want_sec:
            secondary.update(secondary_rows_f3_f4_a4(torch, abi, dev, with_cpu))
        if not args.no_pmc:  # counter traffic of the three small secondaries (children under rocprofv3, two --pmc passes each)
            for key, child, inc in (("spmv", ["spmv", "--reps", "3"], ("k_spmv", "k_spmm_fixup")),
                                    ("gram_sparse", ["syrk", "--reps", "2"], ("mi::",)),
                                    ("spmmd", ["spmmd", "--reps", "2"], ("k_spmmd", "k_fill_dense"))):
                if isinstance(secondary.get(key), dict) and "error" not in secondary[key]:
                    exc = () if key in ("spmv", "spmmd") else ("k_spmv", "k_spmm", "k_check_", "k_widen_ptr", "k_count_descents", "k_count_start_descents")
                    attach_traffic(secondary[key], child, int(child[-1]) + 1, inc, True, exclude=exc, need_free_bytes=8 << 30)
        if "gemm" in want_sec:
            try:
                secondary["gemm_dense"] = secondary_gemm(torch, abi, dev, with_cpu)  # (before the SpMM's CPU baselines: see its docstring)
            except AssertionError:
                raise
            except Exception as exc:  # noqa: BLE001
                secondary["gemm_dense"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}
        if with_cpu:
            line["cpu_baseline"] = cpu_baseline_spmm(indptr.cpu().numpy(), indices.cpu().numpy(), vals.cpu().numpy(), n, B.cpu().numpy())
        else:
            line["cpu_baseline"] = None
        del B, indptr, indices, vals
        torch.cuda.empty_cache()
        for key, fn in (("spgemm_uniform", lambda: secondary_spgemm(torch, abi, dev, "uniform", with_cpu)),
                        ("spgemm_rmat_literal", lambda: secondary_spgemm(torch, abi, dev, "rmat", with_cpu)),
                        ("gram_dense", lambda: secondary_gram(torch, abi, dev, with_cpu))):
            tag = {"spgemm_uniform": "spgemm", "spgemm_rmat_literal": "spgemm_rmat", "gram_dense": "gram"}[key]
            if tag not in want_sec:
                continue
            try:
                secondary[key] = fn()
            except AssertionError:
                raise
            except Exception as exc:  # noqa: BLE001 -- a secondary must not cost the headline line
                secondary[key] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}
            torch.cuda.synchronize()
            sda.mi_set_option("pool_trim", 1)
            torch.cuda.empty_cache()
            if isinstance(secondary[key], dict) and "error" not in secondary[key]:
                # counter traffic of THIS secondary, collected now in a child under rocprofv3 (two --pmc passes)
                child = {"spgemm_uniform": (["spgemm", "--no-order", "--reps", "2"], 3),
                         "spgemm_rmat_literal": (["spgemm", "--kind", "rmat", "--scale", "20", "--per-row", "16", "--no-order", "--reps", "1"], 2),
                         "gram_dense": (["gram", "--dense", "--cols", "262144", "--rows-log2", "22", "--reps", "1"], 2)}[key]
                inc = ("k_syrkd",) if key == "gram_dense" else ("mi::",)
                exc = ("k_spmv", "k_spmm", "k_check_", "k_widen_ptr", "k_count_descents", "k_count_start_descents")
                need = {"spgemm_uniform": 16 << 30, "spgemm_rmat_literal": 150 << 30, "gram_dense": 272 << 30}[key]
                attach_traffic(secondary[key], child[0], child[1], inc, not args.no_pmc, exclude=exc, need_free_bytes=need)
        line["secondary"] = secondary
    elif rank == 0:
        line["cpu_baseline"] = None
        if world == 1 and with_cpu:
            line["cpu_baseline"] = cpu_baseline_spmm(indptr.cpu().numpy(), indices.cpu().numpy(), vals.cpu().numpy(), n, B.cpu().numpy())
    if rank == 0:
        if world > 1:  # the three readings of an N-GPU run, labelled, for whoever reads the log (stderr: the contract line stays alone on stdout)
            for label, key, what in (("value", "value", "end to end: bcast(B) -> local kernels -> all-gatherv(C), what the contract times"),
                                     ("resident_B_value", "resident_B_value", "B already on every rank: local kernels -> all-gatherv(C)"),
                                     ("compute_only_value", "compute_only_value", "the local kernels alone, max over ranks")):
                print("[bench --gpus %d] %-18s %10.2f GFLOP/s   (%s)" % (world, label, line.get(key) or float("nan"), what), file=sys.stderr, flush=True)
        print(json.dumps(line), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
