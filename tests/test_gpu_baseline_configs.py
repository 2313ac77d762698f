"""GPU: every BASELINE.json config that fits one MI355X, at its STATED workload, through the C ABI with device
pointers.  The oracle would need hours at these sizes, so parity is checked through size-independent
properties plus row samples that ARE evaluated exactly on the host (numpy Gustavson / fp64 dot products):

  configs[2]  two R-MAT 2^20 x 2^20, 16 edges/row, fp64, SpGEMM -> nnz(C) = 9.7e9 (> INT32_MAX, 116 GB)
  configs[3]  A^T A of a uniform 4 M x 262144 CSR, 64/row, fp32, dense output (256 GiB) -- tried literally; if the
              output cannot be allocated the reason is written to gpurun_out/ and the widest power of two that fits runs
  configs[4]  the 16 M x 16 M x 256 product's shape on ONE GPU (A 4 GB + B 17 GB + C 17 GB resident); its 8-GPU
              row partition is covered by tests/test_distributed_cpu.py (gloo) and bench.py --gpus N

configs[1] at scale lives in test_gpu_parity.py::test_config2_scale_properties, configs[0] in ::test_config1_*.
"""
import ctypes as ct
import os
import sys

import numpy as np
import pytest
import scipy.sparse as sps

pytestmark = pytest.mark.gpu

F32_TOL = 1e-5   # north_star: fp32 within 1e-5 rel
F64_TOL = 1e-12  # north_star: fp64 within 1e-12 rel

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def _hip():
    from sparse_dot_amd._mi_interface import _library
    lib = _library._torch_hip or ct.CDLL("libamdhip64.so")
    lib.hipMemcpy.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_size_t, ct.c_int]
    lib.hipMemcpy.restype = ct.c_int
    return lib


def _d2h(dev_ptr, count, dtype, offset_elems=0):
    """Copy `count` elements of `dtype` starting at element `offset_elems` of a raw device pointer."""
    out = np.empty(count, dtype=dtype)
    if count:
        rc = _hip().hipMemcpy(out.ctypes.data, ct.c_void_p(dev_ptr + offset_elems * out.itemsize), out.nbytes, 2)
        assert rc == 0, "hipMemcpy D2H failed: %d" % rc
    return out


def _abi():
    from sparse_dot_amd._mi_interface import MI, matrix_descr, sparse_matrix_t, _check_return_value
    return MI, matrix_descr, sparse_matrix_t, _check_return_value


def test_config3_literal_rmat_spgemm(gpu):
    """BASELINE configs[2] as literally stated: two R-MAT scale-20 matrices, 16 edges/row (seeds 21, 23), fp64.
    (1) nnz(C) equals the count obtained in round 1 for these seeds and exceeds INT32_MAX;
    (2) C 1 = A (B 1) to 1e-12;  (3) >= 64 sampled rows -- among them the 5 longest a host Gustavson can afford -- have
    bit-exact structure (sorted column sets) and 1e-12 values against a host Gustavson of just those rows;  (4) the 5
    rows with the most products and the 3 longest rows of C -- whatever their size -- the same against a chunked fp64 Gustavson on the device (torch ops only)."""
    torch = pytest.importorskip("torch")
    import bench
    MI, matrix_descr, sparse_matrix_t, check = _abi()
    dev = torch.device("cuda", 0)
    n = 1 << 20
    a = bench.rmat_csr(torch, 20, 16, 21, dev)
    b = bench.rmat_csr(torch, 20, 16, 23, dev)
    av, bv = a[2].double(), b[2].double()
    gpu.mi_set_stream(torch.cuda.current_stream().cuda_stream)
    handles = []
    try:
        def mk(ip, idx, v):
            h = sparse_matrix_t()
            check(MI.call("mi_sparse_d_create_csr", ct.byref(h), 0, n, n, ip.data_ptr(), ip.data_ptr() + 4, idx.data_ptr(),
                          v.data_ptr()), "create")
            handles.append(h)
            return h
        ha, hb = mk(a[0], a[1], av), mk(b[0], b[1], bv)
        hc = sparse_matrix_t()
        check(MI.call("mi_sparse_spmm", 10, ha, hb, ct.byref(hc)), "spmm")
        handles.append(hc)
        torch.cuda.synchronize()
        rows, cols, nnz = ct.c_int64(), ct.c_int64(), ct.c_int64()
        check(MI.call("mi_sparse_get_info", hc, ct.byref(rows), ct.byref(cols), ct.byref(nnz), None, None), "info")
        assert (rows.value, cols.value) == (n, n)
        assert nnz.value > np.iinfo(np.int32).max
        assert nnz.value == 9_714_729_534, nnz.value  # same generator, same seeds as round 1's measurement

        # (2) row sums through the library's own SpMV (a different kernel family)
        def mv(h, x, y):
            check(MI.call("mi_sparse_d_mv", 10, 1.0, h, matrix_descr(), x.data_ptr(), 0.0, y.data_ptr()), "mv")
        ones = torch.ones(n, device=dev, dtype=torch.float64)
        b1, ab1, c1 = (torch.empty(n, device=dev, dtype=torch.float64) for _ in range(3))
        mv(hb, ones, b1)
        mv(ha, b1, ab1)
        mv(hc, ones, c1)
        torch.cuda.synchronize()
        assert float(((c1 - ab1).abs() / ab1.abs().clamp(min=1e-300)).max()) <= F64_TOL

        # (3) sampled rows against a host Gustavson
        p_ptr, p_col, p_val = ct.c_void_p(), ct.c_void_p(), ct.c_void_p()
        check(MI.call("mi_sparse_get_device_csr", hc, ct.byref(p_ptr), ct.byref(p_col), ct.byref(p_val)), "devcsr")
        cptr = _d2h(p_ptr.value, n + 1, np.int64)
        assert cptr[0] == 0 and cptr[-1] == nnz.value and np.all(np.diff(cptr) >= 0)
        clen = np.diff(cptr)
        a_ptr, a_idx, a_val = a[0].cpu().numpy().astype(np.int64), a[1].cpu().numpy(), av.cpu().numpy()
        b_ptr, b_idx, b_val = b[0].cpu().numpy().astype(np.int64), b[1].cpu().numpy(), bv.cpu().numpy()
        # the five longest rows of C whose host Gustavson stays below 2e8 products each, searched among the 400 longest (the
        # very longest rows have > 1e9: tens of GB of numpy temporaries and minutes on a busy host -- those are checked on
        # the device in (4))
        b_len = np.diff(b_ptr)
        longest = []
        for r in np.argsort(clen)[::-1][:400].tolist():
            if int(b_len[a_idx[a_ptr[r]:a_ptr[r + 1]]].sum()) <= 200_000_000:
                longest.append(r)
            if len(longest) == 5:
                break
        assert longest and clen[longest[0]] > 100_000
        rng = np.random.default_rng(0)
        sample = np.unique(np.concatenate([np.array(longest, dtype=np.int64), rng.integers(0, n, 70)]))
        assert len(sample) >= 64
        for r in sample.tolist():
            ks = a_idx[a_ptr[r]:a_ptr[r + 1]]
            avs = a_val[a_ptr[r]:a_ptr[r + 1]]
            lens = b_ptr[ks + 1] - b_ptr[ks]
            tot = int(lens.sum())
            # flat list of the row's products: (column of B, a * b)
            starts = np.repeat(b_ptr[ks], lens)
            within = np.arange(tot, dtype=np.int64) - np.repeat(np.cumsum(lens) - lens, lens)
            pos = starts + within
            acc = np.bincount(b_idx[pos], weights=np.repeat(avs, lens) * b_val[pos], minlength=n)
            hit = np.zeros(n, dtype=bool)
            hit[b_idx[pos]] = True
            want_cols = np.flatnonzero(hit)
            got_cols = _d2h(p_col.value, int(clen[r]), np.int32, int(cptr[r]))
            got_vals = _d2h(p_val.value, int(clen[r]), np.float64, int(cptr[r]))
            order = np.argsort(got_cols, kind="stable")
            assert np.array_equal(got_cols[order], want_cols), "row %d: structure differs" % r
            want_vals = acc[want_cols]
            err = np.max(np.abs(got_vals[order] - want_vals) / np.maximum(np.abs(want_vals), 1e-300)) if tot else 0.0
            assert err <= F64_TOL, (r, err)
        # (4) the five rows with the MOST PRODUCTS and the three longest rows of C, without any size filter (the hub rows that
        # take the deepest path of the big-row kernels) against a Gustavson evaluation on the device in fp64 with torch ops only (index_add_ into a dense
        # accumulator, a chunk of <= 1e8 products at a time) -- exact structure, 1e-12 values
        ip_a, ip_b = a[0].to(torch.int64), b[0].to(torch.int64)
        idx_b = b[1].to(torch.int64)
        prod_cum = np.concatenate([[0], np.cumsum(b_len[a_idx])])
        prods = prod_cum[a_ptr[1:]] - prod_cum[a_ptr[:-1]]  # products per row of C
        top = list(dict.fromkeys(np.argsort(prods)[::-1][:5].tolist() + np.argsort(clen)[::-1][:3].tolist()))
        assert prods[top[0]] == prods.max() and int(np.argmax(clen)) in top
        for r in top:
            lo, hi = int(a_ptr[r]), int(a_ptr[r + 1])
            acc = torch.zeros(n, device=dev, dtype=torch.float64)
            hit = torch.zeros(n, device=dev, dtype=torch.bool)
            ks_all, av_all = a[1][lo:hi].to(torch.int64), av[lo:hi]
            lens_all = ip_b[ks_all + 1] - ip_b[ks_all]
            bounds = [0]
            run = 0
            for i, l in enumerate(lens_all.tolist()):
                run += l
                if run >= 100_000_000:
                    bounds.append(i + 1)
                    run = 0
            if bounds[-1] != hi - lo:
                bounds.append(hi - lo)
            for c0, c1 in zip(bounds[:-1], bounds[1:]):
                ks, lens = ks_all[c0:c1], lens_all[c0:c1]
                tot = int(lens.sum())
                if tot == 0:
                    continue
                first = torch.cumsum(lens, 0) - lens
                pos = torch.repeat_interleave(ip_b[ks] - first, lens) + torch.arange(tot, device=dev)
                cols_p = idx_b[pos]
                acc.index_add_(0, cols_p, torch.repeat_interleave(av_all[c0:c1], lens) * bv[pos])
                hit[cols_p] = True
                del pos, cols_p
            want_cols = torch.nonzero(hit).flatten()
            got_cols = torch.from_numpy(_d2h(p_col.value, int(clen[r]), np.int32, int(cptr[r])).astype(np.int64)).to(dev)
            got_vals = torch.from_numpy(_d2h(p_val.value, int(clen[r]), np.float64, int(cptr[r]))).to(dev)
            order = torch.argsort(got_cols)
            assert got_cols.numel() == want_cols.numel() and torch.equal(got_cols[order], want_cols), "hub row %d: structure differs" % r
            want_vals = acc[want_cols]
            err = float(((got_vals[order] - want_vals).abs() / want_vals.abs().clamp(min=1e-300)).max())
            assert err <= F64_TOL, (r, err)
    finally:
        for h in handles:
            MI.call("mi_sparse_destroy", h)
        gpu.mi_set_stream(0)
        gpu.mi_set_option("pool_trim", 1)
        torch.cuda.empty_cache()


def test_config4_literal_gram_dense(gpu):
    """BASELINE configs[3]: A^T A of a uniform CSR 2^22 x 262144, 64/row, fp32, dense=True -> 256 GiB output array.
    Tried literally; on an allocation failure the error text goes to gpurun_out/cfg4_alloc_error.txt and the next
    narrower power of two runs instead (the id of the shape that ran is in the assertion messages).
    Checks: diag(C) = column sums of A.^2; sampled entries = fp64 column dot products; sampled blocks of the strict
    lower triangle untouched."""
    torch = pytest.importorskip("torch")
    import bench
    MI, matrix_descr, sparse_matrix_t, check = _abi()
    dev = torch.device("cuda", 0)
    m = 1 << 22
    gpu.mi_set_option("pool_trim", 1)
    torch.cuda.empty_cache()
    C, ncols, errors = None, None, []
    for cand in (262144, 131072, 65536):
        try:
            C = torch.full((cand, cand), -7.0, device=dev, dtype=torch.float32)
            ncols = cand
            break
        except Exception as e:  # noqa: BLE001
            errors.append("n=%d: %s: %s" % (cand, type(e).__name__, e))
            torch.cuda.empty_cache()
    if errors:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "cfg4_alloc_error.txt"), "w") as f:
            f.write("\n".join(errors) + "\n")
    assert C is not None, errors
    ip, idx, val, _ = bench.uniform_csr(torch, m, 64, 3, dev, ncols=ncols)
    gpu.mi_set_stream(torch.cuda.current_stream().cuda_stream)
    h = sparse_matrix_t()
    checks = []
    try:
        check(MI.call("mi_sparse_s_create_csr", ct.byref(h), 0, m, ncols, ip.data_ptr(), ip.data_ptr() + 4, idx.data_ptr(),
                      val.data_ptr()), "create")
        check(MI.call("mi_sparse_s_syrkd", 11, h, 1.0, 0.0, C.data_ptr(), 101, ncols), "syrkd")
        torch.cuda.synchronize()
        # every check is collected first and asserted AFTER the 256 GiB array has been released: a failing assert
        # inside this block would keep C alive through the traceback and starve every later test of memory
        colsq = torch.zeros(ncols, device=dev, dtype=torch.float64)
        colsq.index_add_(0, idx.long(), val.double() ** 2)
        derr = float(((torch.diagonal(C).double() - colsq).abs() / colsq.clamp(min=1e-30)).max())
        checks.append(("diag(C) = column sums of A.^2 (n=%d): %g" % (ncols, derr), derr <= F32_TOL))
        # strict lower triangle never touched: blocks on the diagonal and a block far below it
        for r0 in (0, ncols // 2, ncols - 2048):
            low = torch.tril(torch.ones(2048, 2048, dtype=torch.bool, device=dev), -1)
            ok = bool((C[r0:r0 + 2048, r0:r0 + 2048][low] == -7.0).all())
            checks.append(("strict lower triangle untouched at %d (n=%d)" % (r0, ncols), ok))
            del low
        checks.append(("block below the diagonal untouched", bool((C[ncols - 1024:, :1024] == -7.0).all())))
        # sampled entries against fp64 dot products of the two columns (host, from the CSC of A)
        a_host = sps.csr_matrix((val.cpu().numpy().astype(np.float64), idx.cpu().numpy(), ip.cpu().numpy()),
                                shape=(m, ncols)).tocsc()
        rng = np.random.default_rng(0)
        pairs = [tuple(sorted(rng.integers(0, ncols, 2).tolist())) for _ in range(48)]
        # make sure some sampled entries are structurally non-zero: two columns of one row of A
        host_ip, host_idx = ip.cpu().numpy(), idx.cpu().numpy()
        for r in rng.integers(0, m, 16).tolist():
            cs = host_idx[host_ip[r]:host_ip[r + 1]]
            if len(cs) >= 2:
                pairs.append((int(cs[0]), int(cs[-1])))
        nonzero = 0
        for i, j in pairs:
            want = float(a_host[:, [i]].multiply(a_host[:, [j]]).sum())
            got = float(C[i, j])
            nonzero += want != 0
            checks.append(("C[%d, %d] = %r, want %r (n=%d)" % (i, j, got, want, ncols), abs(got - want) <= F32_TOL * abs(want)))
        checks.append(("at least 8 structurally non-zero samples", nonzero >= 8))
    finally:
        if h:
            MI.call("mi_sparse_destroy", h)
        gpu.mi_set_stream(0)
        C = None
        gpu.mi_set_option("pool_trim", 1)
        torch.cuda.empty_cache()
    bad = [name for name, ok in checks if not ok]
    assert checks and not bad, bad


def test_config5_shape_single_gpu(gpu):
    """BASELINE configs[4]'s operands on ONE GPU: R-MAT scale 24, 32 edges/row (~5.2e8 nnz) x dense 2^24 x 256 fp32
    (A 4 GB + B 17 GB + C 17 GB resident; 34-bit byte offsets into B and C).  Checks: A @ ones = row sums of A,
    empty rows are zero, and a row sample (the longest row included) equals an fp64 evaluation."""
    torch = pytest.importorskip("torch")
    import bench
    MI, matrix_descr, sparse_matrix_t, check = _abi()
    dev = torch.device("cuda", 0)
    gpu.mi_set_option("pool_trim", 1)
    torch.cuda.empty_cache()
    indptr, indices, vals, n = bench.rmat_csr(torch, 24, 32, 7, dev)
    nnz = indices.numel()
    assert 4.9e8 < nnz < 5.4e8, nnz
    N = 256
    gpu.mi_set_stream(torch.cuda.current_stream().cuda_stream)
    h = sparse_matrix_t()
    try:
        check(MI.call("mi_sparse_s_create_csr", ct.byref(h), 0, n, n, indptr.data_ptr(), indptr.data_ptr() + 4,
                      indices.data_ptr(), vals.data_ptr()), "create")

        def mm(b, c):
            check(MI.call("mi_sparse_s_mm", 10, 1.0, h, matrix_descr(), 101, b.data_ptr(), N, N, 0.0, c.data_ptr(), N), "mm")
        ip = indptr.to(torch.int64)
        lens = ip[1:] - ip[:-1]
        B = torch.ones((n, N), device=dev)
        C = torch.empty((n, N), device=dev)
        for _ in range(3):  # the third call runs with the hot / cold tags adopted (same results required)
            mm(B, C)
            torch.cuda.synchronize()
        # row sums of A as differences of an fp64 running sum (an index_add_ of 5e8 sorted fp64 atomics takes minutes)
        cs = torch.cat([torch.zeros(1, device=dev, dtype=torch.float64), torch.cumsum(vals.double(), 0)])
        rowsum = cs[ip[1:]] - cs[ip[:-1]]
        del cs
        err = ((C[:, 0].double() - rowsum).abs() / rowsum.clamp(min=1e-30))[lens > 0].max().item()
        assert err <= F32_TOL, err
        assert (C[lens == 0] == 0).all()
        assert torch.equal(C[:, :1].expand(-1, N), C)
        g = torch.Generator(device=dev)
        g.manual_seed(1)
        B.copy_(torch.rand((n, N), generator=g, device=dev) + 0.5)
        mm(B, C)
        torch.cuda.synchronize()
        sel = torch.cat([torch.argmax(lens).reshape(1), torch.tensor([0, n - 1], device=dev), torch.randint(0, n, (40,), device=dev)])
        for r in sel.tolist():
            lo, hi = int(ip[r]), int(ip[r + 1])
            if hi == lo:
                assert float(C[r].abs().max()) == 0.0
                continue
            want = (vals[lo:hi].double()[:, None] * B[indices[lo:hi].long()].double()).sum(0)
            assert ((C[r].double() - want).abs() / want.abs()).max().item() <= F32_TOL, r
    finally:
        if h:
            MI.call("mi_sparse_destroy", h)
        gpu.mi_set_stream(0)
        gpu.mi_set_option("pool_trim", 1)
        torch.cuda.empty_cache()


def _hub_csr(torch, dev, rows, cols, per_row, hubs, seed):
    """Skewed CSR on the device: half of every row's entries fall on `hubs` hub columns spread over the WHOLE column
    range (so that hot and cold rows of the dense operand both lie beyond any 2 GiB / 4 GiB boundary), the rest are
    uniform.  De-duplicated, sorted; int32 indptr / indices, fp32 values U[0.5, 1.5)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    hub_cols = (torch.arange(hubs, device=dev, dtype=torch.int64) * (cols // hubs) + (cols // hubs) // 2)
    r = torch.arange(rows, device=dev, dtype=torch.int64).repeat_interleave(per_row)
    pick = torch.randint(0, hubs, (rows * per_row,), generator=g, device=dev)
    uni = torch.randint(0, cols, (rows * per_row,), generator=g, device=dev)
    is_hub = torch.rand(rows * per_row, generator=g, device=dev) < 0.5
    c = torch.where(is_hub, hub_cols[pick], uni)
    key = torch.unique(r * cols + c)
    rr = key // cols
    indices = (key % cols).to(torch.int32)
    indptr = torch.zeros(rows + 1, dtype=torch.int64, device=dev)
    indptr[1:] = torch.cumsum(torch.bincount(rr, minlength=rows), 0)
    vals = torch.rand(indices.numel(), generator=g, device=dev, dtype=torch.float32) + 0.5
    return indptr.to(torch.int32), indices, vals


@pytest.mark.parametrize("k_cols,n_dense,want_mode", [
    (1 << 22, 128, 1),        # B = exactly 2 GiB: raw buffer loads, 32-bit offsets up to 2^31
    (3 << 20, 256, 1),        # B = 3 GiB: offsets in [2^31, 2^32) -- zeros came back here with num_records = 0x7fffffff
    (5 << 20, 256, 0),        # B = 5 GiB: beyond 32-bit offsets -> the untagged gather (and correct)
])
def test_spmm_tagged_gather_large_dense_operand(gpu, k_cols, n_dense, want_mode):
    """The hot / cold TAGGED gather (third call onward on a skewed matrix) addresses B through a buffer resource whose
    range check returns ZERO for out-of-range offsets.  Dense operands of 2-4 GiB (32-bit offsets beyond 2^31) and
    beyond 4 GiB (where the gather stays untagged) must give, bit for bit, the untagged result, and a row sample that references
    columns in the upper half of B must equal an fp64 evaluation."""
    torch = pytest.importorskip("torch")
    MI, matrix_descr, sparse_matrix_t, check = _abi()
    dev = torch.device("cuda", 0)
    gpu.mi_set_option("pool_trim", 1)
    torch.cuda.empty_cache()
    rows = 1 << 18
    indptr, indices, vals = _hub_csr(torch, dev, rows, k_cols, 36, 4096, 5)
    nnz = indices.numel()
    assert nnz >= 1 << 23  # large enough for the sampled analysis to run
    gpu.mi_set_stream(torch.cuda.current_stream().cuda_stream)
    h = sparse_matrix_t()
    try:
        check(MI.call("mi_sparse_s_create_csr", ct.byref(h), 0, rows, k_cols, indptr.data_ptr(), indptr.data_ptr() + 4,
                      indices.data_ptr(), vals.data_ptr()), "create")
        g = torch.Generator(device=dev)
        g.manual_seed(2)
        B = torch.rand((k_cols, n_dense), generator=g, device=dev) + 0.5
        assert B.numel() * 4 == k_cols * n_dense * 4 >= 1 << 31
        C = torch.empty((rows, n_dense), device=dev)

        def mm(c):
            check(MI.call("mi_sparse_s_mm", 10, 1.0, h, matrix_descr(), 101, B.data_ptr(), n_dense, n_dense, 0.0,
                          c.data_ptr(), n_dense), "mm")
        for _ in range(4):  # tags are analysed behind call 2 and adopted by the next call that finds them landed
            mm(C)
            torch.cuda.synchronize()
        assert gpu.mi_get_counter("spmm_last_tagged") == float(want_mode)
        assert 0.3 < gpu.mi_get_counter("spmm_hot_coverage") < 0.8  # the analysis itself ran and found the hub columns
        # (1) bit-identical to the untagged gather
        C0 = torch.empty_like(C)
        gpu.mi_set_option("spmm_hot_kb", 0)
        try:
            mm(C0)
            torch.cuda.synchronize()
            assert gpu.mi_get_counter("spmm_last_tagged") == 0.0
        finally:
            gpu.mi_set_option("spmm_hot_kb", 8192)
        assert torch.equal(C, C0)
        # (2) sampled rows against fp64, every one of them referencing columns in the upper half of B
        ip = indptr.to(torch.int64)
        sel = torch.randint(0, rows, (48,), device=dev).tolist() + [0, rows - 1]
        upper = 0
        for r in sel:
            lo, hi = int(ip[r]), int(ip[r + 1])
            if hi == lo:
                assert float(C[r].abs().max()) == 0.0
                continue
            cols = indices[lo:hi].long()
            upper += int((cols >= k_cols // 2).sum())
            want = (vals[lo:hi].double()[:, None] * B[cols].double()).sum(0)
            assert ((C[r].double() - want).abs() / want.abs()).max().item() <= F32_TOL, r
        assert upper >= 100
        # (3) no silent zeros anywhere: every non-empty row of a positive product is positive in every column
        lens = ip[1:] - ip[:-1]
        assert bool((C[lens > 0] > 0).all())
    finally:
        if h:
            MI.call("mi_sparse_destroy", h)
        gpu.mi_set_stream(0)
        gpu.mi_set_option("pool_trim", 1)
        B = C = C0 = None
        torch.cuda.empty_cache()


@pytest.mark.parametrize("where", ["nowhere", "first_rows", "late_row_only"])
def test_order_of_a_large_matrix_looks_at_its_first_rows_first(where):
    """mi_sparse_order (reference _common.py:683 mkl_sparse_order) on 6.7e7 entries aliased from caller HBM: rows_sorted() decides
    from the first rows / 256 rows when they already show a descent (round 6: a SpGEMM result about to be ordered), and must
    still find a single unsorted row far behind that prefix.  Rows in order stay as they are; every value stays with its column."""
    import torch
    MI, matrix_descr, sparse_matrix_t, check = _abi()
    import sparse_dot_amd as sda
    dev = torch.device("cuda", 0)
    sda.mi_set_stream(torch.cuda.current_stream().cuda_stream)
    rows, per = 1 << 20, 64
    ncols = 1 << 20
    j = torch.arange(per, device=dev, dtype=torch.int64)
    r = torch.arange(rows, device=dev, dtype=torch.int64)
    col = (j[None, :] * 16000 + (r[:, None] * 7) % 16000).to(torch.int32)        # ascending inside every row, < 2^20
    assert int(col.max()) < ncols
    bad = []
    if where == "first_rows":
        bad = [3, 100, 4000]
    elif where == "late_row_only":
        bad = [rows - 5]            # far beyond rows / 256
    for b in bad:
        col[b] = torch.flip(col[b], dims=[0])
    col = col.reshape(-1).contiguous()
    val = col.to(torch.float64) * 0.5 + 1.0                                          # value = f(column): pairing is checkable
    ip = (torch.arange(rows + 1, device=dev, dtype=torch.int64) * per).to(torch.int32)
    assert col.numel() >= 1 << 26
    h = sparse_matrix_t()
    check(MI.call("mi_sparse_d_create_csr", ct.byref(h), 0, rows, ncols, ip.data_ptr(), ip.data_ptr() + 4, col.data_ptr(),
                  val.data_ptr()), "create")
    try:
        check(MI.call("mi_sparse_order", h), "order")
        torch.cuda.synchronize()
        c2 = col.reshape(rows, per)
        assert bool((c2[:, 1:] > c2[:, :-1]).all())                                  # the caller's arrays ARE the matrix: sorted in place
        assert bool(torch.equal(val, col.to(torch.float64) * 0.5 + 1.0))
    finally:
        MI.call("mi_sparse_destroy", h)
