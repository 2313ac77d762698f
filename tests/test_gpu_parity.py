"""GPU: parity of the HIP path against the CPU oracle on seeded inputs that stress the kernels'
code paths (lane-group widths, chunk boundaries, hub rows, empty rows, hash-table bins, the
global-memory hash fallback, 64-bit indices, device pointers), plus size-independent properties
at the BASELINE config-2 scale."""
import ctypes as ct

import numpy as np
import pytest
import scipy.sparse as sps

pytestmark = pytest.mark.gpu

F32_TOL = 1e-5   # north_star: fp32 within 1e-5 rel
F64_TOL = 1e-12  # north_star: fp64 within 1e-12 rel


def rel_err(got, want):
    """max |got - want| / (|A| |B|-style scale): elementwise relative to the magnitude of the exact
    result computed in float64 with positive data (no cancellation), floor 1e-30."""
    want = np.asarray(want, dtype=np.complex128 if np.iscomplexobj(want) else np.float64)
    return float(np.max(np.abs(got - want) / np.maximum(np.abs(want), 1e-30))) if want.size else 0.0


def pos_csr(m, n, density, dtype, seed, fmt="csr"):
    a = sps.random(m, n, density=density, format="csr", dtype=np.float64, random_state=seed)
    rng = np.random.default_rng(seed + 1000)
    a.data[:] = rng.uniform(0.5, 1.5, a.nnz)
    if np.dtype(dtype).kind == "c":
        a = a.astype(dtype)
        a.data += 1j * rng.uniform(0.5, 1.5, a.nnz)
    return a.astype(dtype).asformat(fmt)


def dense(shape, dtype, seed, order="C"):
    rng = np.random.default_rng(seed)
    x = rng.uniform(0.5, 1.5, shape)
    if np.dtype(dtype).kind == "c":
        x = x + 1j * rng.uniform(0.5, 1.5, shape)
    return np.asarray(x.astype(dtype), order=order)


def tol(dtype):
    return F32_TOL if np.dtype(dtype) in (np.dtype(np.float32), np.dtype(np.complex64)) else F64_TOL


# ---- SpMM -----------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.complex64, np.complex128])
@pytest.mark.parametrize("n", [1, 3, 4, 8, 17, 32, 64, 96, 128, 200, 256, 260, 512, 1000])
def test_spmm_widths(gpu, oracle, dtype, n):
    """Every lane-group width of the vector path and the scalar path (N not a multiple of 16 B)."""
    a = pos_csr(700, 500, 0.03, dtype, 1)
    b = dense((500, n), dtype, 2)
    got = gpu.dot_product_mkl(a, b) if n > 1 else gpu.dot_product_mkl(a, b.reshape(500, 1).copy())
    want = oracle.spmm(a.astype(np.complex128 if np.dtype(dtype).kind == "c" else np.float64),
                       b.astype(np.complex128 if np.dtype(dtype).kind == "c" else np.float64))
    assert got.dtype == dtype and got.shape == (700, n)
    assert rel_err(got, want) <= tol(dtype)


@pytest.mark.parametrize("chunk", [128, 256, 512, 1024])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_spmm_skewed_rows_and_chunks(gpu, oracle, dtype, chunk):
    """Hub rows spanning many chunks, runs of empty rows (also leading / trailing), 1-nnz rows."""
    rng = np.random.default_rng(7)
    lens = rng.integers(0, 40, 3000)
    lens[:50] = 0
    lens[-70:] = 0
    lens[100] = 5000   # spans ~20 chunks of 256
    lens[101] = 1
    lens[1500:1600] = 0
    lens[2000] = 1300
    ncols = 6000
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    indices = np.concatenate([np.sort(rng.choice(ncols, l, replace=False)) for l in lens]).astype(np.int32)
    data = rng.uniform(0.5, 1.5, indices.size).astype(dtype)
    a = sps.csr_matrix((data, indices, indptr), shape=(3000, ncols))
    b = dense((ncols, 128 if dtype == np.float32 else 64), dtype, 3)
    gpu.mi_set_option("spmm_chunk", chunk)
    try:
        got = gpu.dot_product_mkl(a, b)
        out = np.full_like(got, 2.0)
        got2 = gpu.dot_product_mkl(a, b, out=out, out_scalar=-1.5)
    finally:
        gpu.mi_set_option("spmm_chunk", 256)
    want = oracle.spmm(a.astype(np.float64), b.astype(np.float64))
    assert rel_err(got, want) <= tol(dtype)
    assert got2 is out
    np.testing.assert_allclose(got2, want - 3.0, rtol=10 * tol(dtype), atol=10 * tol(dtype))
    # empty rows are written (zeros), not skipped
    assert not got[:50].any() and not got[-70:].any()


@pytest.mark.parametrize("dtype,n", [(np.float32, 128), (np.float32, 256), (np.float64, 64), (np.float64, 128),
                                     (np.complex64, 64), (np.complex128, 32)])
def test_spmm_hot_cold_tagged_gather(gpu, oracle, dtype, n):
    """The hot / cold tagged gather (buffer loads, nt policy on cold columns) on a power-law column
    distribution; forced on for a test-sized matrix.  Same values as the untagged path, bit for bit."""
    rng = np.random.default_rng(17)
    m, k = 3000, 4000
    lens = rng.integers(0, 60, m)
    p = 1.0 / np.arange(1, k + 1) ** 1.1
    p /= p.sum()
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    indices = np.concatenate([np.sort(rng.choice(k, l, replace=False, p=p)) for l in lens]).astype(np.int32)
    data = rng.uniform(0.5, 1.5, indices.size)
    wide = np.complex128 if np.dtype(dtype).kind == "c" else np.float64
    if np.dtype(dtype).kind == "c":
        data = data + 1j * rng.uniform(0.5, 1.5, indices.size)
    a = sps.csr_matrix((data.astype(dtype), indices, indptr), shape=(m, k))
    b = dense((k, n), dtype, 18)
    plain = gpu.dot_product_mkl(a, b)
    assert gpu.mi_get_counter("spmm_last_tagged") == 0.0
    gpu.mi_set_option("spmm_hot_force", 1)
    gpu.mi_set_option("spmm_hot_kb", 64)
    try:
        tagged = gpu.dot_product_mkl(a, b)
        assert gpu.mi_get_counter("spmm_last_tagged") == 1.0
        assert 0.0 < gpu.mi_get_counter("spmm_hot_coverage") < 1.0
    finally:
        gpu.mi_set_option("spmm_hot_force", 0)
        gpu.mi_set_option("spmm_hot_kb", 8192)
    assert np.array_equal(tagged, plain)  # cache policy must not change a single bit
    assert rel_err(tagged, oracle.spmm(a.astype(wide), b.astype(wide))) <= tol(dtype)


def test_order_invalidates_the_cached_tagged_columns(gpu):
    """The SpMM plan caches a hot/cold-tagged copy of the column indices in storage order; mi_sparse_order moves
    the entries, so a product after ordering must not use the stale copy."""
    rng = np.random.default_rng(19)
    m, k, n = 600, 500, 32
    lens = rng.integers(1, 40, m)
    p = 1.0 / np.arange(1, k + 1) ** 1.1
    p /= p.sum()
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    indices = np.concatenate([rng.choice(k, l, replace=False, p=p) for l in lens]).astype(np.int32)  # unsorted rows
    data = rng.uniform(0.5, 1.5, indices.size)
    a = sps.csr_matrix((data, indices, indptr), shape=(m, k))
    assert not a.has_sorted_indices
    b = dense((k, n), np.float64, 20)
    want = a.toarray() @ b
    gpu.mi_set_option("spmm_hot_force", 1)
    gpu.mi_set_option("spmm_hot_kb", 64)
    try:
        A = gpu.to_device(a)
        first = gpu.dot_product_mkl(A, b)
        assert gpu.mi_get_counter("spmm_last_tagged") == 1.0
        A.handle.order()
        second = gpu.dot_product_mkl(A, b)
        assert gpu.mi_get_counter("spmm_last_tagged") == 1.0
        A.free()
    finally:
        gpu.mi_set_option("spmm_hot_force", 0)
        gpu.mi_set_option("spmm_hot_kb", 8192)
    assert rel_err(first, want) <= 1e-12 and rel_err(second, want) <= 1e-12


def test_spmm_determinism(gpu):
    a = pos_csr(4000, 3000, 0.02, np.float32, 5)
    b = dense((3000, 128), np.float32, 6)
    r1 = gpu.dot_product_mkl(a, b)
    r2 = gpu.dot_product_mkl(a, b)
    assert np.array_equal(r1, r2)  # no atomics in the SpMM path: bitwise reproducible


@pytest.mark.parametrize("order", ["C", "F"])
@pytest.mark.parametrize("fmt", ["csr", "csc", "bsr"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.complex128])
def test_spmm_both_sides_layouts_formats(gpu, oracle, dtype, fmt, order):
    a = pos_csr(120, 90, 0.1, dtype, 11)
    a_f = a.asformat(fmt) if fmt != "bsr" else a.tobsr(blocksize=(10, 10))
    b = dense((90, 33), dtype, 12, order)
    wide = np.complex128 if np.dtype(dtype).kind == "c" else np.float64
    want = a.astype(wide).toarray() @ b.astype(wide)
    got = gpu.dot_product_mkl(a_f, b)
    assert got.flags["C_CONTIGUOUS" if order == "C" else "F_CONTIGUOUS"]
    assert rel_err(got, want) <= tol(dtype)
    # dense on the left
    d = dense((33, 120), dtype, 13, order)
    want = d.astype(wide) @ a.astype(wide).toarray()
    got = gpu.dot_product_mkl(d, a_f)
    assert got.shape == (33, 90) and rel_err(got, want) <= tol(dtype)
    out = np.asarray(np.ones((33, 90), dtype=dtype), order=order)
    got = gpu.dot_product_mkl(d, a_f, out=out, out_scalar=2.0)
    assert got is out and rel_err(got, want + 2.0) <= 4 * tol(dtype)


def test_spmm_int64_indices_and_unmodified_input(gpu, oracle):
    a = pos_csr(300, 200, 0.05, np.float64, 21)
    a64 = a.copy()
    a64.indices = a64.indices.astype(np.int64)  # (the scipy constructor would narrow them again)
    a64.indptr = a64.indptr.astype(np.int64)
    b = dense((200, 16), np.float64, 22)
    got = gpu.dot_product_mkl(a64, b)
    assert a64.indices.dtype == np.int64 and a64.indptr.dtype == np.int64  # not cast in place
    assert rel_err(got, oracle.spmm(a, b)) <= F64_TOL


def test_spmv_shapes(gpu, oracle):
    a = pos_csr(150, 80, 0.1, np.float64, 31)
    v = dense((80,), np.float64, 32)
    want = a.toarray() @ v
    r = gpu.dot_product_mkl(a, v)
    assert r.shape == (150,) and rel_err(r, want) <= F64_TOL
    r = gpu.dot_product_mkl(a, v.reshape(80, 1))
    assert r.shape == (150, 1) and rel_err(r[:, 0], want) <= F64_TOL
    u = dense((150,), np.float64, 33)
    r = gpu.dot_product_mkl(u, a)
    assert r.shape == (80,) and rel_err(r, u @ a.toarray()) <= F64_TOL
    r = gpu.dot_product_mkl(u.reshape(1, 150), a)
    assert r.shape == (1, 80) and rel_err(r[0], u @ a.toarray()) <= F64_TOL
    out = np.ones(150)
    r = gpu.dot_product_mkl(a, v, out=out, out_scalar=3.0)
    assert r is out and rel_err(r, want + 3.0) <= 4 * F64_TOL


@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.complex128])
@pytest.mark.parametrize("chunk", [128, 256, 1024])
def test_spmv_skewed_rows(gpu, oracle, dtype, chunk):
    """The SpMV kernel (lanes over nonzeros, per-row reduction of LDS-parked products) on hub rows that
    span many chunks, rows of every length around the 32-product lane / wave split, and empty runs."""
    rng = np.random.default_rng(23)
    lens = np.concatenate([rng.integers(0, 70, 2500), [0] * 40, [31, 32, 33, 64, 127, 128, 129, 4000, 1, 700], [0] * 25])
    ncols = 5000
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    indices = np.concatenate([np.sort(rng.choice(ncols, l, replace=False)) for l in lens]).astype(np.int32)
    data = rng.uniform(0.5, 1.5, indices.size)
    wide = np.complex128 if np.dtype(dtype).kind == "c" else np.float64
    if np.dtype(dtype).kind == "c":
        data = data + 1j * rng.uniform(0.5, 1.5, indices.size)
    a = sps.csr_matrix((data.astype(dtype), indices, indptr), shape=(lens.size, ncols))
    x = dense((ncols,), dtype, 24)
    want = a.astype(wide).toarray() @ x.astype(wide)
    gpu.mi_set_option("spmm_chunk", chunk)
    try:
        got = gpu.dot_product_mkl(a, x)
        out = np.full(lens.size, 2.0, dtype=dtype)
        got2 = gpu.dot_product_mkl(a, x, out=out, out_scalar=0.5)
        gotT = gpu.dot_product_mkl(dense((lens.size,), dtype, 25), a)     # x^T A through the cached transpose
        again = gpu.dot_product_mkl(a, x)
    finally:
        gpu.mi_set_option("spmm_chunk", 256)
    assert got.shape == (lens.size,) and rel_err(got[want != 0], want[want != 0]) <= tol(dtype)
    assert not got[want == 0].any()
    assert got2 is out and rel_err(out, want + 1.0) <= 4 * tol(dtype)
    assert rel_err(gotT, dense((lens.size,), dtype, 25).astype(wide) @ a.astype(wide).toarray()) <= tol(dtype)
    assert np.array_equal(got, again)   # deterministic for a given partition


def test_c_abi_with_device_pointers(gpu, oracle):
    """HBM-resident operands: device pointers (torch) straight into the C ABI, zero copy."""
    torch = pytest.importorskip("torch")
    from sparse_dot_amd._mi_interface import MI, matrix_descr, sparse_matrix_t, _check_return_value
    a = pos_csr(500, 400, 0.04, np.float32, 41)
    b = dense((400, 128), np.float32, 42)
    dev = torch.device("cuda", 0)
    t_ptr = torch.from_numpy(a.indptr.astype(np.int32)).to(dev)
    t_idx = torch.from_numpy(a.indices.astype(np.int32)).to(dev)
    t_val = torch.from_numpy(a.data).to(dev)
    t_b = torch.from_numpy(b).to(dev)
    t_c = torch.full((500, 128), 5.0, device=dev, dtype=torch.float32)
    gpu.mi_set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        h = sparse_matrix_t()
        _check_return_value(MI.call("mi_sparse_s_create_csr", ct.byref(h), 0, 500, 400, t_ptr.data_ptr(),
                                    t_ptr.data_ptr() + 4, t_idx.data_ptr(), t_val.data_ptr()), "create")
        _check_return_value(MI.call("mi_sparse_s_mm", 10, 2.0, h, matrix_descr(), 101, t_b.data_ptr(), 128, 128,
                                    0.5, t_c.data_ptr(), 128), "mm")
        torch.cuda.synchronize()
        got = t_c.cpu().numpy()
        _check_return_value(MI.call("mi_sparse_destroy", h), "destroy")
    finally:
        gpu.mi_set_stream(0)
    want = 2.0 * oracle.spmm(a.astype(np.float64), b.astype(np.float64)) + 2.5
    assert rel_err(got, want) <= F32_TOL


def test_device_matrix_reuse(gpu, oracle):
    """sparse_dot_amd.to_device: A stays on the GPU (one upload, cached plan) across products."""
    a = pos_csr(600, 450, 0.05, np.float64, 44)
    A = gpu.to_device(a)
    assert A.shape == (600, 450) and A.dtype == np.float64 and "resident" in repr(A)
    for seed, order in ((1, "C"), (2, "F"), (3, "C")):
        b = dense((450, 24), np.float64, seed, order)
        got = gpu.dot_product_mkl(A, b)
        assert np.array_equal(got, gpu.dot_product_mkl(a, b))      # same kernels, same bits
        assert rel_err(got, oracle.spmm(a, b)) <= F64_TOL
    d = dense((30, 600), np.float64, 5)
    assert rel_err(gpu.dot_product_mkl(d, A), d @ a.toarray()) <= F64_TOL
    v = dense((450,), np.float64, 6)
    r = gpu.dot_product_mkl(A, v)
    assert r.shape == (600,) and rel_err(r, a.toarray() @ v) <= F64_TOL
    out = np.ones((600, 24))
    b = dense((450, 24), np.float64, 7)
    assert gpu.dot_product_mkl(A, b, out=out, out_scalar=2.0) is out
    assert rel_err(out, oracle.spmm(a, b) + 2.0) <= 4 * F64_TOL
    with pytest.raises(ValueError):
        gpu.dot_product_mkl(A, b.astype(np.float32))               # no implicit casts of a resident matrix
    with pytest.raises(ValueError):
        gpu.dot_product_mkl(A, a.T.tocsr())
    A.free()
    with pytest.raises(ValueError):
        gpu.dot_product_mkl(A, b)
    csc = gpu.to_device(a.tocsc())
    assert rel_err(gpu.dot_product_mkl(csc, b), oracle.spmm(a, b)) <= F64_TOL
    csc.free()


def test_sharded_dot_product_rccl_single_rank(gpu, oracle):
    """The multi-GPU path on the one GPU available here: RCCL process group of size 1 -- broadcast(B),
    HIP kernel on device pointers, all-gather(C).  (World size 2 runs under gloo on CPU in
    tests/test_distributed_cpu.py; more ranks need more GPUs.)"""
    torch = pytest.importorskip("torch")
    import os
    import socket
    import torch.distributed as dist
    from sparse_dot_amd import distributed as D
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        a = pos_csr(900, 700, 0.03, np.float32, 55)
        b = dense((700, 128), np.float32, 56)
        bounds = D.partition_rows(a.indptr, 1)
        assert bounds.tolist() == [0, 900]
        got = D.sharded_dot_product(D.row_block(a, 0, 900), b, [900])
        assert got.shape == (900, 128) and got.dtype == np.float32
        assert rel_err(got, oracle.spmm(a.astype(np.float64), b.astype(np.float64))) <= F32_TOL
        part = D.sharded_dot_product(D.row_block(a, 0, 900), b, [900], gather=False)
        assert np.array_equal(part, got)
    finally:
        dist.destroy_process_group()
        gpu.mi_set_stream(0)


def test_rccl_single_rank_every_collective_path(gpu, oracle):
    """Every RCCL code path of sparse_dot_amd.distributed on the one GPU available here (degenerate groups of one rank), so
    that the first 8-GPU run is not also the first execution of that code on the nccl backend: ShardedCSR.dot with the three
    all-gatherv forms and both broadcast forms, the panel pipeline with its second process group for the gathers, the sparse
    x sparse product (broadcast of B's CSR arrays, gather of indices / values) and the gram matrix by output bands."""
    torch = pytest.importorskip("torch")
    import os
    import socket
    import torch.distributed as dist
    from sparse_dot_amd import distributed as D
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
        a = pos_csr(1100, 640, 0.03, np.float32, 61)
        b = dense((640, 128), np.float32, 62)
        want = oracle.spmm(a.astype(np.float64), b.astype(np.float64))
        bounds = D.partition_rows(a.indptr, 1)
        g2 = dist.new_group([0])  # the gathers' own communicator (bench.py --gpus N does the same)
        with D.ShardedCSR(a, bounds) as sh:
            bt = torch.from_numpy(b).to(dev)
            for gmode in ("bcast", "padded", "p2p"):
                for bmode in ("bcast", "scatter_allgather"):
                    got = sh.dot(bt.clone(), src=0, gather_mode=gmode, bcast_mode=bmode)
                    torch.cuda.synchronize()
                    assert rel_err(got.cpu().numpy(), want) <= F32_TOL, (gmode, bmode)
            # collectives called directly on device tensors
            full = torch.from_numpy(want.astype(np.float32)).to(dev)
            D.gather_rows_p2p(full, bounds).wait()
            D.broadcast_rows(full, 0, None, "scatter_allgather").wait()
            w = dist.broadcast(full, src=0, async_op=True)
            w.wait()
            torch.cuda.synchronize()
            assert rel_err(full.cpu().numpy(), want) <= F32_TOL
            # panel pipeline: 4 panels of 32 columns
            bp = torch.from_numpy(np.ascontiguousarray(b.reshape(640, 4, 32).transpose(1, 0, 2))).to(dev)
            cp = sh.dot_pipelined(bp, src=0, gather_group=g2, depth=2)
            torch.cuda.synchronize()
            got = cp.permute(1, 0, 2).reshape(1100, 128).cpu().numpy()
            assert rel_err(got, want) <= F32_TOL
        # sparse x sparse and gram through the same backend
        a64 = pos_csr(700, 500, 0.02, np.float64, 63)
        b64 = pos_csr(500, 900, 0.02, np.float64, 64)
        c = D.sharded_sparse_dot_product(a64, b64, [700], src=0)
        ref = (a64 @ b64).tocsr()
        ref.sort_indices()
        c.sort_indices()
        assert np.array_equal(c.indptr, ref.indptr) and np.array_equal(c.indices, ref.indices) and rel_err(c.data, ref.data) <= F64_TOL
        x = pos_csr(900, 300, 0.05, np.float64, 65)
        gm = D.sharded_gram_matrix(x)
        assert rel_err(np.triu(gm), np.triu((x.T @ x).toarray())) <= F64_TOL
    finally:
        dist.destroy_process_group()
        gpu.mi_set_stream(0)


def test_concurrent_host_threads(gpu, oracle):
    """ctypes releases the GIL, so products may be issued from several Python threads at once
    (Threading): per-thread context / scratch, handles are independent."""
    import threading
    mats = [(pos_csr(400 + 37 * t, 300, 0.05, np.float64, 70 + t), dense((300, 16 + t), np.float64, 80 + t)) for t in range(6)]
    sp = [(pos_csr(120, 150, 0.05, np.float64, 90 + t), pos_csr(150, 90, 0.05, np.float64, 95 + t)) for t in range(6)]
    want = [oracle.spmm(a, b) for a, b in mats]
    want_sp = [(a @ b).toarray() for a, b in sp]
    errors = []

    def work(t):
        try:
            for _ in range(8):
                a, b = mats[t]
                got = gpu.dot_product_mkl(a, b)
                if rel_err(got, want[t]) > F64_TOL:
                    errors.append(("spmm", t))
                x, y = sp[t]
                c = gpu.dot_product_mkl(x, y)
                if not np.allclose(c.toarray(), want_sp[t], rtol=1e-12, atol=1e-14):
                    errors.append(("spgemm", t))
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(6)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


def test_c_abi_status_codes(gpu):
    from sparse_dot_amd._mi_interface import MI, matrix_descr, sparse_matrix_t
    null = sparse_matrix_t()
    assert MI.call("mi_sparse_destroy", null) == 1                      # NOT_INITIALIZED
    assert MI.call("mi_sparse_order", null) == 1
    out = sparse_matrix_t()
    assert MI.call("mi_sparse_spmm", 10, null, null, ct.byref(out)) == 1
    a = pos_csr(10, 7, 0.5, np.float64, 1)
    b = pos_csr(9, 4, 0.5, np.float64, 2)  # inner dimensions disagree
    from sparse_dot_amd._mi_interface import SparseHandle
    with SparseHandle.from_scipy(a) as ha, SparseHandle.from_scipy(b) as hb:
        assert MI.call("mi_sparse_spmm", 10, ha.ptr, hb.ptr, ct.byref(out)) == 3   # INVALID_VALUE
        c = np.zeros((10, 3))
        x = np.zeros((7, 3))
        assert MI.call("mi_sparse_d_mm", 99, 1.0, ha.ptr, matrix_descr(), 101, x.ctypes.data, 3, 3, 0.0,
                       c.ctypes.data, 3) == 3
        assert MI.call("mi_sparse_s_mm", 10, 1.0, ha.ptr, matrix_descr(), 101, x.ctypes.data, 3, 3, 0.0,
                       c.ctypes.data, 3) == 3    # value type mismatch
        assert "value" in MI.last_error()


# ---- handles: create -> export round trips -----------------------------------------------------------
@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.complex64, np.complex128])
@pytest.mark.parametrize("fmt", ["csr", "csc"])
def test_create_export_roundtrip(gpu, dtype, fmt):
    from sparse_dot_amd._mi_interface import _create_mi_sparse, _export_mi, _destroy_mi_handle, _convert_to_csr
    a = pos_csr(60, 45, 0.15, dtype, 51, fmt)
    h, dbl, cplx = _create_mi_sparse(a)
    back = _export_mi(h, dbl, cplx, fmt + "_matrix")
    assert back.format == fmt and np.array_equal(back.toarray(), a.toarray())
    assert np.array_equal(back.indptr, a.indptr) and np.array_equal(back.indices, a.indices)
    other = "csc" if fmt == "csr" else "csr"
    conv = _export_mi(h, dbl, cplx, other + "_array")
    assert conv.format == other and np.array_equal(conv.toarray(), a.toarray()) and conv.has_canonical_format
    c = _convert_to_csr(h)
    as_csr = _export_mi(c, dbl, cplx, "csr_matrix")
    assert np.array_equal(as_csr.toarray(), a.toarray())
    _destroy_mi_handle(c)
    _destroy_mi_handle(h)
    with pytest.raises(ValueError):
        _destroy_mi_handle(h)  # double destroy is an error, not a crash


def _export_csr_via_abi(MI, h, letter, dtype, wide=False):
    from sparse_dot_amd._mi_interface import _check_return_value
    it = ct.c_int64 if wide else ct.c_int32
    npi = np.int64 if wide else np.int32
    base, r, c = ct.c_int(), it(), it()
    ps, pe, pi, pv = ct.c_void_p(), ct.c_void_p(), ct.c_void_p(), ct.c_void_p()
    name = "mi_sparse_%s_export_csr%s" % (letter, "_64" if wide else "")
    _check_return_value(MI.call(name, h, ct.byref(base), ct.byref(r), ct.byref(c), ct.byref(ps), ct.byref(pe),
                                ct.byref(pi), ct.byref(pv)), name)
    assert base.value == 0 and pe.value == ps.value + np.dtype(npi).itemsize   # rows_end == rows_start + 1
    indptr = np.frombuffer((ct.c_char * ((r.value + 1) * np.dtype(npi).itemsize)).from_address(ps.value), dtype=npi).copy()
    nnz = int(indptr[-1])
    idx = np.frombuffer((ct.c_char * (nnz * np.dtype(npi).itemsize)).from_address(pi.value), dtype=npi).copy()
    val = np.frombuffer((ct.c_char * (nnz * np.dtype(dtype).itemsize)).from_address(pv.value), dtype=dtype).copy()
    return sps.csr_matrix((val, idx, indptr), shape=(r.value, c.value))


def test_c_abi_four_array_csr_base_one_and_wide_indices(gpu):
    """Handle creation exactly as MKL's create_csr allows it: separate rows_start / rows_end arrays with
    gaps between rows (not an indptr), 1-based indices, 64-bit index arrays; plus rejection of malformed
    input with INVALID_VALUE instead of an out-of-bounds gather."""
    from sparse_dot_amd._mi_interface import MI, matrix_descr, sparse_matrix_t, _check_return_value
    a = pos_csr(50, 40, 0.2, np.float64, 77)
    ad = a.toarray()
    # 4-array form with 3 slots of padding after every row, shuffled row storage order
    lens = np.diff(a.indptr)
    order = np.random.default_rng(1).permutation(50)
    starts = np.zeros(50, dtype=np.int32)
    pos = 5
    for r in order:
        starts[r] = pos
        pos += lens[r] + 3
    ends = (starts + lens).astype(np.int32)
    col4 = np.full(pos, 0, dtype=np.int32)
    val4 = np.full(pos, np.nan)
    for r in range(50):
        col4[starts[r]:ends[r]] = a.indices[a.indptr[r]:a.indptr[r + 1]]
        val4[starts[r]:ends[r]] = a.data[a.indptr[r]:a.indptr[r + 1]]
    h = sparse_matrix_t()
    _check_return_value(MI.call("mi_sparse_d_create_csr", ct.byref(h), 0, 50, 40, starts.ctypes.data, ends.ctypes.data,
                                col4.ctypes.data, val4.ctypes.data), "create 4-array")
    assert np.array_equal(_export_csr_via_abi(MI, h, "d", np.float64).toarray(), ad)
    x = dense((40, 8), np.float64, 3)
    y = np.zeros((50, 8))
    _check_return_value(MI.call("mi_sparse_d_mm", 10, 1.0, h, matrix_descr(), 101, x.ctypes.data, 8, 8, 0.0, y.ctypes.data, 8), "mm")
    assert rel_err(y, ad @ x) <= F64_TOL
    MI.call("mi_sparse_destroy", h)
    # base 1, 64-bit indices
    ip1 = (a.indptr.astype(np.int64) + 1)
    ix1 = (a.indices.astype(np.int64) + 1)
    h = sparse_matrix_t()
    _check_return_value(MI.call("mi_sparse_d_create_csr_64", ct.byref(h), 1, 50, 40, ip1.ctypes.data, ip1.ctypes.data + 8,
                                ix1.ctypes.data, a.data.ctypes.data), "create base1 _64")
    back = _export_csr_via_abi(MI, h, "d", np.float64, wide=True)
    assert np.array_equal(back.toarray(), ad)   # (read through 64-bit index pointers; scipy narrows them again)
    assert np.array_equal(_export_csr_via_abi(MI, h, "d", np.float64, wide=False).toarray(), ad)   # narrow export of a wide handle
    MI.call("mi_sparse_destroy", h)
    # malformed: column index out of range / decreasing row pointer / wrong base
    bad_idx = a.indices.copy()
    bad_idx[7] = 40
    h = sparse_matrix_t()
    assert MI.call("mi_sparse_d_create_csr", ct.byref(h), 0, 50, 40, a.indptr.ctypes.data, a.indptr.ctypes.data + 4,
                   bad_idx.ctypes.data, a.data.ctypes.data) == 3
    assert not h
    bad_ptr = a.indptr.copy()
    bad_ptr[10], bad_ptr[11] = bad_ptr[11], bad_ptr[10] - 1
    assert MI.call("mi_sparse_d_create_csr", ct.byref(h), 0, 50, 40, bad_ptr.ctypes.data, bad_ptr.ctypes.data + 4,
                   a.indices.ctypes.data, a.data.ctypes.data) == 3
    assert MI.call("mi_sparse_d_create_csr", ct.byref(h), 7, 50, 40, a.indptr.ctypes.data, a.indptr.ctypes.data + 4,
                   a.indices.ctypes.data, a.data.ctypes.data) == 3
    assert MI.call("mi_sparse_d_create_csr", ct.byref(h), 0, -1, 40, a.indptr.ctypes.data, a.indptr.ctypes.data + 4,
                   a.indices.ctypes.data, a.data.ctypes.data) == 3
    assert MI.call("mi_sparse_d_create_csr", None, 0, 50, 40, a.indptr.ctypes.data, a.indptr.ctypes.data + 4,
                   a.indices.ctypes.data, a.data.ctypes.data) == 1


def test_c_abi_csc_export_and_syrk_spmmd_direct(gpu, oracle):
    """export_csc of a CSR-created handle, and the syrk / spmmd entry points called directly."""
    from sparse_dot_amd._mi_interface import MI, SparseHandle, sparse_matrix_t, _check_return_value
    a = pos_csr(70, 55, 0.1, np.float32, 78)
    with SparseHandle.from_scipy(a) as h:
        csc = h.export("csc_matrix")
        assert csc.format == "csc" and np.array_equal(csc.toarray(), a.toarray()) and csc.has_sorted_indices
        out = sparse_matrix_t()
        _check_return_value(MI.call("mi_sparse_syrk", 11, h.ptr, ct.byref(out)), "syrk")
        with SparseHandle(out, "s") as g:
            got = g.export("csr_matrix")
        want = oracle.syrk_sparse(a)
        got.sort_indices()
        assert np.array_equal(got.indptr, want.indptr) and np.array_equal(got.indices, want.indices)
        np.testing.assert_allclose(got.data, want.data, rtol=F32_TOL)
        assert MI.call("mi_sparse_syrk", 99, h.ptr, ct.byref(out)) == 3
        b = pos_csr(55, 30, 0.1, np.float32, 79)
        with SparseHandle.from_scipy(b) as hb:
            for layout, order in ((101, "C"), (102, "F")):
                c = np.full((70, 30), np.nan, dtype=np.float32, order=order)
                _check_return_value(MI.call("mi_sparse_s_spmmd", 10, h.ptr, hb.ptr, layout, c.ctypes.data,
                                            30 if order == "C" else 70), "spmmd")
                np.testing.assert_allclose(c, a.toarray() @ b.toarray(), rtol=F32_TOL, atol=1e-6)
            assert MI.call("mi_sparse_d_spmmd", 10, h.ptr, hb.ptr, 101, c.ctypes.data, 30) == 3   # wrong precision entry point


def test_bsr_handle_and_order(gpu):
    from sparse_dot_amd._mi_interface import SparseHandle
    a = pos_csr(40, 60, 0.2, np.float64, 61).tobsr(blocksize=(4, 4))
    with SparseHandle.from_scipy(a) as h:
        assert np.array_equal(h.export("csr_matrix").toarray(), a.toarray())
    # order: shuffled rows (with a duplicate) come back sorted, values follow, caller arrays re-ordered
    rng = np.random.default_rng(0)
    b = pos_csr(500, 9000, 0.05, np.float32, 62)   # rows of ~450 entries, some > 512 -> block tier
    sh = b.copy()
    for i in range(b.shape[0]):
        lo, hi = b.indptr[i], b.indptr[i + 1]
        p = rng.permutation(hi - lo)
        sh.indices[lo:hi] = b.indices[lo:hi][p]
        sh.data[lo:hi] = b.data[lo:hi][p]
    with SparseHandle.from_scipy(sh) as h:
        h.order()
        out = h.export("csr_matrix")
    assert np.array_equal(out.indices, b.indices) and np.array_equal(out.data, b.data)
    assert np.array_equal(sh.indices, b.indices) and np.array_equal(sh.data, b.data)  # MKL-style in-place order


def test_order_huge_row_global_tier(gpu):
    """A row longer than the LDS sort tiers (8192) goes through the global-memory bitonic sort."""
    from sparse_dot_amd._mi_interface import SparseHandle
    rng = np.random.default_rng(3)
    n = 20000
    cols0 = rng.permutation(50000)[:n].astype(np.int32)
    vals0 = rng.uniform(0.5, 1.5, n)
    small = pos_csr(30, 50000, 0.001, np.float64, 63)
    a = sps.vstack([sps.csr_matrix((vals0, cols0, [0, n]), shape=(1, 50000)), small]).tocsr()
    a_unsorted = sps.csr_matrix((np.concatenate([vals0, small.data]), np.concatenate([cols0, small.indices]),
                                 np.concatenate([[0], small.indptr + n])), shape=a.shape)
    with SparseHandle.from_scipy(a_unsorted) as h:
        h.order()
        out = h.export("csr_matrix")
    ref = a_unsorted.copy()
    ref.sort_indices()
    assert np.array_equal(out.indices, ref.indices) and np.array_equal(out.data, ref.data)


@pytest.mark.parametrize("dup", [False, True])
def test_order_big_rows_bitmap_counting_sort(gpu, dup):
    """Rows longer than the LDS comparison-sort tiers are ordered by a counting sort through an LDS column
    bitmap; a row with a repeated column falls back to the (stable) comparison sort."""
    from sparse_dot_amd._mi_interface import SparseHandle
    rng = np.random.default_rng(31)
    ncols = 200000
    lens = [20000, 3, 9000, 0, 150000, 600, 8193]
    cols = [rng.permutation(ncols)[:l].astype(np.int32) for l in lens]
    if dup:
        cols[0][5] = cols[0][9000]
        cols[4][0] = cols[4][1] = cols[4][149999]
        cols[5][7] = cols[5][300]
    ind = np.concatenate(cols)
    dat = rng.uniform(0.5, 1.5, ind.size)
    ptr = np.concatenate([[0], np.cumsum(lens)])
    a = sps.csr_matrix((dat, ind, ptr), shape=(len(lens), ncols))
    with SparseHandle.from_scipy(a) as h:
        h.order()
        out = h.export("csr_matrix")
    ri, rd = ind.copy(), dat.copy()
    for r in range(len(lens)):
        lo, hi = ptr[r], ptr[r + 1]
        o = np.argsort(ind[lo:hi], kind="stable")
        ri[lo:hi], rd[lo:hi] = ind[lo:hi][o], dat[lo:hi][o]
    assert np.array_equal(out.indices, ri) and np.array_equal(out.data, rd)


@pytest.mark.parametrize("ncols", [(1 << 23) - 1, (1 << 23) + 5, 3_000_000_0])
def test_order_small_rows_wide_matrices(gpu, ncols):
    """Rows of up to 512 entries are sorted with (column, position) keys packed in 32 bits when the matrix has
    fewer than 2^23 columns and in 64 bits otherwise: exercise both, with columns at the top of the range."""
    from sparse_dot_amd._mi_interface import SparseHandle
    rng = np.random.default_rng(41)
    lens = [512, 1, 300, 0, 511, 2, 64]
    cols = [rng.choice(np.arange(ncols - 5000, ncols), l, replace=False).astype(np.int32) for l in lens]
    cols[0][0] = ncols - 1
    cols[4][510] = 0
    ind = np.concatenate(cols)
    dat = rng.uniform(0.5, 1.5, ind.size)
    ptr = np.concatenate([[0], np.cumsum(lens)])
    a = sps.csr_matrix((dat, ind, ptr), shape=(len(lens), ncols))
    with SparseHandle.from_scipy(a) as h:
        h.order()
        out = h.export("csr_matrix")
    ref = sps.csr_matrix((dat.copy(), ind.copy(), ptr.copy()), shape=a.shape)
    ref.sort_indices()
    assert np.array_equal(out.indices, ref.indices) and np.array_equal(out.data, ref.data)


# ---- SpGEMM -----------------------------------------------------------------------------------------
def _check_spgemm(got, want, dtype):
    g = got.tocsr().copy()
    g.sort_indices()
    assert np.array_equal(g.indptr, want.indptr), "indptr"
    assert np.array_equal(g.indices, want.indices), "indices"
    assert rel_err(g.data, want.astype(np.complex128 if np.dtype(dtype).kind == "c" else np.float64).data) <= tol(dtype)


@pytest.mark.parametrize("shape", [(1 << 19, 200000, 10), (1 << 18, 1 << 20, 20)])
def test_transpose_column_histogram_through_lds(gpu, shape):
    """A CSC operand is transposed on the device (counting sort by column, handle.hip).  From 2^22 entries on, the column
    histogram runs through LDS ranges of 32 768 counters (k_col_hist_lds, round 4; 7 and 32 ranges here, a hub column in the
    first range and one in the last); option transpose_lds_hist = 0 is the one-atomic-per-entry kernel.  Same product either way."""
    m, n, per = shape
    rng = np.random.default_rng(91)
    cols = rng.integers(0, n, m * per)
    cols[rng.choice(m * per, 50000, replace=False)] = 5          # hub columns: many counts on one LDS word
    cols[rng.choice(m * per, 30000, replace=False)] = n - 3
    a = sps.coo_matrix((rng.uniform(0.5, 1.5, m * per), (np.repeat(np.arange(m), per), cols)), shape=(m, n)).tocsc()
    assert a.nnz >= 1 << 22
    b = rng.uniform(0.5, 1.5, (n, 3))
    want = a @ b
    for opt in (1, 0):
        gpu.mi_set_option("transpose_lds_hist", opt)
        try:
            got = gpu.dot_product_mkl(a, b)
        finally:
            gpu.mi_set_option("transpose_lds_hist", 1)
        assert rel_err(got, want) <= 1e-12


@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.complex64, np.complex128])
@pytest.mark.parametrize("m,n,per", [(1 << 17, 1 << 18, 20), (40000, 300, 60), (1 << 16, (1 << 20) + 5, 36), (2500, 1 << 24, 900)])
def test_transpose_stable_radix_sort(gpu, dtype, m, n, per):
    """Transposes of 2^21 entries and more are a stable radix sort of the entries by column (handle.hip, round 5: two 9-bit
    passes for 2^18 columns, one pass for 300, three for 2^20 + 5 and for 2^24): the CSC export of a CSR handle equals
    scipy's canonical tocsc() EXACTLY -- pointers, sorted indices, values -- hub columns, empty columns and a ragged last
    tile included; option transpose_radix = 0 (histogram + atomic scatter + row sort) gives the same arrays."""
    from sparse_dot_amd._mi_interface import SparseHandle
    rng = np.random.default_rng(5)
    cols = rng.integers(0, n, m * per)
    cols[rng.choice(m * per, 40000, replace=False)] = 7
    cols[rng.choice(m * per, 20000, replace=False)] = n - 2
    vals = rng.uniform(0.5, 1.5, m * per)
    if np.dtype(dtype).kind == "c":
        vals = vals + 1j * rng.uniform(0.5, 1.5, m * per)
    a = sps.coo_matrix((vals.astype(dtype), (np.repeat(np.arange(m), per), cols)), shape=(m, n)).tocsr()
    a.sort_indices()
    assert a.nnz >= (1 << 21) and a.nnz % 8192 != 0
    want = a.tocsc()
    want.sort_indices()
    for opt in (1, 0):
        gpu.mi_set_option("transpose_radix", opt)
        try:
            with SparseHandle.from_scipy(a) as h:
                got = h.export("csc_matrix")
        finally:
            gpu.mi_set_option("transpose_radix", 1)
        assert np.array_equal(got.indptr, want.indptr) and np.array_equal(got.indices, want.indices)
        assert np.array_equal(got.data, want.data)


def test_transpose_stable_radix_sort_keeps_the_order_of_duplicates(gpu):
    """Unsorted rows with duplicate entries: the transpose keeps both copies of a duplicate, in source order (stable)."""
    from sparse_dot_amd._mi_interface import SparseHandle
    rng = np.random.default_rng(6)
    m, n, per = 70000, 5000, 32
    ind = rng.integers(0, n, m * per).astype(np.int32)
    ind[1::per] = ind[0::per]  # a duplicate in every row
    dat = rng.uniform(0.5, 1.5, m * per)
    ptr = np.arange(0, m * per + 1, per).astype(np.int32)
    a = sps.csr_matrix((dat, ind, ptr), shape=(m, n))
    assert a.nnz == m * per >= 1 << 21 and not a.has_canonical_format
    with SparseHandle.from_scipy(a) as h:
        got = h.export("csc_matrix")
    order = np.argsort(ind, kind="stable")
    assert np.array_equal(got.indices, np.repeat(np.arange(m), per)[order])
    assert np.array_equal(got.data, dat[order])
    assert np.array_equal(got.indptr, np.concatenate([[0], np.cumsum(np.bincount(ind, minlength=n))]))


@pytest.mark.parametrize("upper", [False, True])
def test_spgemm_upper_bound_pass_long_rows_and_narrow_pointer(gpu, upper):
    """Phase 0 (k_row_ub / k_row_ub_long / k_narrow_ptr, round 4): rows of A on both sides of the 512-nonzero limit of the
    16-lane kernel (512, 513, 600, 5000: the longer ones are taken by a workgroup each), operands large enough for the int32
    copy of B's row pointer (>= 2^18 nonzeros of A, >= 2^16 rows of B), the option on and off; product and upper triangle
    of a gram matrix (extents cut at the diagonal by a search inside both kernels)."""
    rng = np.random.default_rng(81)
    n = 70000
    def uniform(rows, cols, per_row, seed):
        r = np.random.default_rng(seed)
        m = sps.coo_matrix((r.uniform(0.5, 1.5, rows * per_row), (np.repeat(np.arange(rows), per_row), r.integers(0, cols, rows * per_row))),
                           shape=(rows, cols)).tocsr()
        m.sort_indices()
        return m
    a = uniform(n, n, 4, 82).tolil()
    for row, cnt in ((3, 512), (4, 513), (1000, 600), (69999, 5000)):
        a[row, :] = 0
        a[row, rng.choice(n, cnt, replace=False)] = 0.75
    for col, cnt in ((7, 513), (9, 700)):  # long rows of A^T (the gram's left operand)
        a[rng.choice(np.arange(10, n - 10), cnt, replace=False), col] = 0.5
    a = a.tocsr()
    a.eliminate_zeros()
    assert a.nnz >= 1 << 18 and sorted(np.diff(a.indptr))[-4:] == [512, 513, 600, 5000] and (np.diff(a.tocsc().indptr) > 512).sum() == 2
    b = a if upper else uniform(n, n, 4, 83)
    want = (sps.triu(a.T @ a) if upper else a @ b).tocsr()
    want.sort_indices()
    for narrow in (1, 0):
        gpu.mi_set_option("spgemm_narrow_ptr", narrow)
        try:
            got = gpu.gram_matrix_mkl(a) if upper else gpu.dot_product_mkl(a, b)
        finally:
            gpu.mi_set_option("spgemm_narrow_ptr", 1)
        _check_spgemm(got, want, np.float64)


@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.complex64, np.complex128])
def test_spgemm_all_bins(gpu, oracle, dtype):
    """Rows whose product count falls in every LDS bin (<=32, <=256, <=2048) and beyond (global hash)."""
    rng = np.random.default_rng(71)
    k, n = 3000, 5000
    b = pos_csr(k, n, 0.004, dtype, 72)          # ~20 nnz per row of B
    lens = np.concatenate([rng.integers(0, 2, 300), rng.integers(2, 12, 300), rng.integers(20, 90, 100),
                           [150, 170, 300, 330, 400, 900, 1500], rng.integers(0, 3, 50)])
    m = lens.size
    indptr = np.concatenate([[0], np.cumsum(lens)])
    indices = np.concatenate([np.sort(rng.choice(k, l, replace=False)) for l in lens]).astype(np.int32)
    data = rng.uniform(0.5, 1.5, indices.size).astype(dtype)
    a = sps.csr_matrix((data, indices, indptr), shape=(m, k))
    want = oracle.spgemm(a, b)
    got = gpu.dot_product_mkl(a, b)
    assert isinstance(got, sps.csr_matrix) and got.dtype == dtype and got.shape == (m, n)
    _check_spgemm(got, want, dtype)
    got = gpu.dot_product_mkl(a, b, reorder_output=True)
    assert np.array_equal(got.indices, want.indices)  # already ordered
    # both forms of the global-memory hash path on the same problem
    for mode in (0, 1):
        gpu.mi_set_option("spgemm_force_global", 1)
        gpu.mi_set_option("spgemm_global_mode", mode)
        try:
            got = gpu.dot_product_mkl(a, b)
        finally:
            gpu.mi_set_option("spgemm_force_global", 0)
            gpu.mi_set_option("spgemm_global_mode", 0)
        _check_spgemm(got, want, dtype)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_spgemm_signed_values_all_paths(gpu, oracle, dtype):
    """SIGNED operands (standard normal: real cancellation) through every accumulation path -- LDS hash bins,
    LDS bitmap + range-partitioned classes (hub rows), global-memory hash.  The accumulators are unordered
    LDS / L2 atomics, so values are judged against the cancellation-free magnitude (|A| |B|)_ij: fp64 1e-12,
    fp32 1e-5 of it; the structure must still be bit-exact (cancelled entries are KEPT, as MKL does)."""
    rng = np.random.default_rng(171)
    k, n = 3000, 5000
    b = pos_csr(k, n, 0.004, dtype, 172)
    b.data[:] = rng.standard_normal(b.nnz).astype(dtype)
    lens = np.concatenate([rng.integers(0, 2, 300), rng.integers(2, 12, 300), rng.integers(20, 90, 100),
                           [150, 170, 300, 330, 400, 900, 1500, 2500], rng.integers(0, 3, 50)])
    m = lens.size
    indptr = np.concatenate([[0], np.cumsum(lens)])
    indices = np.concatenate([np.sort(rng.choice(k, l, replace=False)) for l in lens]).astype(np.int32)
    a = sps.csr_matrix((rng.standard_normal(indices.size).astype(dtype), indices, indptr), shape=(m, k))
    want = oracle.spgemm(a.astype(np.float64), b.astype(np.float64))   # sorted rows, zeros kept
    scale = oracle.spgemm(abs(a).astype(np.float64), abs(b).astype(np.float64))
    assert np.array_equal(want.indices, scale.indices)
    bar = (1e-5 if dtype == np.float32 else 1e-12) * scale.data

    def check(got):
        got = got.copy()
        got.sort_indices()
        assert np.array_equal(got.indptr, want.indptr) and np.array_equal(got.indices, want.indices)
        assert np.all(np.abs(got.data.astype(np.float64) - want.data) <= bar)
    check(gpu.dot_product_mkl(a, b))
    for opt, val in (("spgemm_force_global", 1), ("spgemm_lds_parts", 0)):
        gpu.mi_set_option(opt, val)
        try:
            check(gpu.dot_product_mkl(a, b))
        finally:
            gpu.mi_set_option(opt, 1 - val)
    # the gram path (A^T A, upper triangle) on signed data: dense and sparse outputs
    x = pos_csr(4000, 300, 0.05, dtype, 173)
    x.data[:] = rng.standard_normal(x.nnz).astype(dtype)
    ref = np.triu((x.T.astype(np.float64) @ x.astype(np.float64)).toarray())
    mag = np.triu((abs(x).T.astype(np.float64) @ abs(x).astype(np.float64)).toarray())
    tol_ = 1e-5 if dtype == np.float32 else 1e-12
    assert np.all(np.abs(gpu.gram_matrix_mkl(x, dense=True) - ref) <= tol_ * mag + 1e-300)
    assert np.all(np.abs(gpu.gram_matrix_mkl(x).toarray() - ref) <= tol_ * mag + 1e-300)


@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.complex64, np.complex128])
def test_spgemm_hub_rows_lds_bitmap_and_partitioned_classes(gpu, dtype):
    """Rows far beyond the LDS hash bins (tens of thousands of distinct columns): symbolic through the LDS
    column bitmap, numeric through hash-partitioned LDS classes; same structure and values as the
    global-memory hash path and as scipy (bit-exact structure, values to tolerance)."""
    rng = np.random.default_rng(5)
    k, n = 4000, 60000
    a = sps.random(200, k, density=0.002, format="lil", random_state=1, dtype=np.float64)
    a[0, rng.choice(k, 700, replace=False)] = 1.5
    a[7, rng.choice(k, 90, replace=False)] = -0.5
    a[13, rng.choice(k, 2500, replace=False)] = 0.25
    a[199, rng.choice(k, 1500, replace=False)] = 0.75
    a = a.tocsr().astype(dtype)
    b = sps.random(k, n, density=100 / n, format="csr", random_state=2, dtype=np.float64).astype(dtype)
    if np.dtype(dtype).kind == "c":
        a.data = a.data + 1j * a.data[::-1]
        b.data = b.data * (1 - 0.5j)
    want = (a.astype(np.complex128 if np.dtype(dtype).kind == "c" else np.float64) @ b).tocsr()
    want.sort_indices()
    assert np.diff(want.indptr).max() > 40000
    for parts, slice_table in ((1, 1), (1, 0), (0, 1)):  # slice table / in-kernel searches / global-memory hash
        gpu.mi_set_option("spgemm_lds_parts", parts)
        gpu.mi_set_option("spgemm_slice_table", slice_table)
        try:
            got = gpu.dot_product_mkl(a, b)
        finally:
            gpu.mi_set_option("spgemm_lds_parts", 1)
            gpu.mi_set_option("spgemm_slice_table", 1)
        _check_spgemm(got, want, dtype)


@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.complex64, np.complex128])
def test_spmm_ordered_is_spmm_plus_order(gpu, dtype):
    """mi_sparse_spmm_ordered (what reorder_output=True calls since round 5; reference _sparse_sparse.py:226-230 =
    mkl_sparse_spmm + mkl_sparse_order): long rows accumulated by rank -- already in order, mi_sparse_order leaves them and
    their values alone -- short and medium rows sorted as before.  Through the C ABI: the same row pointer and column indices
    as mi_sparse_spmm followed by mi_sparse_order, values to tolerance; complex double (no rank kernel) takes the two steps."""
    import ctypes as ct
    from sparse_dot_amd._mi_interface import MI, SparseHandle, _check_return_value, sparse_matrix_t
    rng = np.random.default_rng(21)
    k, n = 5000, 200000
    a = sps.random(400, k, density=0.003, format="lil", random_state=5, dtype=np.float64)
    a[1, rng.choice(k, 900, replace=False)] = 1.5     # long rows of C (tens of thousands of entries) ...
    a[2, rng.choice(k, 2600, replace=False)] = 0.5
    a[399, rng.choice(k, 60, replace=False)] = 2.0    # ... medium (513 .. 4096) ...
    a[17, :] = 0                                      # ... and an empty one
    a = a.tocsr()
    a.data[:] = rng.uniform(0.5, 1.5, a.nnz)
    b = sps.random(k, n, density=80 / n, format="csr", random_state=6, dtype=np.float64)
    b.data[:] = rng.uniform(0.5, 1.5, b.nnz)
    a, b = a.astype(dtype), b.astype(dtype)
    if np.dtype(dtype).kind == "c":
        a.data = a.data * (1 + 0.5j)
        b.data = b.data * (0.5 - 1j)
    wide = np.complex128 if np.dtype(dtype).kind == "c" else np.float64
    want = (a.astype(wide) @ b.astype(wide)).tocsr()
    want.sort_indices()
    lens = np.diff(want.indptr)
    assert lens.max() > 40000 and ((lens > 512) & (lens <= 4096)).any() and (lens == 0).any()
    with SparseHandle.from_scipy(a) as ha, SparseHandle.from_scipy(b) as hb:
        outs = []
        for ordered in (False, True):
            out = sparse_matrix_t()
            if ordered:
                _check_return_value(MI.call("mi_sparse_spmm_ordered", 10, ha.ptr, hb.ptr, ct.byref(out)), "spmm_ordered")
            else:
                _check_return_value(MI.call("mi_sparse_spmm", 10, ha.ptr, hb.ptr, ct.byref(out)), "spmm")
                _check_return_value(MI.call("mi_sparse_order", out), "order")
            with SparseHandle(out, ha.letter) as hc:
                outs.append(hc.export("csr_matrix"))
    for got in outs:
        assert np.array_equal(got.indptr, want.indptr) and np.array_equal(got.indices, want.indices)
        assert rel_err(got.data, want.data) <= tol(dtype)
    got = gpu.dot_product_mkl(a, b, reorder_output=True)   # the public path
    assert np.array_equal(got.indices, want.indices) and rel_err(got.data, want.data) <= tol(dtype)


@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.complex64, np.complex128])
def test_spgemm_hub_rows_accumulate_by_rank(gpu, dtype):
    """Option spgemm_rank = 1 (round 4): the symbolic phase keeps the big rows' column bitmaps, the numeric phase adds every
    product at the RANK of its column (k_spgemm_rank).  Same structure and values as scipy, and the big rows come out SORTED
    without mi_sparse_order.  Rows of A beyond one chunk of 512 nonzeros, rows of C beyond one item (4096 entries) and beyond
    one group (16 items), slices of B on both sides of the 64-entry segment threshold; a staged product re-runs the numeric
    phase on the kept pattern; the upper-triangle form goes through the same kernels (gram)."""
    rng = np.random.default_rng(15)
    k, n = 6000, 300000
    a = sps.random(300, k, density=0.002, format="lil", random_state=11, dtype=np.float64)
    a[0, rng.choice(k, 700, replace=False)] = 1.5
    a[7, rng.choice(k, 90, replace=False)] = -0.5
    a[13, rng.choice(k, 2500, replace=False)] = 0.25
    a[299, rng.choice(k, 1500, replace=False)] = 0.75
    a = a.tocsr()
    seg = a.data[a.indptr[7]:a.indptr[8]]
    seg[seg > 0] = 0  # row 7 is negative throughout, every other row positive: no sum of a row of C cancels
    a.eliminate_zeros()
    a = a.astype(dtype)
    # ~60 entries per row drawn directly (scipy's sps.random samples k * n = 1.8e9 positions without replacement: minutes)
    rb = np.random.default_rng(12)
    bi, bj = np.repeat(np.arange(k), 60), rb.integers(0, n, 60 * k)
    bv = rb.uniform(0.1, 1.0, 60 * k)
    long_rows = rng.choice(k, 40, replace=False)  # long rows of B: slices of hundreds of entries per item
    keep = ~np.isin(bi, long_rows)
    li = np.repeat(long_rows, 20000)
    lj = np.concatenate([rng.choice(n, 20000, replace=False) for _ in long_rows])
    b = sps.coo_matrix((np.concatenate([bv[keep], np.full(li.size, 0.5)]), (np.concatenate([bi[keep], li]), np.concatenate([bj[keep], lj]))),
                       shape=(k, n)).tocsr().astype(dtype)  # duplicates of the random part are summed
    b.sort_indices()
    if np.dtype(dtype).kind == "c":
        a.data = a.data + 1j * a.data[::-1]
        b.data = b.data * (1 - 0.5j)
    want = (a.astype(np.complex128 if np.dtype(dtype).kind == "c" else np.float64) @ b).tocsr()
    want.sort_indices()
    assert np.diff(want.indptr).max() > 16 * 4096
    gpu.mi_set_option("spgemm_rank", 1)
    try:
        got = gpu.dot_product_mkl(a, b)
        big = np.nonzero(np.diff(got.indptr) > 8192)[0]
        assert big.size >= 3
        if np.dtype(dtype) != np.complex128:  # complex double stays on the hash path (accumulators of a block do not fit)
            for r in big:
                assert np.all(np.diff(got.indices[got.indptr[r]:got.indptr[r + 1]]) > 0), "big rows come out sorted"
        _check_spgemm(got, want, dtype)
        got = gpu.dot_product_mkl(a, b, reorder_output=True)
        assert np.array_equal(got.indices, want.indices)
    finally:
        gpu.mi_set_option("spgemm_rank", 0)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_gram_sparse_hub_rows_accumulate_by_rank(gpu, dtype):
    """Upper triangle of A^T A with hub rows through k_spgemm_rank (extents cut at the diagonal)."""
    rng = np.random.default_rng(16)
    a = sps.random(3000, 30000, density=20 / 30000, format="lil", random_state=13, dtype=np.float64)
    a[rng.choice(3000, 800, replace=False), 5] = 1.0
    a[rng.choice(3000, 1500, replace=False), 29000] = 2.0
    a[rng.choice(3000, 1200, replace=False), 14000] = 0.5
    a = a.tocsr().astype(dtype)
    want = sps.triu((a.T.astype(np.float64) @ a.astype(np.float64))).tocsr()
    want.sort_indices()
    gpu.mi_set_option("spgemm_rank", 1)
    try:
        got = gpu.gram_matrix_mkl(a)
    finally:
        gpu.mi_set_option("spgemm_rank", 0)
    _check_spgemm(got, want, dtype)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_gram_sparse_hub_rows(gpu, dtype):
    """A^T A (upper triangle, mkl_sparse_syrk semantics) with hub rows in the product."""
    rng = np.random.default_rng(6)
    a = sps.random(3000, 30000, density=20 / 30000, format="lil", random_state=3, dtype=np.float64)
    a[rng.choice(3000, 800, replace=False), 5] = 1.0      # column 5 of A is heavy -> row 5 of A^T A is a hub
    a[rng.choice(3000, 1500, replace=False), 29000] = 2.0  # hub near the end: short upper-triangular row
    a = a.tocsr().astype(dtype)
    want = sps.triu((a.T.astype(np.float64) @ a.astype(np.float64))).tocsr()
    want.sort_indices()
    for slice_table in (1, 0):
        gpu.mi_set_option("spgemm_slice_table", slice_table)
        try:
            got = gpu.gram_matrix_mkl(a, reorder_output=True)
        finally:
            gpu.mi_set_option("spgemm_slice_table", 1)
        g = got.tocsr()
        # explicit zeros can only come from cancellation; all values here are positive
        assert np.array_equal(g.indptr, want.indptr) and np.array_equal(g.indices, want.indices)
        assert rel_err(g.data, want.data) <= tol(dtype)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_gram_sparse_wide_matrix_by_column_panels(gpu, dtype):
    """Upper triangle of A^T A (mkl_sparse_syrk, reference _gram_matrix.py:70-74) of a matrix with 2^21 + 9 columns: the
    product's hub rows go panel by panel (three panels of 2^20 columns) with the extents of every panel cut at the SHIFTED
    diagonal.  Heavy columns on both sides of every panel boundary (their rows of A^T A start in one panel's last column /
    another's first), one in the last column; same pattern and values as scipy, with and without option deterministic;
    option spgemm_col_panels = 0 is the global-memory hash."""
    rng = np.random.default_rng(31)
    W = 1 << 20
    m, n = 3000, 2 * W + 9
    rows, cols = [], []
    for r in range(m):
        c = rng.integers(0, n, 22)
        rows.append(np.full(c.size, r))
        cols.append(c)
    for hub, share in ((5, 0.9), (W - 1, 0.5), (W, 0.5), (2 * W - 1, 0.4), (2 * W, 0.4), (n - 1, 0.6), (W + 12345, 0.8)):
        rr = rng.choice(m, int(share * m), replace=False)
        rows.append(rr)
        cols.append(np.full(rr.size, hub))
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    a = sps.coo_matrix((np.ones(rows.size), (rows, cols)), shape=(m, n)).tocsr()  # duplicates summed
    a.data[:] = rng.uniform(0.5, 1.5, a.nnz)
    a = a.astype(dtype)
    want = sps.triu((a.T.astype(np.float64) @ a.astype(np.float64))).tocsr()
    want.sort_indices()
    assert np.diff(want.indptr).max() > 40000
    for panels, det in ((1, 0), (1, 1), (0, 0)):
        gpu.mi_set_option("spgemm_col_panels", panels)
        gpu.mi_set_option("deterministic", det)
        gpu.mi_get_counter("reset")
        try:
            got = gpu.gram_matrix_mkl(a, reorder_output=True)
            used = gpu.mi_get_counter("spgemm_panels")
        finally:
            gpu.mi_set_option("spgemm_col_panels", 1)
            gpu.mi_set_option("deterministic", 0)
        assert used == (3 if panels else 0)
        g = got.tocsr()
        assert np.array_equal(g.indptr, want.indptr) and np.array_equal(g.indices, want.indices)
        assert rel_err(g.data, want.data) <= tol(dtype)


def test_device_block_cache_options(gpu, oracle):
    """Released device blocks are cached for reuse (hipMalloc of large results is slow); the cache can be
    switched off, capped and trimmed through mi_sparse_set_option, and results do not depend on it."""
    a = pos_csr(400, 300, 0.05, np.float64, 5)
    b = pos_csr(300, 350, 0.05, np.float64, 6)
    want = oracle.spgemm(a, b)
    try:
        for setting in (("pool_enable", 0), ("pool_enable", 1), ("pool_max_mb", 1), ("pool_trim", 1), ("pool_max_mb", -1)):
            gpu.mi_set_option(*setting)
            for _ in range(3):
                _check_spgemm(gpu.dot_product_mkl(a, b), want, np.float64)
    finally:
        gpu.mi_set_option("pool_enable", 1)
        gpu.mi_set_option("pool_max_mb", -1)


def test_gram_sparse_unsorted_input_rows(gpu):
    """The upper-triangle shortcut (skip the part of each B row left of the diagonal by a search) needs sorted
    rows; a matrix whose rows are not sorted takes the drop-one-by-one path and gives the same result."""
    rng = np.random.default_rng(8)
    a = sps.random(500, 700, density=0.02, format="csr", random_state=4, dtype=np.float64)
    want = sps.triu(a.T @ a).tocsr()
    want.sort_indices()
    ind, dat = a.indices.copy(), a.data.copy()
    for r in range(a.shape[0]):  # shuffle inside every row
        lo, hi = a.indptr[r], a.indptr[r + 1]
        o = rng.permutation(hi - lo)
        ind[lo:hi], dat[lo:hi] = a.indices[lo:hi][o], a.data[lo:hi][o]
    shuffled = sps.csr_matrix((dat, ind, a.indptr.copy()), shape=a.shape)
    assert not shuffled.has_sorted_indices
    for m in (a, shuffled):
        g = gpu.gram_matrix_mkl(m, reorder_output=True).tocsr()
        assert np.array_equal(g.indptr, want.indptr) and np.array_equal(g.indices, want.indices)
        assert rel_err(g.data, want.data) <= 1e-12


def _rand_rows(k, n, per_row, rng):
    """k x n CSR with ~per_row distinct random columns per row (sps.random is far too slow for n ~ 1e6)."""
    cols = [np.unique(rng.integers(0, n, per_row)) for _ in range(k)]
    ptr = np.concatenate([[0], np.cumsum([c.size for c in cols])])
    ind = np.concatenate(cols).astype(np.int32)
    return sps.csr_matrix((rng.uniform(0.5, 1.5, ind.size), ind, ptr), shape=(k, n))


def _hub_case(kind):
    """Crafted SpGEMM operands around the edges of the big-row (bitmap / column-range) path."""
    rng = np.random.default_rng(sum(map(ord, kind)))
    if kind == "exact_range_multiples":
        # C rows with exactly 1024, 2048, 3072 and 4097 distinct columns (ranges hold 1024 columns when B is wide)
        n, k = 70000, 8
        bcols = [np.sort(rng.choice(n, m, replace=False)) for m in (1024, 2048, 3072, 4097, 5, 0, 8193, 16384)]
        b = sps.csr_matrix((np.concatenate([rng.uniform(0.5, 1.5, c.size) for c in bcols]), np.concatenate(bcols),
                            np.concatenate([[0], np.cumsum([c.size for c in bcols])])), shape=(k, n))
        a = sps.csr_matrix(np.vstack([np.eye(k), np.ones((2, k)), np.zeros((1, k))]))
        return a, b
    if kind == "wide_bitmap_limit":
        # B as wide as a 2^20-column bitmap allows, hub row of A selecting many B rows
        n, k = 1 << 20, 3000
        b = _rand_rows(k, n, 40, rng)
        a = sps.random(40, k, density=0.01, format="lil", random_state=12, dtype=np.float64)
        a[3, rng.choice(k, 900, replace=False)] = 1.25
        a[17, :] = 0
        return a.tocsr(), b
    if kind == "too_wide_for_bitmap":
        n, k = 1_300_000, 2000   # falls back to the global-memory hash
        b = _rand_rows(k, n, 30, rng)
        a = sps.random(30, k, density=0.01, format="lil", random_state=14, dtype=np.float64)
        a[5, rng.choice(k, 700, replace=False)] = 0.5
        return a.tocsr(), b
    if kind == "duplicates_in_b":
        # non-canonical B: repeated column entries inside a row (sorted non-strictly); they must be summed
        n, k = 50000, 600
        b0 = sps.random(k, n, density=60 / n, format="csr", random_state=15, dtype=np.float64)
        ind = np.repeat(b0.indices, 2)
        dat = np.repeat(b0.data, 2) * np.tile([0.25, 0.75], b0.nnz)
        b = sps.csr_matrix((dat, ind, b0.indptr * 2), shape=b0.shape)
        a = sps.random(20, k, density=0.02, format="lil", random_state=16, dtype=np.float64)
        a[0, rng.choice(k, 400, replace=False)] = 2.0
        return a.tocsr(), b
    if kind == "unsorted_b":
        n, k = 50000, 600
        b0 = sps.random(k, n, density=60 / n, format="csr", random_state=17, dtype=np.float64)
        ind, dat = b0.indices.copy(), b0.data.copy()
        for r in range(k):
            lo, hi = b0.indptr[r], b0.indptr[r + 1]
            o = rng.permutation(hi - lo)
            ind[lo:hi], dat[lo:hi] = b0.indices[lo:hi][o], b0.data[lo:hi][o]
        b = sps.csr_matrix((dat, ind, b0.indptr.copy()), shape=b0.shape)
        a = sps.random(20, k, density=0.02, format="lil", random_state=18, dtype=np.float64)
        a[0, rng.choice(k, 400, replace=False)] = 2.0
        return a.tocsr(), b
    raise KeyError(kind)


def test_spgemm_unsorted_b_is_sorted_on_ingest(gpu):
    """B with shuffled rows and hub rows in the product: the library multiplies by a sorted COPY of B (round 5; the global-memory
    hash before) -- same structure as with the sorted B, values to 1e-12, B's own arrays untouched; option spgemm_sort_ingest = 0
    is the old path.  mkl_sparse_spmm takes unsorted input without penalty (reference _sparse_sparse.py:35-40)."""
    rng = np.random.default_rng(77)
    n = 1 << 14
    deg = np.minimum((n / (np.arange(n) + 1.0) ** 0.8).astype(np.int64) + 2, n // 2)  # power-law rows: the first ones are hubs
    def mat(seed):
        r = np.random.default_rng(seed)
        ptr = np.concatenate([[0], np.cumsum(deg)])
        ind = np.concatenate([np.sort(r.choice(n, d, replace=False)) for d in deg]).astype(np.int32)
        return sps.csr_matrix((r.uniform(0.5, 1.5, ind.size), ind, ptr), shape=(n, n))
    a, b = mat(1), mat(2)
    ind, dat = b.indices.copy(), b.data.copy()
    for i in range(n):
        lo, hi = b.indptr[i], b.indptr[i + 1]
        o = rng.permutation(hi - lo)
        ind[lo:hi], dat[lo:hi] = ind[lo:hi][o], dat[lo:hi][o]
    bs = sps.csr_matrix((dat, ind, b.indptr.copy()), shape=b.shape)
    assert not bs.has_sorted_indices
    keep_ind, keep_dat = bs.indices.copy(), bs.data.copy()
    want = gpu.dot_product_mkl(a, b, reorder_output=True)
    for opt in (1, 0):
        gpu.mi_set_option("spgemm_sort_ingest", opt)
        try:
            got = gpu.dot_product_mkl(a, bs, reorder_output=True)
        finally:
            gpu.mi_set_option("spgemm_sort_ingest", 1)
        assert np.array_equal(got.indptr, want.indptr) and np.array_equal(got.indices, want.indices)
        assert rel_err(got.data, want.data) <= 1e-12
    assert np.array_equal(bs.indices, keep_ind) and np.array_equal(bs.data, keep_dat)


@pytest.mark.parametrize("shuffled", [False, True])
@pytest.mark.parametrize("dtype", [np.float64, np.float32, np.complex128])
def test_spgemm_wide_b_by_column_panels(gpu, dtype, shuffled):
    """B with 2^22 + 77 columns (five panels of 2^20, the last one 77 wide) and hub rows in the product: the big-row path's LDS
    bitmap does not reach that far, so the product runs panel by panel on the fast path and the rows of the result are the
    panels' rows one after the other (round 5; the global-memory hash before -- option spgemm_col_panels = 0).  Hub columns on
    both sides of every panel boundary, rows of A with and without hub status, empty panels inside a row, rows of B shuffled.
    mkl_sparse_spmm takes any width / order (reference _sparse_sparse.py:35-40).  Oracle: scipy in the wide type."""
    rng = np.random.default_rng(5)
    n, k, W = (1 << 22) + 77, 2500, 1 << 20
    cols = []
    for r in range(k):
        c = rng.integers(0, n, 24)
        if r % 3 == 0:
            c = np.concatenate([c, [W - 1, W, 2 * W - 1, 2 * W, 3 * W, 4 * W - 1, 4 * W, n - 1]])  # the boundaries, shared: collisions
        if r % 7 == 0:
            c = c[c < W]  # rows of B that live in the first panel only
        cols.append(np.unique(c))
    ptr = np.concatenate([[0], np.cumsum([c.size for c in cols])])
    ind = np.concatenate(cols).astype(np.int32)
    dat = rng.uniform(0.5, 1.5, ind.size)
    if shuffled:
        for r in range(k):
            o = rng.permutation(ptr[r + 1] - ptr[r])
            ind[ptr[r]:ptr[r + 1]], dat[ptr[r]:ptr[r + 1]] = ind[ptr[r]:ptr[r + 1]][o], dat[ptr[r]:ptr[r + 1]][o]
    b = sps.csr_matrix((dat, ind, ptr), shape=(k, n))
    a = sps.random(60, k, density=0.004, format="lil", random_state=3, dtype=np.float64)
    a[4, rng.choice(k, 1200, replace=False)] = 0.75   # hub rows of A: ~30 000 products each
    a[31, rng.choice(k, 2000, replace=False)] = 1.5
    a[50, :] = 0
    a = a.tocsr()
    a.data[:] = rng.uniform(0.5, 1.5, a.nnz)
    a, b = a.astype(dtype), b.astype(dtype)
    if np.dtype(dtype).kind == "c":
        a.data = a.data * (1 + 0.5j)
        b.data = b.data * (0.5 - 1j)
    wide = np.complex128 if np.dtype(dtype).kind == "c" else np.float64
    want = (a.astype(wide) @ b.astype(wide)).tocsr()
    want.sort_indices()
    keep = b.indices.copy()
    for opt in (1, 0):
        gpu.mi_set_option("spgemm_col_panels", opt)
        gpu.mi_get_counter("reset")
        try:
            got = gpu.dot_product_mkl(a, b)
            panels = gpu.mi_get_counter("spgemm_panels")
            srt = gpu.dot_product_mkl(a, b, reorder_output=True)
        finally:
            gpu.mi_set_option("spgemm_col_panels", 1)
        assert panels == (5 if opt else 0)
        _check_spgemm(got, want, dtype)
        assert np.array_equal(srt.indices, want.indices) and np.array_equal(srt.indptr, want.indptr)
    assert np.array_equal(b.indices, keep)


def test_spgemm_wide_b_panels_with_gaps_int64_and_deterministic(gpu):
    """Column panels again: B 3 x 2^20 + 5 columns with NOTHING in the second panel and a single column in the last one, rows
    of A that are big in one panel and small in another, int64 index arrays, and option deterministic (values re-formed in a
    fixed order on the final pattern): same structure as scipy, values to 1e-12, three deterministic runs bit-identical."""
    rng = np.random.default_rng(11)
    W = 1 << 20
    n, k = 3 * W + 5, 1800
    cols = []
    for r in range(k):
        c = rng.integers(0, W, 30)                      # panel 0
        if r % 2:
            c = np.concatenate([c, 2 * W + rng.integers(0, W, 4)])  # panel 2, thin
        if r % 5 == 0:
            c = np.concatenate([c, [n - 1]])            # panel 3: one column
        cols.append(np.unique(c))
    ptr = np.concatenate([[0], np.cumsum([c.size for c in cols])]).astype(np.int64)
    ind = np.concatenate(cols).astype(np.int64)
    b = sps.csr_matrix((rng.uniform(0.5, 1.5, ind.size), ind, ptr), shape=(k, n))
    a = sps.random(40, k, density=0.003, format="lil", random_state=8, dtype=np.float64)
    a[7, rng.choice(k, 900, replace=False)] = 1.0
    a[8, rng.choice(k, 1500, replace=False)] = 1.0
    a = a.tocsr()
    a.data[:] = rng.uniform(0.5, 1.5, a.nnz)
    a.indices, a.indptr = a.indices.astype(np.int64), a.indptr.astype(np.int64)
    want = (a @ b).tocsr()
    want.sort_indices()
    gpu.mi_get_counter("reset")
    got = gpu.dot_product_mkl(a, b)
    assert gpu.mi_get_counter("spgemm_panels") == 4
    _check_spgemm(got, want, np.float64)
    gpu.mi_set_option("deterministic", 1)
    try:
        runs = [gpu.dot_product_mkl(a, b) for _ in range(3)]
    finally:
        gpu.mi_set_option("deterministic", 0)
    for r in runs:
        _check_spgemm(r, want, np.float64)
        assert np.array_equal(r.indices, runs[0].indices) and np.array_equal(r.data, runs[0].data)


@pytest.mark.parametrize("dtype", [np.float64, np.complex64])
@pytest.mark.parametrize("kind", ["exact_range_multiples", "wide_bitmap_limit", "too_wide_for_bitmap", "duplicates_in_b",
                                  "unsorted_b"])
def test_spgemm_big_row_edges(gpu, kind, dtype):
    a, b = _hub_case(kind)
    a, b = a.astype(dtype), b.astype(dtype)
    if np.dtype(dtype).kind == "c":
        a.data = a.data * (1 + 0.5j)
        b.data = b.data * (0.5 - 1j)
    wide = np.complex128 if np.dtype(dtype).kind == "c" else np.float64
    want = (a.astype(wide) @ b.astype(wide)).tocsr()   # scipy sums duplicates and keeps structural entries
    want.sort_indices()
    got = gpu.dot_product_mkl(a, b)
    _check_spgemm(got, want, dtype)
    got = gpu.dot_product_mkl(a, b, reorder_output=True)
    assert np.array_equal(got.indices, want.indices) and np.array_equal(got.indptr, want.indptr)


def test_spgemm_keeps_cancelled_entries_and_sums_duplicates(gpu):
    a = sps.csr_matrix(np.array([[1.0, -1.0, 0.0], [0.0, 2.0, 0.0]]))
    b = sps.csr_matrix(np.array([[1.0, 0.0], [1.0, 0.0], [0.0, 3.0]]))
    c = gpu.dot_product_mkl(a, b, reorder_output=True)
    assert c.nnz == 2 and c[0, 0] == 0.0 and (0 in c.indices[c.indptr[0]:c.indptr[1]])  # explicit zero kept (MKL)
    ua = sps.csr_matrix((np.array([1.0, 2.0, 3.0, 4.0, 5.0]), np.array([2, 0, 2, 1, 0]), np.array([0, 3, 5])), shape=(2, 3))
    ub = sps.csr_matrix((np.array([1.0, 2.0, 3.0, 4.0]), np.array([1, 0, 1, 1]), np.array([0, 2, 3, 4])), shape=(3, 2))
    c = gpu.dot_product_mkl(ua, ub)
    np.testing.assert_allclose(c.toarray(), ua.toarray() @ ub.toarray(), rtol=1e-14)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_spgemm_dense_output_and_formats(gpu, oracle, dtype):
    a = pos_csr(130, 170, 0.05, dtype, 81)
    b = pos_csr(170, 90, 0.05, dtype, 82)
    want = oracle.spgemm(a, b)
    d = gpu.dot_product_mkl(a, b, dense=True)
    assert isinstance(d, np.ndarray) and d.flags.c_contiguous
    np.testing.assert_allclose(d, want.toarray(), rtol=tol(dtype), atol=tol(dtype))
    out = np.full((130, 90), 9.0, dtype=dtype)
    assert gpu.dot_product_mkl(a, b, dense=True, out=out) is out
    np.testing.assert_allclose(out, want.toarray(), rtol=tol(dtype), atol=tol(dtype))  # overwritten, not added
    for fa, fb, cls in (("csc", "csc", sps.csc_matrix), ("csr", "csc", sps.csr_matrix), ("csc", "csr", sps.csc_matrix)):
        r = gpu.dot_product_mkl(a.asformat(fa), b.asformat(fb))
        assert isinstance(r, cls)
        np.testing.assert_allclose(r.toarray(), want.toarray(), rtol=tol(dtype), atol=tol(dtype))
    r = gpu.dot_product_mkl(sps.csr_array(a), sps.csr_array(b))
    assert isinstance(r, sps.csr_array)
    ab, bb = a.tobsr(blocksize=(10, 10)), b.tobsr(blocksize=(10, 10))
    r = gpu.dot_product_mkl(ab, bb)
    assert r.format == "bsr" and r.blocksize == (10, 10)
    np.testing.assert_allclose(r.toarray(), want.toarray(), rtol=tol(dtype), atol=tol(dtype))


def test_spgemm_low_density_and_full_density(gpu, oracle):
    for m, k, n, d in ((2000, 3000, 1000, 5e-4), (2000, 3000, 1000, 5e-6), (10, 50, 20, 1.0)):
        a = pos_csr(m, k, d, np.float64, 91)
        b = pos_csr(k, n, d, np.float64, 92)
        got = gpu.dot_product_mkl(a, b)
        if a.nnz == 0 or b.nnz == 0:
            assert got.nnz == 0 and got.shape == (m, n)
            continue
        _check_spgemm(got, oracle.spgemm(a, b), np.float64)


# ---- gram ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("transpose", [False, True])
def test_gram_paths(gpu, oracle, dtype, transpose):
    a = pos_csr(400, 150, 0.05, dtype, 101)
    ad = a.astype(np.float64).toarray()
    full = ad @ ad.T if transpose else ad.T @ ad
    want = np.triu(full)
    t = 10 * tol(dtype) if dtype == np.float32 else tol(dtype)  # reference tests use decimal=5 for fp32 gram
    s = gpu.gram_matrix_mkl(a, transpose=transpose)
    assert isinstance(s, sps.csr_matrix)
    np.testing.assert_allclose(s.toarray(), want, rtol=t, atol=t)
    so = oracle.syrk_sparse(a, aat=transpose)
    g = s.copy()
    g.sort_indices()
    assert np.array_equal(g.indptr, so.indptr) and np.array_equal(g.indices, so.indices)
    assert _rows_sorted(gpu.gram_matrix_mkl(a, transpose=transpose, reorder_output=True))
    d = gpu.gram_matrix_mkl(a, transpose=transpose, dense=True)
    np.testing.assert_allclose(d, want, rtol=t, atol=t)            # strict lower triangle is zero
    out = np.ones_like(d)
    r = gpu.gram_matrix_mkl(a, transpose=transpose, dense=True, out=out, out_scalar=2.0)
    assert r is out
    iu = np.triu_indices(out.shape[0])
    np.testing.assert_allclose(out[iu], (want + 2.0)[iu], rtol=t, atol=t)
    assert (np.tril(out, -1) == np.tril(np.ones_like(out), -1)).all()  # lower triangle untouched
    for order in ("C", "F"):
        dd = gpu.gram_matrix_mkl(np.asarray(a.toarray(), order=order), transpose=transpose)
        iu = np.triu_indices(dd.shape[0])
        np.testing.assert_allclose(dd[iu], want[iu], rtol=t, atol=t)
    c = gpu.gram_matrix_mkl(a.tocsc(), transpose=transpose, cast=True, dense=True)
    np.testing.assert_allclose(c, want, rtol=t, atol=t)


def _rows_sorted(m):
    return all(np.all(np.diff(m.indices[m.indptr[i]:m.indptr[i + 1]]) > 0) for i in range(m.shape[0]))


# ---- dense fallback (MFMA) -------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.complex64, np.complex128])
@pytest.mark.parametrize("shape", [(1, 1, 1), (5, 7, 3), (64, 64, 64), (65, 130, 33), (200, 17, 300)])
def test_dense_gemm(gpu, dtype, shape):
    m, k, n = shape
    wide = np.complex128 if np.dtype(dtype).kind == "c" else np.float64
    for oa in ("C", "F"):
        for ob in ("C", "F"):
            a, b = dense((m, k), dtype, 111, oa), dense((k, n), dtype, 112, ob)
            got = gpu.dot_product_mkl(a, b)
            want = a.astype(wide) @ b.astype(wide)
            assert got.dtype == dtype and got.flags["C_CONTIGUOUS" if oa == "C" else "F_CONTIGUOUS"]
            assert rel_err(got, want) <= tol(dtype) * 4
    a, b = dense((m, k), dtype, 113), dense((k, n), dtype, 114)
    out = np.ones((m, n), dtype=dtype)
    got = gpu.dot_product_mkl(a, b, out=out, out_scalar=3.0)
    assert got is out and rel_err(got, a.astype(wide) @ b.astype(wide) + 3.0) <= tol(dtype) * 4


def test_dense_gemm_is_transpose_detecting(gpu):
    """Asymmetric operands: A = I must return B exactly (catches swapped MFMA row/col maps)."""
    for dtype in (np.float32, np.float64):
        b = np.arange(70 * 90, dtype=dtype).reshape(70, 90)
        got = gpu.dot_product_mkl(np.eye(70, dtype=dtype), b)
        assert np.array_equal(got, b)


def test_config1_reference_cpu_case(gpu, oracle):
    """BASELINE configs[0]: scipy.sparse.random CSR 10k x 10k density 0.01 fp64 x dense 10k x 64 through
    the public API with host arrays (the reference's own CPU-runnable case), fp64 within 1e-12 of the
    oracle and of scipy."""
    a = sps.random(10000, 10000, density=0.01, format="csr", dtype=np.float64, random_state=0)
    b = np.random.default_rng(1).random((10000, 64))
    got = gpu.dot_product_mkl(a, b)
    want = oracle.spmm(a, b)
    assert got.shape == (10000, 64) and got.dtype == np.float64 and got.flags.c_contiguous
    assert rel_err(got, want) <= F64_TOL
    assert rel_err(got, a @ b) <= F64_TOL
    out = np.ones_like(got)
    assert gpu.dot_product_mkl(a, b, out=out, out_scalar=0.5) is out
    assert rel_err(out, want + 0.5) <= F64_TOL


# ---- properties at BASELINE config-2 scale ------------------------------------------------------------------
def test_config2_scale_properties(gpu):
    """R-MAT 2^20 x 2^20 (~31 M nnz) x dense 2^20 x 128 fp32, all device resident:
    (1) A @ ones = row sums of A;  (2) linearity A(B1 + B2) = A B1 + A B2;  (3) a row sample equals
    an fp64 evaluation.  Size-independent checks -- the oracle would take minutes here."""
    torch = pytest.importorskip("torch")
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    import bench
    from sparse_dot_amd._mi_interface import MI, matrix_descr, sparse_matrix_t, _check_return_value
    dev = torch.device("cuda", 0)
    indptr, indices, vals, n = bench.rmat_csr(torch, 20, 32, 7, dev)
    nnz = indices.numel()
    assert 30_000_000 < nnz < 33_000_000
    N = 128
    gpu.mi_set_stream(torch.cuda.current_stream().cuda_stream)
    h = sparse_matrix_t()
    try:
        _check_return_value(MI.call("mi_sparse_s_create_csr", ct.byref(h), 0, n, n, indptr.data_ptr(),
                                    indptr.data_ptr() + 4, indices.data_ptr(), vals.data_ptr()), "create")

        def mm(b, c, alpha=1.0, beta=0.0):
            _check_return_value(MI.call("mi_sparse_s_mm", 10, alpha, h, matrix_descr(), 101, b.data_ptr(), N, N, beta,
                                        c.data_ptr(), N), "mm")
        ones = torch.ones((n, N), device=dev)
        c1 = torch.empty((n, N), device=dev)
        mm(ones, c1)
        torch.cuda.synchronize()
        ip = indptr.to(torch.int64)
        rowsum = torch.zeros(n, device=dev, dtype=torch.float64)
        rowsum.index_add_(0, torch.repeat_interleave(torch.arange(n, device=dev), ip[1:] - ip[:-1]), vals.double())
        err = ((c1[:, 0].double() - rowsum).abs() / rowsum.clamp(min=1e-30))[rowsum > 0].max().item()
        assert err <= F32_TOL, err
        assert (c1[rowsum == 0] == 0).all()          # empty rows are zero
        assert torch.equal(c1[:, :1].expand(-1, N), c1)  # every column identical
        g = torch.Generator(device=dev)
        g.manual_seed(1)
        b1 = torch.rand((n, N), generator=g, device=dev) + 0.5
        b2 = torch.rand((n, N), generator=g, device=dev) + 0.5
        r1, r2, r12 = (torch.empty((n, N), device=dev) for _ in range(3))
        mm(b1, r1)
        mm(b2, r2)
        mm(b1 + b2, r12)
        torch.cuda.synchronize()
        lin = ((r12 - (r1 + r2)).abs() / r12.abs().clamp(min=1e-30)).max().item()
        assert lin <= 4 * F32_TOL, lin
        # beta path: C := A b1 + 1.0 * C(=r2)  == r1 + r2
        acc = r2.clone()
        mm(b1, acc, 1.0, 1.0)
        torch.cuda.synchronize()
        assert ((acc - (r1 + r2)).abs() / acc.abs().clamp(min=1e-30)).max().item() <= 4 * F32_TOL
        # (3) fp64 row sample on the kernels the bench TIMES: by now the handle has taken more than two products, so this one runs
        # the column-partitioned plan (long rows by XCD partition + combine, short rows row-owned) -- asserted through the
        # library's own counter; rows: the 8 longest, 8 of 64..127 entries (just past the partition threshold), 24 random
        r3 = torch.empty((n, N), device=dev)
        mm(b1, r3)
        torch.cuda.synchronize()
        assert gpu.mi_get_counter("spmm_last_kpart") == 8.0
        lens = ip[1:] - ip[:-1]
        mid = torch.nonzero((lens >= 64) & (lens < 128)).flatten()
        sel = torch.cat([torch.topk(lens, 8).indices, mid[torch.randperm(mid.numel(), device=dev)[:8]],
                         torch.randint(0, n, (24,), device=dev)])
        assert mid.numel() >= 8
        for res in (r1, r3):  # r1: the second product of the handle (row-owned kernel); r3: the partitioned plan
            for r in sel.tolist():
                lo, hi = int(ip[r]), int(ip[r + 1])
                if hi == lo:
                    assert not res[r].any()
                    continue
                want = (vals[lo:hi].double()[:, None] * b1[indices[lo:hi].long()].double()).sum(0)
                assert ((res[r].double() - want).abs() / want.abs()).max().item() <= F32_TOL
    finally:
        if h:
            MI.call("mi_sparse_destroy", h)
        gpu.mi_set_stream(0)


def _uniform_csr_torch(torch, dev, n_rows, n_cols, per_row, seed, dtype):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    rows = torch.arange(n_rows, device=dev, dtype=torch.int64).repeat_interleave(per_row)
    cols = torch.randint(0, n_cols, (n_rows * per_row,), generator=g, device=dev, dtype=torch.int64)
    key = torch.unique(rows * n_cols + cols)
    idx = (key % n_cols).to(torch.int32)
    ip = torch.zeros(n_rows + 1, dtype=torch.int64, device=dev)
    ip[1:] = torch.cumsum(torch.bincount(key // n_cols, minlength=n_rows), 0)
    val = (torch.rand(idx.numel(), generator=g, device=dev, dtype=torch.float64) + 0.5).to(dtype)
    return ip.to(torch.int32), idx, val


def test_config3_scale_properties(gpu):
    """BASELINE configs[2] shape (uniform variant): two CSR 2^20 x 2^20, 16 nnz/row, fp64, device resident.
    Size-independent checks: C 1 = A (B 1) to 1e-12; nnz(C) <= #products; after mi_sparse_order every row
    of the exported C is strictly increasing (no duplicate columns) and every value is positive."""
    torch = pytest.importorskip("torch")
    from sparse_dot_amd._mi_interface import MI, SparseHandle, matrix_descr, sparse_matrix_t, _check_return_value
    dev = torch.device("cuda", 0)
    n = 1 << 20
    a = _uniform_csr_torch(torch, dev, n, n, 16, 1, torch.float64)
    b = _uniform_csr_torch(torch, dev, n, n, 16, 2, torch.float64)
    gpu.mi_set_stream(torch.cuda.current_stream().cuda_stream)
    handles = []
    try:
        def mk(t):
            h = sparse_matrix_t()
            _check_return_value(MI.call("mi_sparse_d_create_csr", ct.byref(h), 0, n, n, t[0].data_ptr(), t[0].data_ptr() + 4,
                                        t[1].data_ptr(), t[2].data_ptr()), "create")
            handles.append(h)
            return h
        ha, hb = mk(a), mk(b)
        hc = sparse_matrix_t()
        _check_return_value(MI.call("mi_sparse_spmm", 10, ha, hb, ct.byref(hc)), "spmm")
        handles.append(hc)

        def mv(h, x, y):
            _check_return_value(MI.call("mi_sparse_d_mv", 10, 1.0, h, matrix_descr(), x.data_ptr(), 0.0, y.data_ptr()), "mv")
        ones = torch.ones(n, device=dev, dtype=torch.float64)
        b1, ab1, c1 = (torch.empty(n, device=dev, dtype=torch.float64) for _ in range(3))
        mv(hb, ones, b1)
        mv(ha, b1, ab1)
        mv(hc, ones, c1)
        torch.cuda.synchronize()
        assert float(((c1 - ab1).abs() / ab1.abs().clamp(min=1e-300)).max()) <= F64_TOL
        products = float((torch.bincount(a[1].long(), minlength=n).double() * (b[0][1:] - b[0][:-1]).double()).sum())
        wrapped = SparseHandle(hc, "d")
        handles.remove(hc)
        rows, cols, nnz, letter, _ = wrapped.info()
        assert (rows, cols, letter) == (n, n, "d") and 0 < nnz <= products
        assert nnz > 2.6e8  # ~2.68e8 for these seeds: almost no collisions at this density
        wrapped.order()
        c = wrapped.export("csr_matrix")
        wrapped.destroy()
        assert c.nnz == nnz and c.indptr[0] == 0 and c.indptr[-1] == nnz and np.all(np.diff(c.indptr) >= 0)
        d = np.diff(c.indices.astype(np.int64))
        row_starts = c.indptr[1:-1]
        row_starts = row_starts[(row_starts > 0) & (row_starts < nnz)]
        inside = np.ones(nnz - 1, dtype=bool)
        inside[row_starts - 1] = False          # pairs that straddle a row boundary
        assert np.all(d[inside] > 0), "columns inside a row must be strictly increasing after order()"
        assert c.data.min() > 0.0
    finally:
        for h in handles:
            MI.call("mi_sparse_destroy", h)
        gpu.mi_set_stream(0)


def test_config4_scaled_gram_properties(gpu):
    """BASELINE configs[3], scaled (the literal 4 M x 256 k dense output is 262 GB): uniform 2^20 x 16384,
    64 nnz/row, fp32, dense=True on device.  Checks: diag(C) = column sums of A.^2; sampled entries equal
    the dot product of the two columns (float64 reference); strict lower triangle untouched; and the same
    call through beta = 1 doubles the upper triangle."""
    torch = pytest.importorskip("torch")
    from sparse_dot_amd._mi_interface import MI, sparse_matrix_t, _check_return_value
    dev = torch.device("cuda", 0)
    m, n = 1 << 20, 16384
    ip, idx, val = _uniform_csr_torch(torch, dev, m, n, 64, 3, torch.float32)
    gpu.mi_set_stream(torch.cuda.current_stream().cuda_stream)
    h = sparse_matrix_t()
    try:
        _check_return_value(MI.call("mi_sparse_s_create_csr", ct.byref(h), 0, m, n, ip.data_ptr(), ip.data_ptr() + 4,
                                    idx.data_ptr(), val.data_ptr()), "create")
        C = torch.full((n, n), -7.0, device=dev, dtype=torch.float32)
        _check_return_value(MI.call("mi_sparse_s_syrkd", 11, h, 1.0, 0.0, C.data_ptr(), 101, n), "syrkd")
        torch.cuda.synchronize()
        colsq = torch.zeros(n, device=dev, dtype=torch.float64)
        colsq.index_add_(0, idx.long(), val.double() ** 2)
        assert float(((torch.diagonal(C).double() - colsq).abs() / colsq).max()) <= F32_TOL
        assert bool((torch.tril(C, -1) == torch.tril(torch.full_like(C, -7.0), -1)).all())  # never touched
        # sampled off-diagonal entries against float64 column dot products (scipy on the host)
        a_host = sps.csr_matrix((val.cpu().numpy().astype(np.float64), idx.cpu().numpy(), ip.cpu().numpy()), shape=(m, n)).tocsc()
        rng = np.random.default_rng(0)
        for _ in range(40):
            i, j = sorted(rng.integers(0, n, 2).tolist())
            want = float(a_host[:, [i]].multiply(a_host[:, [j]]).sum())
            got = float(C[i, j])
            assert abs(got - want) <= F32_TOL * max(abs(want), 1e-30) + 1e-30 or (want == 0 and got == 0), (i, j, got, want)
        upper = torch.triu(C).clone()
        _check_return_value(MI.call("mi_sparse_s_syrkd", 11, h, 1.0, 1.0, C.data_ptr(), 101, n), "syrkd")
        torch.cuda.synchronize()
        rel = ((torch.triu(C) - 2 * upper).abs() / (2 * upper).abs().clamp(min=1e-30)).max()
        assert float(rel) <= 4 * F32_TOL
    finally:
        if h:
            MI.call("mi_sparse_destroy", h)
        gpu.mi_set_stream(0)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_gram_dense_multi_tile_every_walk(gpu, dtype):
    """Dense gram whose output rows span several LDS tiles (n = 70 000: 3-9 tiles per row depending on dtype / tile
    size), device resident, through every walk the library has: sliced (slice table + 8 lanes per selected row) and
    whole-row, 128 and 64 KiB tiles, persistent and one-workgroup-per-tile grids, row bands (mi_sparse_?_syrkd_rows),
    and an input whose rows are NOT sorted (falls back to the whole-row walk).  Sampled output rows equal scipy's
    float64 A^T A on and above the diagonal to the north-star tolerance; left of the diagonal the array is untouched."""
    torch = pytest.importorskip("torch")
    from sparse_dot_amd._mi_interface import MI, sparse_matrix_t, _check_return_value
    dev = torch.device("cuda", 0)
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    pre = "s" if dtype == np.float32 else "d"
    tolv = F32_TOL if dtype == np.float32 else F64_TOL
    m, n = 30000, 70000
    ip, idx, val = _uniform_csr_torch(torch, dev, m, n, 24, 11, tdt)
    a_host = sps.csr_matrix((val.cpu().numpy().astype(np.float64), idx.cpu().numpy(), ip.cpu().numpy()), shape=(m, n))
    want = (a_host.T @ a_host).tocsr()
    rng = np.random.default_rng(3)
    sample = np.unique(np.concatenate([[0, 1, 16383, 16384, 32767, 32768, 65535, 65536, n - 1], rng.integers(0, n, 150)]))

    def check_rows(C, row0, rows):
        got = C[torch.as_tensor(rows - row0, device=dev)].cpu().numpy().astype(np.float64)
        for k, i in enumerate(rows.tolist()):
            ref = np.asarray(want[i].todense()).ravel()
            assert np.all(got[k, :i] == -7.0), i                                   # strict lower part never written
            assert np.all(np.abs(got[k, i:] - ref[i:]) <= tolv * np.abs(ref[i:]) + 1e-300), i

    gpu.mi_set_stream(torch.cuda.current_stream().cuda_stream)
    handles = []
    try:
        def mk(ip_, idx_, val_):
            h = sparse_matrix_t()
            _check_return_value(MI.call("mi_sparse_%s_create_csr" % pre, ct.byref(h), 0, m, n, ip_.data_ptr(), ip_.data_ptr() + 4,
                                        idx_.data_ptr(), val_.data_ptr()), "create")
            handles.append(h)
            return h
        h = mk(ip, idx, val)
        one, zero = (ct.c_float(1.0), ct.c_float(0.0)) if dtype == np.float32 else (ct.c_double(1.0), ct.c_double(0.0))
        # (sliced, tile KiB, persistent): sliced 2 = the pipelined slice walk, 0 = the whole-row walk; persistent = workgroups
        # per LDS slot walking the tile list (0: one workgroup per tile, -1: default)
        # heads = 1: slice bounds travel with the entries of X^T (rows of X <= 255 entries); 0: per-row table
        for sliced, tile_kb, persistent, heads in ((2, 128, -1, 1), (2, 64, 4, 1), (2, 64, -1, 0), (2, 128, 1, 0), (2, 128, 0, 1),
                                                   (2, 64, 0, 1), (0, 128, -1, 1), (0, 64, 1, 1), (1, 128, -1, 1),
                                                   (2, 152, -1, 1), (2, 152, 0, 0), (2, 0, -1, 1), (0, 152, -1, 1), (1, 0, -1, 1)):
            gpu.mi_set_option("gram_heads", heads)
            gpu.mi_set_option("gram_sliced", sliced)
            gpu.mi_set_option("gram_tile_kb", tile_kb)
            gpu.mi_set_option("gram_persistent", persistent)
            C = torch.full((n, n), -7.0, device=dev, dtype=tdt)
            _check_return_value(MI.call("mi_sparse_%s_syrkd" % pre, 11, h, one, zero, C.data_ptr(), 101, n), "syrkd")
            torch.cuda.synchronize()
            check_rows(C, 0, sample)
            C = None
        gpu.mi_set_option("gram_sliced", 2)
        gpu.mi_set_option("gram_tile_kb", 152)
        gpu.mi_set_option("gram_persistent", -1)
        # a band of output rows that starts inside a tile (152 KiB tiles: boundaries that are not powers of two)
        r0, r1 = 20001, 20001 + 4099
        band = torch.full((r1 - r0, n), -7.0, device=dev, dtype=tdt)
        _check_return_value(MI.call("mi_sparse_%s_syrkd_rows" % pre, 11, h, one, zero, band.data_ptr(), 101, n, r0, r1), "rows")
        torch.cuda.synchronize()
        check_rows(band, r0, np.unique(np.concatenate([[r0, r1 - 1], rng.integers(r0, r1, 40)])))
        band = None
        # column-major output with beta = 1 on a prefilled array, through A A^T of the transpose (op = 10 on a handle
        # of A^T): the scalar write-out path with the read-modify-write, multi-tile
        at = a_host.T.tocsr()
        at.sort_indices()
        tip = torch.as_tensor(at.indptr.astype(np.int32), device=dev)
        tidx = torch.as_tensor(at.indices.astype(np.int32), device=dev)
        tval = torch.as_tensor(at.data.astype(dtype), device=dev)
        ht = sparse_matrix_t()
        _check_return_value(MI.call("mi_sparse_%s_create_csr" % pre, ct.byref(ht), 0, n, m, tip.data_ptr(), tip.data_ptr() + 4,
                                    tidx.data_ptr(), tval.data_ptr()), "create")
        handles.append(ht)
        Ct = torch.full((n, n), -7.0, device=dev, dtype=tdt)   # column-major: element (i, j) lives at Ct[j, i]
        _check_return_value(MI.call("mi_sparse_%s_syrkd" % pre, 10, ht, one, one, Ct.data_ptr(), 102, n), "syrkd col-major")
        torch.cuda.synchronize()
        rows_cm = sample[::3]
        got = Ct[:, torch.as_tensor(rows_cm, device=dev)].T.cpu().numpy().astype(np.float64)
        for k, i in enumerate(rows_cm.tolist()):
            ref = np.asarray(want[i].todense()).ravel()
            assert np.all(got[k, :i] == -7.0), i
            assert np.all(np.abs(got[k, i:] - (ref[i:] - 7.0)) <= tolv * (np.abs(ref[i:]) + 7.0) + 1e-300), i
        Ct = None
        # rows of A in shuffled order inside each row: not sorted -> whole-row walk
        perm = torch.argsort(torch.rand(idx.numel(), device=dev) + torch.repeat_interleave(
            torch.arange(m, device=dev, dtype=torch.float32), (ip[1:] - ip[:-1]).long()) * 2.0)
        idx2, val2 = idx[perm].contiguous(), val[perm].contiguous()
        assert bool((idx2 != idx).any())
        h2 = mk(ip, idx2, val2)
        C = torch.full((n, n), -7.0, device=dev, dtype=tdt)
        _check_return_value(MI.call("mi_sparse_%s_syrkd" % pre, 11, h2, one, zero, C.data_ptr(), 101, n), "syrkd")
        torch.cuda.synchronize()
        check_rows(C, 0, sample[::4])
        C = None
    finally:
        gpu.mi_set_option("gram_sliced", 1)
        gpu.mi_set_option("gram_tile_kb", 0)
        gpu.mi_set_option("gram_persistent", -1)
        gpu.mi_set_option("gram_heads", 1)
        for h in handles:
            MI.call("mi_sparse_destroy", h)
        gpu.mi_set_stream(0)
        torch.cuda.empty_cache()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_deterministic_option_bitwise_reproducible(gpu, dtype):
    """mi_sparse_set_option("deterministic", 1) (sanitizer row): SpGEMM, sparse gram and dense gram give the
    SAME BITS run after run (hub rows / many products per entry, where the default path's LDS atomics add in arrival order),
    and agree with the default path within the north_star's tolerance."""
    import scipy.sparse as sps
    tol = 1e-5 if dtype == np.float32 else 1e-12
    rng = np.random.default_rng(5)

    def pos(m, n, density, seed):
        x = sps.random(m, n, density=density, format="csr", dtype=np.float64, random_state=seed)
        x.data[:] = np.random.default_rng(seed + 1).uniform(0.5, 1.5, x.nnz)
        return x.astype(dtype)
    a = sps.vstack([pos(1200, 900, 0.02, 1), pos(3, 900, 0.7, 2)]).tocsr()   # hub rows: the big-row kernels too
    b = pos(900, 5000, 0.02, 3)
    x = pos(20000, 300, 0.05, 4)  # dense gram: ~50 products per entry
    gpu.mi_set_option("deterministic", 0)
    ref_prod = gpu.dot_product_mkl(a, b, reorder_output=True)
    ref_gram = gpu.gram_matrix_mkl(x, dense=True)
    ref_sgram = gpu.gram_matrix_mkl(x, reorder_output=True)
    gpu.mi_set_option("deterministic", 1)
    try:
        runs = []
        for _ in range(3):
            p = gpu.dot_product_mkl(a, b, reorder_output=True)
            g = gpu.gram_matrix_mkl(x, dense=True)
            s = gpu.gram_matrix_mkl(x, reorder_output=True)
            runs.append((p, g, s))
        for p, g, s in runs[1:]:
            assert np.array_equal(p.indices, runs[0][0].indices) and np.array_equal(p.data, runs[0][0].data)
            assert np.array_equal(g, runs[0][1])
            assert np.array_equal(s.indices, runs[0][2].indices) and np.array_equal(s.data, runs[0][2].data)
        p, g, s = runs[0]
        assert np.array_equal(p.indptr, ref_prod.indptr) and np.array_equal(p.indices, ref_prod.indices)
        assert np.allclose(p.data, ref_prod.data, rtol=tol, atol=0)
        assert np.allclose(np.triu(g), np.triu(ref_gram), rtol=tol, atol=0)
        assert np.array_equal(s.indices, ref_sgram.indices) and np.allclose(s.data, ref_sgram.data, rtol=tol, atol=0)
    finally:
        gpu.mi_set_option("deterministic", 0)
