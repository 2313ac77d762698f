"""GPU: SURVEY section 8 f3 -- handles created from BSR arrays run SpMM on the block form (k_bsr_spmm: one column index
per block, bs consecutive rows of B per gather) instead of the CSR expansion; results must equal the expansion's."""
import numpy as np
import pytest
import scipy.sparse as sps

pytestmark = pytest.mark.gpu


def _tol(dtype):
    return 1e-5 if np.dtype(dtype) in (np.dtype(np.float32), np.dtype(np.complex64)) else 1e-12


@pytest.mark.parametrize("bs", [1, 2, 3, 4, 8])
@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.complex64, np.complex128])
def test_bsr_block_kernel_matches_expansion(gpu, oracle, dtype, bs):
    rng = np.random.default_rng(bs)
    mb, kb = 300, 260
    d = (rng.random((mb * bs, kb * bs)) < 0.02) * rng.uniform(0.5, 1.5, (mb * bs, kb * bs))
    d[5 * bs:6 * bs] = rng.uniform(0.5, 1.5, (bs, kb * bs))   # a full block row (many blocks per wave)
    d[9 * bs:12 * bs] = 0.0                                      # empty block rows
    if np.dtype(dtype).kind == "c":
        d = d + 1j * (d != 0) * rng.uniform(0.5, 1.5, d.shape)
    a = sps.bsr_matrix(d.astype(dtype), blocksize=(bs, bs))
    wide = np.complex128 if np.dtype(dtype).kind == "c" else np.float64
    for n in (128, 36, 4, 3):
        b = rng.uniform(0.5, 1.5, (kb * bs, n)).astype(dtype)
        want = oracle.spmm(sps.csr_matrix(d.astype(wide)), b.astype(wide))
        gpu.mi_get_counter("reset")
        got = gpu.dot_product_mkl(a, b)
        native = gpu.mi_get_counter("bsr_native_calls")
        expect_native = bs in (2, 4, 8) and not (bs == 8 and dtype == np.complex128) and (n * np.dtype(dtype).itemsize) % 16 == 0
        assert native == (1.0 if expect_native else 0.0), (bs, n, native)
        assert got.dtype == dtype and got.shape == want.shape
        assert np.max(np.abs(got - want) / np.maximum(np.abs(want), 1e-30)) <= _tol(dtype)
        out = np.ones_like(got)
        assert gpu.dot_product_mkl(a, b, out=out, out_scalar=2.0) is out
        assert np.max(np.abs(out - (want + 2)) / np.abs(want + 2)) <= _tol(dtype)
        # the same product through the CSR expansion (option) and with the operand on the right (op = T: expansion)
        gpu.mi_set_option("bsr_native", 0)
        try:
            ref = gpu.dot_product_mkl(a, b)
        finally:
            gpu.mi_set_option("bsr_native", 1)
        assert np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-30)) <= _tol(dtype)
        left = gpu.dot_product_mkl(np.ascontiguousarray(b.T), sps.bsr_matrix(d.T.astype(dtype), blocksize=(bs, bs)))
        assert np.max(np.abs(left - want.T) / np.maximum(np.abs(want.T), 1e-30)) <= _tol(dtype)


def test_bsr_block_layout_column_major_and_device_matrix(gpu, oracle):
    """Column-major blocks through the C ABI, and a resident BSR DeviceMatrix reused across products."""
    import ctypes as ct
    from sparse_dot_amd._mi_interface import MI, SparseHandle, matrix_descr, sparse_matrix_t, _check_return_value
    rng = np.random.default_rng(1)
    bs, mb, kb, n = 4, 50, 40, 64
    d = (rng.random((mb * bs, kb * bs)) < 0.05) * rng.uniform(0.5, 1.5, (mb * bs, kb * bs))
    a = sps.bsr_matrix(d, blocksize=(bs, bs))
    blocks_cm = np.ascontiguousarray(a.data.transpose(0, 2, 1))  # every block stored column-major
    b = rng.uniform(0.5, 1.5, (kb * bs, n))
    want = d @ b
    h = sparse_matrix_t()
    ip, idx = a.indptr.astype(np.int32), a.indices.astype(np.int32)
    _check_return_value(MI.call("mi_sparse_d_create_bsr", ct.byref(h), 0, 102, mb, kb, bs, ip.ctypes.data, ip.ctypes.data + 4,
                                idx.ctypes.data, blocks_cm.ctypes.data), "create_bsr")
    with SparseHandle(h, "d") as hh:
        c = np.empty((mb * bs, n))
        gpu.mi_get_counter("reset")
        _check_return_value(MI.call("mi_sparse_d_mm", 10, 1.0, hh.ptr, matrix_descr(), 101, b.ctypes.data, n, n, 0.0,
                                    c.ctypes.data, n), "mm")
        assert gpu.mi_get_counter("bsr_native_calls") == 1.0
        assert np.allclose(c, want, rtol=1e-12, atol=0)
    dm = gpu.to_device(a)
    try:
        for k in range(3):
            got = gpu.dot_product_mkl(dm, b * (k + 1))
            assert np.allclose(got, want * (k + 1), rtol=1e-12, atol=0)
    finally:
        dm.free()


@pytest.mark.parametrize("bs", [1, 3, 4, 10])
@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.complex128])
def test_bsr_product_exported_as_bsr(gpu, dtype, bs):
    """BSR x BSR -> BSR: the product handle remembers the block size and mi_sparse_?_export_bsr re-blocks it on the device
    (reference _common.py:503-609; MKL keeps the operands' format)."""
    rng = np.random.default_rng(bs + 40)
    d1 = (rng.random((30 * bs, 20 * bs)) < 0.05) * rng.uniform(0.5, 1.5, (30 * bs, 20 * bs))
    d2 = (rng.random((20 * bs, 25 * bs)) < 0.05) * rng.uniform(0.5, 1.5, (20 * bs, 25 * bs))
    if np.dtype(dtype).kind == "c":
        d1 = d1 + 1j * (d1 != 0)
        d2 = d2 - 1j * (d2 != 0)
    a = sps.bsr_matrix(d1.astype(dtype), blocksize=(bs, bs))
    b = sps.bsr_matrix(d2.astype(dtype), blocksize=(bs, bs))
    got = gpu.dot_product_mkl(a, b)
    # block structure = the structural product of the two BLOCK patterns (a stored block is bs x bs stored values, zeros
    # included, exactly as MKL treats BSR operands; scipy's dense round trip would drop numerically empty blocks)
    pa = sps.csr_matrix((np.ones(a.indices.size), a.indices, a.indptr), shape=(30, 20))
    pb = sps.csr_matrix((np.ones(b.indices.size), b.indices, b.indptr), shape=(20, 25))
    want = (pa @ pb).tocsr()
    want.sort_indices()
    assert isinstance(got, sps.bsr_matrix) and got.blocksize == (bs, bs) and got.dtype == dtype
    assert got.shape == (30 * bs, 25 * bs)
    assert np.array_equal(got.indptr, want.indptr) and np.array_equal(got.indices, want.indices)
    assert np.allclose(got.toarray(), d1 @ d2, rtol=_tol(dtype), atol=0)
    # the BSR array class, and a handle with no block size -> NOT_SUPPORTED through the ABI
    ga = gpu.dot_product_mkl(sps.bsr_array(a), sps.bsr_array(b))
    assert type(ga).__name__ == "bsr_array" and np.allclose(ga.toarray(), d1 @ d2, rtol=_tol(dtype), atol=0)
    from sparse_dot_amd._mi_interface import SparseHandle
    with SparseHandle.from_scipy(a.tocsr()) as h:
        with pytest.raises(ValueError, match="NOT_SUPPORTED"):
            h.export_bsr()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_bsr_handle_exports_the_arrays_it_was_created_from(gpu, dtype):
    """MKL's BSR handle aliases the caller's arrays, so create -> export hands the SAME arrays back: block order as given
    (scipy's csr -> bsr conversion leaves the blocks of a row unsorted), explicit zeros inside blocks, block layout.  The
    reference's own round-trip test compares the raw arrays (reference tests/test_mkl.py:230-249); after mi_sparse_order
    the export is the re-blocked, ordered matrix instead."""
    from sparse_dot_amd._mi_interface import SparseHandle
    m = sps.random(200, 300, density=0.1, format="csr", dtype=np.float64, random_state=50).astype(dtype)
    a = sps.bsr_matrix(m, blocksize=(2, 2))
    perm_a = a.copy()
    # make sure the block order inside rows is NOT ascending somewhere
    rng = np.random.default_rng(3)
    for i in range(0, len(a.indptr) - 1, 7):
        lo, hi = a.indptr[i], a.indptr[i + 1]
        p = rng.permutation(hi - lo)
        perm_a.indices[lo:hi] = a.indices[lo:hi][p]
        perm_a.data[lo:hi] = a.data[lo:hi][p]
    assert not np.array_equal(perm_a.indices, a.indices)
    with SparseHandle.from_scipy(perm_a) as h:
        back = h.export_bsr()
        assert back.blocksize == (2, 2) and back.dtype == dtype
        assert np.array_equal(back.indptr, perm_a.indptr)
        assert np.array_equal(back.indices, perm_a.indices)   # same block order, not a re-blocked / sorted one
        assert np.array_equal(back.data, perm_a.data)         # bit for bit, explicit zeros included
        h.order()
        ordered = h.export_bsr()
    ref = perm_a.copy()
    ref.sort_indices()
    assert np.array_equal(ordered.indptr, ref.indptr) and np.array_equal(ordered.indices, ref.indices)
    assert np.array_equal(ordered.toarray(), ref.toarray())
