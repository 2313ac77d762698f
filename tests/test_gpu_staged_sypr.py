"""GPU: SURVEY section 8 f4 -- the two-stage (symbolic / numeric) product mi_sparse_sp2m with pattern reuse, and the
symmetric triple products mi_sparse_sypr / mi_sparse_?_syprd (reference _sparse_sypr.py, dead upstream: the oracle
here is scipy / numpy on the same operands, structure bit-exact after ordering)."""
import numpy as np
import pytest
import scipy.sparse as sps

pytestmark = pytest.mark.gpu


def _tol_chained(dtype):
    """sypr / syprd chain TWO products; judged at the north_star's bars like every single product: fp64 1e-12, fp32 1e-5
    (measured on positive data from 200 to 18 000 products per entry: <= 6.8e-7, profiles/r05_sypr_fp32_error.log -- the
    4e-5 allowance of rounds 3-4 was never needed)."""
    return 1e-12 if np.dtype(dtype) == np.dtype(np.float64) else 1e-5


def _tol(dtype):
    return 1e-5 if np.dtype(dtype) == np.float32 else 1e-12


def _pos(m, n, density, dtype, seed):
    a = sps.random(m, n, density=density, format="csr", dtype=np.float64, random_state=seed)
    a.data[:] = np.random.default_rng(seed + 7).uniform(0.5, 1.5, a.nnz)
    return a.astype(dtype)


def _same(got, want, dtype):
    want = want.tocsr()
    want.sort_indices()
    got = got.tocsr()
    assert got.shape == want.shape and np.array_equal(got.indptr, want.indptr) and np.array_equal(got.indices, want.indices)
    assert np.allclose(got.data, want.data, rtol=_tol(dtype), atol=0)


@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.complex128])
def test_staged_product_pattern_reuse(gpu, dtype):
    """NNZ_COUNT once, FINALIZE_MULT for several value sets; hub rows so that the bitmap / range-partitioned path and
    its stored range table are reused too."""
    rng = np.random.default_rng(3)
    a = _pos(1500, 1200, 0.01, dtype, 1)
    hub = _pos(3, 1200, 0.6, dtype, 2)
    a = sps.vstack([a[:700], hub, a[700:]]).tocsr()
    b = _pos(1200, 9000, 0.01, dtype, 4)
    with gpu.StagedProduct(a, b, reorder_output=True) as p:
        nnz = p.count()
        ref = (a.astype(np.complex128 if np.dtype(dtype).kind == "c" else np.float64) @ b).tocsr()
        assert nnz == ref.nnz
        _same(p.finalize(), ref, dtype)
        for k in range(2):
            a2, b2 = a.copy(), b.copy()
            a2.data[:] = rng.uniform(0.5, 1.5, a.nnz).astype(dtype)
            b2.data[:] = rng.uniform(0.5, 1.5, b.nnz).astype(dtype)
            p.set_values(a=a2.data, b=b2.data if k else None)
            want = a2.astype(ref.dtype) @ (b2 if k else b).astype(ref.dtype)
            _same(p.finalize(), want, dtype)
        _same(p.full(), want, dtype)
    # transposed operands and a CSC input
    with gpu.StagedProduct(a, a.tocsc(), transpose_b=True, reorder_output=True) as p:
        _same(p.finalize(), a.astype(ref.dtype) @ a.astype(ref.dtype).T, dtype)


@pytest.mark.parametrize("hub", [False, True])
def test_staged_product_order_between_stages(gpu, hub):
    """mi_sparse_order(A) (or of B) between NNZ_COUNT and FINALIZE moves the entries the symbolic phase's per-entry tables
    were indexed by (max abs error 1.68 before the entry-order generation check).  Unsorted operands, ordered
    after the symbolic phase: the product must still be exact."""
    import ctypes as ct
    from sparse_dot_amd._mi_interface import MI, SparseHandle, matrix_descr, sparse_matrix_t
    rng = np.random.default_rng(11)

    def unsorted(m, seed):
        m = m.tocsr().copy()
        r = np.random.default_rng(seed)
        for i in range(m.shape[0]):  # shuffle the entries inside every row
            lo, hi = m.indptr[i], m.indptr[i + 1]
            perm = r.permutation(hi - lo)
            m.indices[lo:hi] = m.indices[lo:hi][perm]
            m.data[lo:hi] = m.data[lo:hi][perm]
        m.has_sorted_indices = False
        return m

    a = _pos(900, 700, 0.02, np.float64, 5)
    if hub:  # hub rows: the bitmap / range-partitioned path and its stored tables as well
        a = sps.vstack([a[:300], _pos(2, 700, 0.7, np.float64, 6), a[300:]]).tocsr()
    b = _pos(700, 6000 if hub else 800, 0.02, np.float64, 7)
    want = (a @ b).tocsr()
    for which in ("a", "b", "both"):
        ua, ub = unsorted(a, 1), unsorted(b, 2)
        with SparseHandle.from_scipy(ua) as ha, SparseHandle.from_scipy(ub) as hb:
            c = sparse_matrix_t()
            assert MI.call("mi_sparse_sp2m", 10, matrix_descr(), ha.ptr, 10, matrix_descr(), hb.ptr, 91, ct.byref(c)) == 0
            if which in ("a", "both"):
                assert MI.call("mi_sparse_order", ha.ptr) == 0
            if which in ("b", "both"):
                assert MI.call("mi_sparse_order", hb.ptr) == 0
            assert MI.call("mi_sparse_sp2m", 10, matrix_descr(), ha.ptr, 10, matrix_descr(), hb.ptr, 92, ct.byref(c)) == 0
            hc = SparseHandle(c, "d")
            hc.order()
            _same(hc.export("csr_matrix"), want, np.float64)
            hc.destroy()


def test_staged_product_errors(gpu):
    import ctypes as ct
    from sparse_dot_amd._mi_interface import MI, SparseHandle, matrix_descr, sparse_matrix_t
    a = _pos(50, 40, 0.1, np.float64, 1)
    b = _pos(40, 30, 0.1, np.float64, 2)
    with SparseHandle.from_scipy(a) as ha, SparseHandle.from_scipy(b) as hb, SparseHandle.from_scipy(a) as other:
        c = sparse_matrix_t()
        assert MI.call("mi_sparse_sp2m", 10, matrix_descr(), ha.ptr, 10, matrix_descr(), hb.ptr, 91, ct.byref(c)) == 0
        hc = SparseHandle(c, "d")
        # FINALIZE with other operands than NNZ_COUNT, a bad request, mismatched shapes
        assert MI.call("mi_sparse_sp2m", 10, matrix_descr(), other.ptr, 10, matrix_descr(), hb.ptr, 92, ct.byref(c)) == 3
        assert MI.call("mi_sparse_sp2m", 10, matrix_descr(), ha.ptr, 10, matrix_descr(), hb.ptr, 77, ct.byref(c)) == 3
        assert MI.call("mi_sparse_sp2m", 10, matrix_descr(), ha.ptr, 10, matrix_descr(), ha.ptr, 90, ct.byref(sparse_matrix_t())) == 3
        assert MI.call("mi_sparse_sp2m", 10, matrix_descr(), ha.ptr, 10, matrix_descr(), hb.ptr, 92, ct.byref(c)) == 0
        hc.destroy()
        # FINALIZE on a handle that never went through NNZ_COUNT
        plain = ha.ptr
        assert MI.call("mi_sparse_sp2m", 10, matrix_descr(), ha.ptr, 10, matrix_descr(), hb.ptr, 92, ct.byref(plain)) == 3


@pytest.mark.parametrize("transpose_a", [False, True])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_sypr_sparse_and_dense(gpu, dtype, transpose_a):
    x = _pos(300, 200, 0.05, dtype, 3)
    k = x.shape[0] if transpose_a else x.shape[1]
    s0 = _pos(k, k, 0.05, dtype, 4)
    sym = (s0 + s0.T).tocsr()
    bu = sps.triu(sym).tocsr()
    opx = (x.T if transpose_a else x).astype(np.float64)
    ref = np.triu((opx @ sym.astype(np.float64) @ opx.T).toarray())
    got = gpu.sparse_sypr(x, bu, transpose_a=transpose_a)
    assert isinstance(got, sps.csr_matrix) and got.dtype == dtype
    assert np.allclose(got.toarray(), ref, rtol=_tol_chained(dtype), atol=0)
    g2 = got.copy()
    g2.sort_indices()
    want = sps.csr_matrix(ref)
    assert np.array_equal(g2.indices, want.indices)  # no entry below the diagonal, none missing
    # entries of B below the diagonal must be ignored (descr: symmetric, upper)
    dirty = (bu + sps.tril(_pos(k, k, 0.05, dtype, 9), -1)).tocsr()
    assert np.allclose(gpu.sparse_sypr(x, dirty, transpose_a=transpose_a).toarray(), ref, rtol=_tol_chained(dtype), atol=0)
    # dense B (syprd): upper triangle referenced; out / scalars; both orders
    bd = np.triu(sym.toarray()).astype(dtype) + np.tril(np.full((k, k), 99.0, dtype=dtype), -1)
    for order in ("C", "F"):
        got = gpu.sparse_sypr(x, np.asarray(bd, order=order), transpose_a=transpose_a)
        assert got.dtype == dtype and np.allclose(np.triu(got), ref, rtol=_tol_chained(dtype), atol=0)
        out = np.asarray(np.ones(ref.shape, dtype=dtype), order=order)
        res = gpu.sparse_sypr(x, np.asarray(bd, order=order), transpose_a=transpose_a, out=out, out_scalar=2.0, scalar=3.0)
        assert res is out and np.allclose(np.triu(out), np.triu(3 * ref + 2), rtol=_tol_chained(dtype), atol=0)
        assert np.all(out[np.tril_indices(out.shape[0], -1)] == 1.0)  # strict lower triangle untouched
    with pytest.raises(ValueError):
        gpu.sparse_sypr(x.tocsc(), bu)
    with pytest.raises(ValueError):
        gpu.sparse_sypr(x.astype(np.complex128), bu.astype(np.complex128))


def test_staged_refinalize_drops_value_copies_of_the_result(gpu):
    """a FINALIZE re-run rewrites the values of an EXISTING result handle in place; what was derived from the
    old values (the column-partitioned SpMM plan's copy, the cached transpose) must go.  NNZ_COUNT, FINALIZE, products with C
    as the left operand until the partitioned plan is in use, set_values(A), FINALIZE, product again."""
    import ctypes as ct
    from sparse_dot_amd._mi_interface import MI, _check_return_value, matrix_descr
    a = _pos(400, 300, 0.05, np.float64, 21)
    a = sps.vstack([a[:100], _pos(4, 300, 0.9, np.float64, 22), a[100:]]).tocsr()  # long rows of C: the plan has something to split
    b = _pos(300, 2000, 0.05, np.float64, 23)
    x = np.random.default_rng(5).uniform(0.5, 1.5, (2000, 32))
    y = np.empty((a.shape[0], 32))

    def mm(hc, op=10):
        out = y if op == 10 else np.empty((2000, 32))
        rhs = x if op == 10 else np.ones((a.shape[0], 32))
        _check_return_value(MI.call("mi_sparse_d_mm", op, ct.c_double(1.0), hc.ptr, matrix_descr(), 101, rhs.ctypes.data,
                                    ct.c_int64(32), ct.c_int64(32), ct.c_double(0.0), out.ctypes.data, ct.c_int64(32)), "mm")
        return out.copy()

    opts = dict(spmm_kpart=2, spmm_kpart_min_row=16)
    for k, v in opts.items():
        gpu.mi_set_option(k, v)
    try:
        with gpu.StagedProduct(a, b) as p:
            p.count()
            p.finalize()
            want = (a @ b) @ x
            for _ in range(3):
                assert np.allclose(mm(p._hc), want, rtol=1e-12)
            assert gpu.mi_get_counter("spmm_last_kpart") == 8.0
            assert np.allclose(mm(p._hc, 11), (a @ b).T @ np.ones((a.shape[0], 32)), rtol=1e-12)  # caches the transpose
            a2 = a.copy()
            a2.data[:] = np.random.default_rng(6).uniform(0.5, 1.5, a.nnz)
            p.set_values(a=a2.data)
            p.finalize()
            assert np.allclose(mm(p._hc), (a2 @ b) @ x, rtol=1e-12)
            assert np.allclose(mm(p._hc, 11), (a2 @ b).T @ np.ones((a.shape[0], 32)), rtol=1e-12)
    finally:
        gpu.mi_set_option("spmm_kpart", 1)
        gpu.mi_set_option("spmm_kpart_min_row", 64)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_staged_product_wide_b_by_column_panels(gpu, dtype):
    """mi_sparse_sp2m with a B of 2^22 + 77 columns and hub rows in the product: the staged product takes the column panels of
    the one-shot product (round 6; the global-memory hash before), the panels are kept on the result handle between the stages,
    and a FINALIZE after set_values(B) refills their value copies; mi_sparse_order(A) between the stages rebuilds the per-entry
    extents of every panel.  Oracle: scipy in fp64."""
    import ctypes as ct
    from sparse_dot_amd._mi_interface import MI
    rng = np.random.default_rng(15)
    n, k, W = (1 << 22) + 77, 2000, 1 << 20
    cols = []
    for r in range(k):
        c = rng.integers(0, n, 24)
        if r % 3 == 0:
            c = np.concatenate([c, [W - 1, W, 2 * W - 1, 2 * W, 3 * W, 4 * W - 1, 4 * W, n - 1]])
        cols.append(np.unique(c))
    ptr = np.concatenate([[0], np.cumsum([c.size for c in cols])])
    ind = np.concatenate(cols).astype(np.int32)
    b = sps.csr_matrix((rng.uniform(0.5, 1.5, ind.size), ind, ptr), shape=(k, n)).astype(dtype)
    a = sps.random(50, k, density=0.004, format="lil", random_state=3, dtype=np.float64)
    a[4, rng.choice(k, 1100, replace=False)] = 0.75   # hub rows of A
    a[31, rng.choice(k, 1800, replace=False)] = 1.5
    a = a.tocsr()
    a.data[:] = rng.uniform(0.5, 1.5, a.nnz)
    a = a.astype(dtype)
    gpu.mi_get_counter("reset")
    with gpu.StagedProduct(a, b, reorder_output=True) as p:
        nnz = p.count()
        assert gpu.mi_get_counter("spgemm_panels") == 0.0  # counted by the numeric phase
        ref = (a.astype(np.float64) @ b.astype(np.float64)).tocsr()
        assert nnz == ref.nnz
        _same(p.finalize(), ref, dtype)
        assert gpu.mi_get_counter("spgemm_panels") == 5.0
        a2, b2 = a.copy(), b.copy()
        a2.data[:] = rng.uniform(0.5, 1.5, a.nnz).astype(dtype)
        b2.data[:] = rng.uniform(0.5, 1.5, b.nnz).astype(dtype)
        p.set_values(a=a2.data, b=b2.data)
        _same(p.finalize(), a2.astype(np.float64) @ b2.astype(np.float64), dtype)
        _same(p.full(), a2.astype(np.float64) @ b2.astype(np.float64), dtype)
    # A with unsorted rows, ordered between the stages
    ua = a.copy()
    for i in range(ua.shape[0]):
        lo, hi = ua.indptr[i], ua.indptr[i + 1]
        o = rng.permutation(hi - lo)
        ua.indices[lo:hi], ua.data[lo:hi] = ua.indices[lo:hi][o], ua.data[lo:hi][o]
    ua.has_sorted_indices = False
    with gpu.StagedProduct(ua, b, reorder_output=True) as p:
        p.count()
        assert MI.call("mi_sparse_order", p._ha.ptr) == 0
        _same(p.finalize(), a.astype(np.float64) @ b.astype(np.float64), dtype)
