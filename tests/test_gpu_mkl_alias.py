"""GPU: the MKL-NAMED boundary on real hardware.

`sparse_dot_amd/libmi_mkl_rt.so` exports the symbols the reference binds (`mkl_sparse_?_create_csr`, `mkl_sparse_?_mm`,
`mkl_sparse_spmm`, `mkl_sparse_order`, `mkl_sparse_?_export_csr`, `mkl_sparse_?_syrkd`, `MKL_Set_Interface_Layer` ...;
reference sparse_dot_mkl/_mkl_interface/_cfunctions.py:43-168, 376-382, 526-649) with MKL's argument conventions.  The
reference itself cannot travel to the GPU box, so the library is driven here by the build's own MKL-name binding
(oracle/mkl_shim.py -- the very binding bench.py uses to time the real MKL) under both interface layers, and its answers
are compared with the oracle."""
import os

import numpy as np
import pytest
import scipy.sparse as sps

pytestmark = pytest.mark.gpu

ALIAS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sparse_dot_amd", "libmi_mkl_rt.so")


def _pos(m, n, density, dtype, seed):
    a = sps.random(m, n, density=density, format="csr", dtype=np.float64, random_state=seed)
    a.data[:] = np.random.default_rng(seed + 1).uniform(0.5, 1.5, a.nnz)
    a.sort_indices()
    return a.astype(dtype)


@pytest.mark.parametrize("interface", [0, 1], ids=["LP64", "ILP64"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_mkl_named_library_answers_like_mkl(gpu, oracle, interface, dtype):
    from oracle import mkl_shim
    assert os.path.exists(ALIAS), "build sparse_dot_amd/libmi_mkl_rt.so first (make -C sparse_dot_amd/csrc)"
    mk = mkl_shim.MklSpmm(path=ALIAS, interface=interface)
    assert "mi_sparse" in mk.version()
    tol = 1e-5 if dtype == np.float32 else 1e-12
    a = _pos(900, 700, 0.02, dtype, 1)
    b = _pos(700, 500, 0.02, dtype, 2)
    dense = np.random.default_rng(3).uniform(0.5, 1.5, (700, 96)).astype(dtype)
    ha, hb = mk.make(a), mk.make(b)
    try:
        # mkl_sparse_?_mm
        out = np.zeros((900, 96), dtype=dtype)
        mk.mm(ha, dense, out)
        want = oracle.spmm(a.astype(np.float64), dense.astype(np.float64))
        assert np.max(np.abs(out - want) / np.abs(want).clip(1e-30)) <= tol
        # mkl_sparse_spmm + mkl_sparse_order + mkl_sparse_?_export_csr
        hc = mk.spmm_handle(ha, hb)
        try:
            assert mk.order(hc) == 0
            indptr, indices, data, shape = mk.export_csr(hc)
        finally:
            mk.destroy(hc)
        wc = oracle.spgemm(a.astype(np.float64), b.astype(np.float64))
        assert shape == (900, 500)
        assert indptr.dtype == (np.int64 if interface else np.int32) and indices.dtype == indptr.dtype
        assert np.array_equal(indptr, wc.indptr) and np.array_equal(indices, wc.indices)  # bit-exact structure
        assert data.dtype == dtype
        np.testing.assert_allclose(data, wc.data, rtol=tol, atol=0)
        # mkl_sparse_?_syrkd (A^T A, upper triangle, row-major)
        g = np.zeros((700, 700), dtype=dtype)
        mk.syrkd(ha, g)
        wg = oracle.syrkd(a.astype(np.float64))
        iu = np.triu_indices(700)
        np.testing.assert_allclose(g[iu], wg[iu], rtol=10 * tol, atol=tol)
    finally:
        mk.destroy(ha)
        mk.destroy(hb)
        mk.lib.MKL_Set_Interface_Layer(0)


def test_mkl_named_library_status_codes(gpu):
    """NULL handles come back as SPARSE_STATUS_NOT_INITIALIZED (1), as the reference's tests expect of MKL
    (reference tests/test_mkl.py:128-141)."""
    import ctypes as ct
    lib = ct.CDLL(ALIAS)
    lib.mkl_sparse_destroy.restype = ct.c_int
    assert lib.mkl_sparse_order(ct.c_void_p(0)) == 1
    c = ct.c_void_p()
    assert lib.mkl_sparse_spmm(ct.c_int(10), ct.c_void_p(0), ct.c_void_p(0), ct.byref(c)) == 1
