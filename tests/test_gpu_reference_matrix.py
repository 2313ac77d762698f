"""GPU: the behaviour matrix the reference pins with its class-inheritance test suites
({C, F} x {full, 1 x n, n x 1} x {csr_matrix, csr_array, csc, bsr} x {real, complex}
x {sparse on the left, on the right} x {no out, out = ones with out_scalar = 3}), restated as one
parametrised test against numpy on densified operands.  Results must have numpy's shape, the
dispatcher's dtype / memory order, and `out` must come back as the same object."""
import zlib

import numpy as np
import pytest
import scipy.sparse as sps

pytestmark = pytest.mark.gpu

SPARSE_MAKERS = {
    "csr_matrix": sps.csr_matrix,
    "csr_array": sps.csr_array,
    "csc_matrix": sps.csc_matrix,
    "bsr_matrix": lambda m: sps.bsr_matrix(m, blocksize=(1, 1)),
}
SHAPES = {"full": (40, 60, 25), "one_row": (1, 60, 25), "one_col": (40, 60, 1), "inner_one": (40, 1, 25)}


def _mk(rng, shape, dtype, density=0.3):
    x = rng.uniform(0.5, 1.5, shape) * (rng.random(shape) < density)
    if np.dtype(dtype).kind == "c":
        x = x + 1j * rng.uniform(0.5, 1.5, shape) * (x != 0)
    return x.astype(dtype)


def _tol(dtype):
    return 1e-5 if np.dtype(dtype) in (np.dtype(np.float32), np.dtype(np.complex64)) else 1e-12  # the north_star's bars


@pytest.mark.parametrize("use_out", [False, True])
@pytest.mark.parametrize("side", ["sparse_left", "sparse_right"])
@pytest.mark.parametrize("shape", list(SHAPES))
@pytest.mark.parametrize("cls", list(SPARSE_MAKERS))
@pytest.mark.parametrize("order", ["C", "F"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.complex64, np.complex128])
def test_sparse_dense_matrix(gpu, dtype, order, cls, shape, side, use_out):
    m, k, n = SHAPES[shape]
    rng = np.random.default_rng(zlib.crc32(repr((shape, side, cls)).encode()))  # stable across processes (hash() is salted)
    a_d, b_d = _mk(rng, (m, k), dtype), _mk(rng, (k, n), dtype, density=1.0)
    if side == "sparse_left":
        a = SPARSE_MAKERS[cls](a_d)
        b = np.asarray(b_d, order=order)
        want = a_d.astype(np.complex128 if np.dtype(dtype).kind == "c" else np.float64) @ b_d
    else:
        a = np.asarray(_mk(rng, (n, m), dtype, density=1.0), order=order)   # dense (n x m) @ sparse (m x k)
        b = SPARSE_MAKERS[cls](a_d)
        want = a.astype(np.complex128 if np.dtype(dtype).kind == "c" else np.float64) @ a_d
    kw = {}
    if use_out:
        # the out array must have the order the product would have: for an ambiguous (vector-like)
        # dense operand the dispatcher takes the order FROM out, so either is accepted
        kw["out"] = np.asarray(np.ones(want.shape, dtype=dtype), order=order)
        kw["out_scalar"] = 3.0
        want = want + 3.0
    got = gpu.dot_product_mkl(a, b, **kw)
    assert got.shape == want.shape and got.dtype == dtype
    if use_out:
        assert got is kw["out"]
    np.testing.assert_allclose(got, want, rtol=_tol(dtype), atol=_tol(dtype))


@pytest.mark.parametrize("vec_shape", ["1d", "col", "row"])
@pytest.mark.parametrize("cls", ["csr_matrix", "csc_matrix", "csr_array"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.complex128])
def test_sparse_vector_matrix(gpu, dtype, cls, vec_shape):
    """SpMV dispatch: the result takes the vector's shape convention ((n,), (n, 1) or (1, n))."""
    rng = np.random.default_rng(5)
    a_d = _mk(rng, (30, 50), dtype)
    a = SPARSE_MAKERS[cls](a_d)
    wide = np.complex128 if np.dtype(dtype).kind == "c" else np.float64
    if vec_shape == "row":    # (1, 30) @ sparse (30, 50)
        v = _mk(rng, (1, 30), dtype, 1.0)
        got = gpu.dot_product_mkl(v, a)
        want = v.astype(wide) @ a_d
    else:
        v = _mk(rng, (50,) if vec_shape == "1d" else (50, 1), dtype, 1.0)
        got = gpu.dot_product_mkl(a, v)
        want = a_d.astype(wide) @ v
        u = _mk(rng, (30,), dtype, 1.0)
        got2 = gpu.dot_product_mkl(u, a)       # 1-d vector on the left
        assert got2.shape == (50,)
        np.testing.assert_allclose(got2, u.astype(wide) @ a_d, rtol=_tol(dtype), atol=_tol(dtype))
    assert got.shape == want.shape and got.dtype == dtype
    np.testing.assert_allclose(got, want, rtol=_tol(dtype), atol=_tol(dtype))


@pytest.mark.parametrize("dense", [False, True])
@pytest.mark.parametrize("cls_b", ["csr_matrix", "csc_matrix"])
@pytest.mark.parametrize("cls_a", ["csr_matrix", "csr_array", "csc_matrix"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.complex64, np.complex128])
def test_sparse_sparse_matrix(gpu, dtype, cls_a, cls_b, dense):
    rng = np.random.default_rng(9)
    a_d, b_d = _mk(rng, (35, 45), dtype), _mk(rng, (45, 20), dtype)
    a, b = SPARSE_MAKERS[cls_a](a_d), SPARSE_MAKERS[cls_b](b_d)
    wide = np.complex128 if np.dtype(dtype).kind == "c" else np.float64
    want = a_d.astype(wide) @ b_d
    got = gpu.dot_product_mkl(a, b, dense=dense)
    if dense:
        assert isinstance(got, np.ndarray) and got.dtype == dtype and got.flags.c_contiguous
        np.testing.assert_allclose(got, want, rtol=_tol(dtype), atol=_tol(dtype))
    else:
        assert type(got) is type(a) and got.dtype == dtype
        np.testing.assert_allclose(got.toarray(), want, rtol=_tol(dtype), atol=_tol(dtype))
        srt = gpu.dot_product_mkl(a, b, reorder_output=True)
        assert srt.has_sorted_indices or np.all(np.diff(srt.indptr) <= 1) or _sorted(srt)


def _sorted(m):
    return all(np.all(np.diff(m.indices[m.indptr[i]:m.indptr[i + 1]]) > 0) for i in range(len(m.indptr) - 1))


def test_cast_rules_on_device(gpu):
    rng = np.random.default_rng(3)
    a_d = _mk(rng, (20, 30), np.float64)
    a32, b64 = sps.csr_matrix(a_d.astype(np.float32)), _mk(rng, (30, 7), np.float64, 1.0)
    with pytest.raises(ValueError):
        gpu.dot_product_mkl(a32, b64)
    r = gpu.dot_product_mkl(a32, b64, cast=True)
    assert r.dtype == np.float64
    np.testing.assert_allclose(r, a32.toarray().astype(np.float64) @ b64, rtol=1e-12)
    ai = sps.csr_matrix((a_d * 10).astype(np.int64))
    r = gpu.dot_product_mkl(ai, b64, cast=True)
    np.testing.assert_allclose(r, ai.toarray().astype(np.float64) @ b64, rtol=1e-12)
    c = sps.csr_matrix(a_d.astype(np.complex64))
    r = gpu.dot_product_mkl(c, b64, cast=True)          # real operand follows the complex one
    assert r.dtype == np.complex64
    r = gpu.dot_product_mkl(a32, a32.T.tocsr().astype(np.float64), cast=True)
    assert r.dtype == np.float64 and sps.issparse(r)
    # inputs untouched by the casts
    assert a32.dtype == np.float32 and ai.dtype == np.int64
