"""
ctypes front end of the CPU oracle (oracle/liboracle.so, built from sparse_oracle.c).

TEST INFRASTRUCTURE ONLY -- may be imported by tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py, never by the product package `sparse_dot_amd`.

The functions take scipy.sparse / numpy objects and return numpy / scipy objects; they mirror
the arithmetic of the MKL routines the reference calls (see oracle_kernels.inc for the
reference file:line of each).  Parity status: pinned by tests/test_oracle_golden.py against
tests/golden/ (captured from the reference on MKL) -- see sparse_oracle.c.
"""
import ctypes as _ct
import os as _os
import subprocess as _sp

import numpy as _np
import scipy.sparse as _sps

_HERE = _os.path.dirname(_os.path.abspath(__file__))
_LIB_PATH = _os.path.join(_HERE, "liboracle.so")

OP_N, OP_T, OP_H = 10, 11, 12
LAYOUT_C, LAYOUT_F = 101, 102
CBLAS_N, CBLAS_T, CBLAS_H = 111, 112, 113
UPPER, LOWER = 121, 122


def build(force=False):
    """Compile liboracle.so with gcc (seconds)."""
    src = [_os.path.join(_HERE, f) for f in ("sparse_oracle.c", "oracle_kernels.inc")]
    if (
        not force
        and _os.path.exists(_LIB_PATH)
        and all(_os.path.getmtime(_LIB_PATH) >= _os.path.getmtime(s) for s in src if _os.path.exists(s))
    ):
        return _LIB_PATH
    _sp.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=_sp.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = _ct.CDLL(_LIB_PATH)
    return _lib


class _C8(_ct.Structure):
    _fields_ = [("re", _ct.c_float), ("im", _ct.c_float)]


class _C16(_ct.Structure):
    _fields_ = [("re", _ct.c_double), ("im", _ct.c_double)]


_PREFIX = {
    _np.dtype(_np.float32): "s",
    _np.dtype(_np.float64): "d",
    _np.dtype(_np.complex64): "c",
    _np.dtype(_np.complex128): "z",
}


def _fn(name, dtype):
    return getattr(lib(), "orc_%s_%s" % (_PREFIX[_np.dtype(dtype)], name))


def _i64(a):
    return _np.ascontiguousarray(a, dtype=_np.int64)


def _p(a):
    return a.ctypes.data_as(_ct.c_void_p)


def _csr_parts(a):
    """(rows_start, rows_end, col, val) of a scipy CSR matrix as int64 / native value arrays."""
    ptr = _i64(a.indptr)
    return ptr[:-1].copy(), ptr[1:].copy(), _i64(a.indices), _np.ascontiguousarray(a.data)


def _is_complex(dtype):
    return _np.dtype(dtype).kind == "c"


def _as_csr(a):
    """CSR view of CSR / CSC / BSR input (conversion by the oracle's own routines)."""
    if _sps.isspmatrix_csr(a) or isinstance(a, _sps.csr_array):
        return a
    if a.format == "csc":
        return csc_to_csr(a)
    if a.format == "bsr":
        return bsr_to_csr(a)
    raise ValueError("unsupported sparse format %r" % a.format)


# --------------------------------------------------------------------------------------------
# SpMM
# --------------------------------------------------------------------------------------------
def spmm(a, b, alpha=1.0, beta=0.0, c=None, op=OP_N):
    """alpha * op(a) @ b + beta * c with a sparse (CSR/CSC/BSR), b dense C- or F-contiguous.

    Complex alpha/beta are folded in numpy (the C routine is called with alpha=1, beta=0 for
    complex dtypes and the scaling applied outside), so no complex-by-value ABI is needed.
    """
    a = _as_csr(a)
    dt = _np.dtype(a.dtype)
    b = _np.asarray(b)
    assert b.dtype == dt, (b.dtype, dt)
    if b.flags.c_contiguous:
        layout, ldb = LAYOUT_C, b.shape[1]
    elif b.flags.f_contiguous:
        layout, ldb = LAYOUT_F, b.shape[0]
    else:
        raise ValueError("b must be contiguous")
    m, k = a.shape
    n = b.shape[1]
    crows = m if op == OP_N else k
    order = "C" if layout == LAYOUT_C else "F"
    out = _np.zeros((crows, n), dtype=dt, order=order)
    ldc = n if layout == LAYOUT_C else crows
    rs, re, col, val = _csr_parts(a)
    if _is_complex(dt):
        # `T _Complex` by value has the ABI of a struct of two T (x86-64 SysV): pass it as one
        f = _fn("csr_mm", dt)
        f.restype = _ct.c_int
        cplx = _C8 if dt == _np.complex64 else _C16
        f.argtypes = [_ct.c_int, cplx, _ct.c_int64, _ct.c_int64] + [_ct.c_void_p] * 4 + [
            _ct.c_int, _ct.c_void_p, _ct.c_int64, _ct.c_int64, cplx, _ct.c_void_p, _ct.c_int64]
        st = f(op, cplx(1.0, 0.0), m, k, _p(rs), _p(re), _p(col), _p(val), layout, _p(b), n, ldb,
               cplx(0.0, 0.0), _p(out), ldc)
        if st:
            raise ValueError("oracle csr_mm returned %d" % st)
        out *= dt.type(alpha)
        if c is not None and beta != 0:
            out += dt.type(beta) * c
        return out
    f = _fn("csr_mm", dt)
    ct = _ct.c_float if dt == _np.float32 else _ct.c_double
    f.restype = _ct.c_int
    f.argtypes = [_ct.c_int, ct, _ct.c_int64, _ct.c_int64] + [_ct.c_void_p] * 4 + [
        _ct.c_int, _ct.c_void_p, _ct.c_int64, _ct.c_int64, ct, _ct.c_void_p, _ct.c_int64]
    if c is not None and beta != 0:
        out[...] = c
    st = f(op, alpha, m, k, _p(rs), _p(re), _p(col), _p(val), layout, _p(b), n, ldb,
           beta if c is not None else 0.0, _p(out), ldc)
    if st:
        raise ValueError("oracle csr_mm returned %d" % st)
    return out


def spmv(a, x, alpha=1.0, beta=0.0, y=None, op=OP_N):
    """alpha * op(a) @ x + beta * y for a 1-D x through the oracle's own `csr_mv` entry point
    (restates mkl_sparse_?_mv, reference _sparse_vector.py:87-95).  Returns a new 1-D array."""
    a = _as_csr(a)
    dt = _np.dtype(a.dtype)
    x = _np.ascontiguousarray(_np.asarray(x).ravel())
    assert x.dtype == dt, (x.dtype, dt)
    m, k = a.shape
    n_out = m if op == OP_N else k
    assert x.size == (k if op == OP_N else m)
    out = _np.zeros(n_out, dtype=dt)
    rs, re, col, val = _csr_parts(a)
    f = _fn("csr_mv", dt)
    f.restype = _ct.c_int
    if _is_complex(dt):
        cplx = _C8 if dt == _np.complex64 else _C16
        f.argtypes = [_ct.c_int, cplx, _ct.c_int64, _ct.c_int64] + [_ct.c_void_p] * 5 + [cplx, _ct.c_void_p]
        st = f(op, cplx(1.0, 0.0), m, k, _p(rs), _p(re), _p(col), _p(val), _p(x), cplx(0.0, 0.0), _p(out))
        if st:
            raise ValueError("oracle csr_mv returned %d" % st)
        out *= dt.type(alpha)
        if y is not None and beta != 0:
            out += dt.type(beta) * _np.asarray(y).ravel()
        return out
    ct = _ct.c_float if dt == _np.float32 else _ct.c_double
    f.argtypes = [_ct.c_int, ct, _ct.c_int64, _ct.c_int64] + [_ct.c_void_p] * 5 + [ct, _ct.c_void_p]
    if y is not None and beta != 0:
        out[...] = _np.asarray(y).ravel()
    st = f(op, alpha, m, k, _p(rs), _p(re), _p(col), _p(val), _p(x), beta if y is not None else 0.0, _p(out))
    if st:
        raise ValueError("oracle csr_mv returned %d" % st)
    return out


# --------------------------------------------------------------------------------------------
# SpGEMM
# --------------------------------------------------------------------------------------------
def _spgemm_csr(a, b, upper=False, prune=False, conj_a=False):
    dt = _np.dtype(a.dtype)
    assert b.dtype == dt
    m, k = a.shape
    n = b.shape[1]
    ars, are, acol, aval = _csr_parts(a)
    brs, bre, bcol, bval = _csr_parts(b)
    cnt = _np.zeros(m, dtype=_np.int64)
    f = _fn("csr_spgemm_count", dt)
    f.restype = _ct.c_int
    f.argtypes = [_ct.c_int64] * 3 + [_ct.c_void_p] * 6 + [_ct.c_int, _ct.c_void_p]
    st = f(m, k, n, _p(ars), _p(are), _p(acol), _p(brs), _p(bre), _p(bcol), int(upper), _p(cnt))
    if st:
        raise ValueError("oracle spgemm_count returned %d" % st)
    cptr = _np.zeros(m + 1, dtype=_np.int64)
    _np.cumsum(cnt, out=cptr[1:])
    nnz = int(cptr[-1])
    ccol = _np.zeros(max(nnz, 1), dtype=_np.int64)
    cval = _np.zeros(max(nnz, 1), dtype=dt)
    out_nnz = _np.zeros(m, dtype=_np.int64)
    f = _fn("csr_spgemm_fill", dt)
    f.restype = _ct.c_int
    f.argtypes = [_ct.c_int64] * 3 + [_ct.c_void_p] * 8 + [_ct.c_int] * 3 + [_ct.c_void_p] * 4
    st = f(m, k, n, _p(ars), _p(are), _p(acol), _p(aval), _p(brs), _p(bre), _p(bcol), _p(bval),
           int(upper), int(prune), int(conj_a), _p(cptr), _p(ccol), _p(cval), _p(out_nnz))
    if st:
        raise ValueError("oracle spgemm_fill returned %d" % st)
    if prune:
        # compact rows
        keep = _np.zeros(nnz, dtype=bool)
        for i in range(m):
            keep[cptr[i]:cptr[i] + out_nnz[i]] = True
        ccol, cval = ccol[:nnz][keep], cval[:nnz][keep]
        cptr = _np.zeros(m + 1, dtype=_np.int64)
        _np.cumsum(out_nnz, out=cptr[1:])
        nnz = int(cptr[-1])
    return _sps.csr_matrix((cval[:nnz], ccol[:nnz], cptr), shape=(m, n))


def spgemm(a, b, prune=False):
    """Sparse a @ b, canonical (sorted) CSR; explicit zeros kept unless prune (scipy) is set."""
    return _spgemm_csr(_as_csr(a), _as_csr(b), prune=prune)


def spmmd(a, b, order="C"):
    """Dense a @ b from two sparse operands."""
    a, b = _as_csr(a), _as_csr(b)
    dt = _np.dtype(a.dtype)
    m, k = a.shape
    n = b.shape[1]
    out = _np.zeros((m, n), dtype=dt, order=order)
    ars, are, acol, aval = _csr_parts(a)
    brs, bre, bcol, bval = _csr_parts(b)
    f = _fn("csr_spmmd", dt)
    f.restype = _ct.c_int
    f.argtypes = [_ct.c_int64] * 3 + [_ct.c_void_p] * 8 + [_ct.c_int, _ct.c_void_p, _ct.c_int64]
    st = f(m, k, n, _p(ars), _p(are), _p(acol), _p(aval), _p(brs), _p(bre), _p(bcol), _p(bval),
           LAYOUT_C if order == "C" else LAYOUT_F, _p(out), n if order == "C" else m)
    if st:
        raise ValueError("oracle spmmd returned %d" % st)
    return out


# --------------------------------------------------------------------------------------------
# format helpers
# --------------------------------------------------------------------------------------------
def transpose(a, conj=False):
    """CSR of a^T (stable)."""
    a = _as_csr(a)
    dt = _np.dtype(a.dtype)
    m, k = a.shape
    rs, re, col, val = _csr_parts(a)
    tptr = _np.zeros(k + 1, dtype=_np.int64)
    tcol = _np.zeros(max(a.nnz, 1), dtype=_np.int64)
    tval = _np.zeros(max(a.nnz, 1), dtype=dt)
    f = _fn("csr_transpose", dt)
    f.restype = _ct.c_int
    f.argtypes = [_ct.c_int64] * 2 + [_ct.c_void_p] * 4 + [_ct.c_int] + [_ct.c_void_p] * 3
    st = f(m, k, _p(rs), _p(re), _p(col), _p(val), int(conj), _p(tptr), _p(tcol), _p(tval))
    if st:
        raise ValueError("oracle transpose returned %d" % st)
    return _sps.csr_matrix((tval[:a.nnz], tcol[:a.nnz], tptr), shape=(k, m))


def csc_to_csr(a):
    """CSC -> CSR: the CSC arrays are the CSR arrays of a^T; transpose that."""
    at = _sps.csr_matrix((a.data, a.indices, a.indptr), shape=(a.shape[1], a.shape[0]))
    return transpose(at)


def bsr_to_csr(a):
    dt = _np.dtype(a.dtype)
    bs = a.blocksize[0]
    assert bs == a.blocksize[1]
    brows = a.shape[0] // bs
    ptr = _i64(a.indptr)
    rs, re = ptr[:-1].copy(), ptr[1:].copy()
    bcol = _i64(a.indices)
    data = _np.ascontiguousarray(a.data)
    nnz = data.size
    cptr = _np.zeros(a.shape[0] + 1, dtype=_np.int64)
    ccol = _np.zeros(max(nnz, 1), dtype=_np.int64)
    cval = _np.zeros(max(nnz, 1), dtype=dt)
    f = _fn("bsr_to_csr", dt)
    f.restype = _ct.c_int
    f.argtypes = [_ct.c_int64, _ct.c_int64, _ct.c_int] + [_ct.c_void_p] * 7
    st = f(brows, bs, LAYOUT_C, _p(rs), _p(re), _p(bcol), _p(data), _p(cptr), _p(ccol), _p(cval))
    if st:
        raise ValueError("oracle bsr_to_csr returned %d" % st)
    return _sps.csr_matrix((cval[:nnz], ccol[:nnz], cptr), shape=a.shape)


def order(a):
    """Copy of CSR `a` with sorted column indices (stable)."""
    a = _as_csr(a).copy()
    dt = _np.dtype(a.dtype)
    rs, re, col, val = _csr_parts(a)
    val = val.copy()
    f = _fn("csr_order", dt)
    f.restype = _ct.c_int
    f.argtypes = [_ct.c_int64] + [_ct.c_void_p] * 4
    st = f(a.shape[0], _p(rs), _p(re), _p(col), _p(val))
    if st:
        raise ValueError("oracle order returned %d" % st)
    return _sps.csr_matrix((val, col, _i64(a.indptr)), shape=a.shape)


# --------------------------------------------------------------------------------------------
# Gram
# --------------------------------------------------------------------------------------------
def syrk_sparse(a, aat=False):
    """Upper-triangular sparse CSR of a^T a (default) or a a^T (aat=True)."""
    a = _as_csr(a)
    at = transpose(a)
    return _spgemm_csr(a, at, upper=True) if aat else _spgemm_csr(at, a, upper=True)


def syrkd(a, aat=False, alpha=1.0, beta=0.0, c=None):
    """Dense upper triangle of alpha * a^T a (or a a^T) + beta * c; strict lower = c's (or 0)."""
    a = _as_csr(a)
    dt = _np.dtype(a.dtype)
    m, k = a.shape
    n = m if aat else k
    out = _np.zeros((n, n), dtype=dt) if c is None else _np.array(c, dtype=dt, order="C")
    rs, re, col, val = _csr_parts(a)
    f = _fn("csr_syrkd", dt)
    ct = _ct.c_float if dt == _np.float32 else _ct.c_double
    f.restype = _ct.c_int
    f.argtypes = [_ct.c_int, _ct.c_int64, _ct.c_int64] + [_ct.c_void_p] * 4 + [
        ct, ct, _ct.c_void_p, _ct.c_int, _ct.c_int64]
    st = f(OP_N if aat else OP_T, m, k, _p(rs), _p(re), _p(col), _p(val), alpha,
           beta if c is not None else 0.0, _p(out), LAYOUT_C, n)
    if st:
        raise ValueError("oracle syrkd returned %d" % st)
    return out


# --------------------------------------------------------------------------------------------
# dense
# --------------------------------------------------------------------------------------------
def _layout_ld(x):
    if x.flags.c_contiguous:
        return LAYOUT_C, x.shape[1]
    if x.flags.f_contiguous:
        return LAYOUT_F, x.shape[0]
    raise ValueError("array must be contiguous")


def gemm(a, b, alpha=1.0, beta=0.0, c=None):
    """alpha * a @ b + beta * c for real dense arrays; output order follows `a`."""
    dt = _np.dtype(a.dtype)
    assert not _is_complex(dt)
    la, lda = _layout_ld(a)
    lb, ldb = _layout_ld(b)
    tb = CBLAS_N if lb == la else CBLAS_T
    m, k = a.shape
    n = b.shape[1]
    order_ = "C" if la == LAYOUT_C else "F"
    out = _np.zeros((m, n), dtype=dt, order=order_) if c is None else _np.array(c, dtype=dt, order=order_)
    ldc = n if la == LAYOUT_C else m
    f = _fn("gemm", dt)
    ct = _ct.c_float if dt == _np.float32 else _ct.c_double
    f.restype = _ct.c_int
    f.argtypes = [_ct.c_int] * 3 + [_ct.c_int64] * 3 + [ct, _ct.c_void_p, _ct.c_int64, _ct.c_void_p,
                                                     _ct.c_int64, ct, _ct.c_void_p, _ct.c_int64]
    st = f(la, CBLAS_N, tb, m, n, k, alpha, _p(a), lda, _p(b), ldb,
           beta if c is not None else 0.0, _p(out), ldc)
    if st:
        raise ValueError("oracle gemm returned %d" % st)
    return out


def syrk(a, aat=False, alpha=1.0, beta=0.0, c=None):
    """Upper triangle of alpha * a^T a (or a a^T) + beta * c for a real dense array."""
    dt = _np.dtype(a.dtype)
    la, lda = _layout_ld(a)
    n, k = a.shape if aat else a.shape[::-1]
    order_ = "C" if la == LAYOUT_C else "F"
    out = _np.zeros((n, n), dtype=dt, order=order_) if c is None else _np.array(c, dtype=dt, order=order_)
    f = _fn("syrk", dt)
    ct = _ct.c_float if dt == _np.float32 else _ct.c_double
    f.restype = _ct.c_int
    f.argtypes = [_ct.c_int] * 3 + [_ct.c_int64] * 2 + [ct, _ct.c_void_p, _ct.c_int64, ct,
                                                     _ct.c_void_p, _ct.c_int64]
    st = f(la, UPPER, CBLAS_N if aat else CBLAS_T, n, k, alpha, _p(a), lda,
           beta if c is not None else 0.0, _p(out), n)
    if st:
        raise ValueError("oracle syrk returned %d" % st)
    return out


# --------------------------------------------------------------------------------------------
# API-level restatement (what the reference's dispatcher computes, minus validation)
# --------------------------------------------------------------------------------------------
def _common_dtype(a, b, cast):
    """dtype rule of the reference's _type_check (_mkl_interface/_common.py:773-866)."""
    valid = [_np.dtype(t) for t in (_np.float32, _np.float64, _np.complex64, _np.complex128)]
    da, db = _np.dtype(a.dtype), _np.dtype(b.dtype)
    if da == db and da in valid:
        return da
    if not cast:
        raise ValueError("dtype mismatch without cast")
    ncx = (da.kind == "c") + (db.kind == "c")
    if ncx == 0:
        return _np.dtype(_np.float64)
    if ncx == 1:
        cx = da if da.kind == "c" else db
        return cx if cx in valid else _np.dtype(_np.complex128)
    return _np.dtype(_np.complex128)


def dot_product(a, b, cast=False, reorder_output=False, dense=False, out=None, out_scalar=None):
    """Oracle restatement of dot_product_mkl for 2-D operands
    (reference sparse_dot_mkl/sparse_dot.py:18-152; empty-input shortcut _common.py:1003-1024)."""
    sa, sb = _sps.issparse(a), _sps.issparse(b)
    beta = 1.0 if out_scalar is None else out_scalar
    empty = min(*a.shape, *b.shape) == 0 or (sa and a.nnz == 0 and a.data.size == 0) or (
        sb and b.nnz == 0 and b.data.size == 0)
    if (sa != sb) and empty and (a.ndim == 1 or b.ndim == 1):  # vector product of an empty operand: zeros of the vector's shape convention
        if out is not None:
            return out
        dt0 = _np.float32 if (a.dtype == b.dtype and a.dtype == _np.float32) else _np.float64
        return _np.zeros((b.shape[1],) if a.ndim == 1 else (a.shape[0],), dtype=dt0)
    if sa and sb:
        if empty:
            if dense:
                return out if out is not None else _np.zeros((a.shape[0], b.shape[1]), dtype=a.dtype)
            return type(a)((a.shape[0], b.shape[1]), dtype=a.dtype)
        dt = _common_dtype(a, b, cast)
        a2, b2 = a.astype(dt), b.astype(dt)
        if dense:
            r = spmmd(a2, b2)
            if out is not None:
                out[...] = r
                return out
            return r
        r = spgemm(a2, b2)
        return type(a)(r) if a.format == "csr" else r.asformat(a.format)
    if empty:
        if out is not None:
            return out
        dt = _np.float32 if (a.dtype == b.dtype and a.dtype == _np.float32) else _np.float64
        return _np.zeros((a.shape[0], b.shape[1]), dtype=dt)
    dt = _common_dtype(a, b, cast)
    # sparse x dense VECTOR (reference sparse_dot.py:96-121 -> _sparse_vector.py:105-174): the result takes the
    # vector's shape convention -- (n,) for a 1-D vector, (n, 1) for a column on the right, (1, n) for a row on the left
    if sb and not sa and (a.ndim == 1 or (a.ndim == 2 and a.shape[0] == 1)):
        r = spmv(b.astype(dt), _np.asarray(a).astype(dt, copy=False), 1.0, beta, out, OP_T)
        r = r if a.ndim == 1 else r.reshape(1, -1)
    elif sa and not sb and (b.ndim == 1 or (b.ndim == 2 and b.shape[1] == 1)):
        r = spmv(a.astype(dt), _np.asarray(b).astype(dt, copy=False), 1.0, beta, out, OP_N)
        r = r if b.ndim == 1 else r.reshape(-1, 1)
    elif sa:
        bb = _np.asarray(b).astype(dt, copy=False)
        r = spmm(a.astype(dt), bb, 1.0, beta, out, OP_N)
    elif sb:
        # (B^T A^T)^T  -- reference _sparse_dense.py:191-208
        at = _np.asarray(a).astype(dt, copy=False).T
        r = spmm(b.astype(dt), at, 1.0, beta, None if out is None else out.T, OP_T).T
    else:
        r = gemm(_np.asarray(a).astype(dt, copy=False), _np.asarray(b).astype(dt, copy=False),
                 1.0, beta, out)
    if out is not None:
        out[...] = r
        return out
    return r


def gram_matrix(a, transpose=False, cast=False, dense=False, reorder_output=False, out=None,
                out_scalar=None):
    """Oracle restatement of gram_matrix_mkl (reference sparse_dot_mkl/_gram_matrix.py:252-335)."""
    beta = 1.0 if out_scalar is None else out_scalar
    valid = [_np.dtype(t) for t in (_np.float32, _np.float64)]
    if _np.dtype(a.dtype) not in valid:
        a = a.astype(_np.float64)
    if not _sps.issparse(a):
        r = syrk(_np.asarray(a), aat=transpose, alpha=1.0, beta=beta, c=out)
    elif dense:
        r = syrkd(a, aat=transpose, alpha=1.0, beta=beta, c=out)
    else:
        return syrk_sparse(a, aat=transpose)
    if out is not None:
        iu = _np.triu_indices(r.shape[0])
        out[iu] = r[iu]
        return out
    return r
