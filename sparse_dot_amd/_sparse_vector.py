"""
Sparse matrix x dense vector (SpMV) and vector x sparse matrix.

Operator interface of the reference module (reference sparse_dot_mkl/_sparse_vector.py:28-174);
the arithmetic is mi_sparse_?_mv (replacement of mkl_sparse_?_mv), i.e. the N = 1 case of the
SpMM kernel.  The result has the vector's shape convention: (n,) for a 1-d vector, (n, 1) / (1, n)
for a 2-d one.
"""
import numpy as _np

from ._mi_interface import (MI, SparseHandle, _check_return_value, _empty_output_check, _is_allowed_sparse_format,
                            _is_dense_vector, _is_double, _mi_beta, _mi_scalar, _out_matrix, _output_dtypes, _sanity_check,
                            _type_check, _type_letters, matrix_descr)


def _sparse_dense_vector_mult(matrix_a, vector_b, scalar=1.0, transpose=False, out=None, out_scalar=None, out_t=None):
    n_out = matrix_a.shape[1] if transpose else matrix_a.shape[0]
    out_shape = (n_out,) if vector_b.ndim == 1 else (n_out, 1)

    if _empty_output_check(matrix_a, vector_b):
        both_single = matrix_a.dtype == vector_b.dtype and matrix_a.dtype == _np.float32
        return _out_matrix(out_shape, _np.float32 if both_single else _np.float64, out_arr=out)

    dbl, cplx = _is_double(matrix_a)
    flat_b = _np.ascontiguousarray(vector_b.ravel())
    output_arr = _out_matrix(out_shape, _output_dtypes[(dbl, cplx)], out_arr=out, out_t=out_t, overwritten=True)
    name = "mi_sparse_%s_mv" % _type_letters[(dbl, cplx)]
    with SparseHandle.from_scipy(matrix_a) as handle:
        ret = MI.call(name, 11 if transpose else 10, _mi_scalar(scalar, cplx, dbl), handle.ptr, matrix_descr(),
                      flat_b.ctypes.data, _mi_beta(out, out_scalar, cplx, dbl), output_arr.ctypes.data)
        _check_return_value(ret, name)
    return output_arr


def _sparse_dot_vector(mv_a, mv_b, cast=False, scalar=1.0, out=None, out_scalar=None):
    """One operand sparse, the other a dense vector ((n,), (n, 1) or (1, n))."""
    _sanity_check(mv_a, mv_b, allow_vector=True)
    mv_a, mv_b = _type_check(mv_a, mv_b, cast=cast)
    if not (_is_allowed_sparse_format(mv_a) and _is_allowed_sparse_format(mv_b)):
        raise ValueError("Only CSR, CSC, and BSR-type sparse matrices are supported")
    if _is_dense_vector(mv_b):
        return _sparse_dense_vector_mult(mv_a, mv_b, scalar=scalar, out=out, out_scalar=out_scalar)
    if _is_dense_vector(mv_a):
        # v @ S == (S^T @ v^T)^T
        if out is None:
            return _sparse_dense_vector_mult(mv_b, mv_a.T, scalar=scalar, transpose=True).T
        _sparse_dense_vector_mult(mv_b, mv_a.T, scalar=scalar, transpose=True, out=out.T, out_scalar=out_scalar,
                                  out_t=True)
        return out
    raise ValueError("Neither mv_a or mv_b is a dense vector")
