"""
Dense x dense fallback: mi_cblas_?gemm (replacement of cblas_?gemm; the backend's only MFMA user).
Operator interface of the reference module (reference sparse_dot_mkl/_dense_dense.py:14-83).
"""
import ctypes as _ct

import numpy as _np

from ._mi_interface import (MI, CBLAS_NO_TRANS, CBLAS_TRANS, LAYOUT_CODE_C, _check_return_value, _empty_output_check,
                            _get_numpy_layout, _is_double, _mi_beta, _mi_scalar, _out_matrix, _output_dtypes, _sanity_check,
                            _type_check, _type_letters, debug_print)


def _dense_matmul(matrix_a, matrix_b, scalar=1.0, out=None, out_scalar=None):
    dbl, cplx = _is_double(matrix_a)
    vector_b = matrix_b.ndim == 1
    if vector_b:
        matrix_b = matrix_b.reshape(-1, 1)
    m, k = matrix_a.shape
    n = matrix_b.shape[1]
    layout_a, ld_a = _get_numpy_layout(matrix_a)
    layout_b, ld_b = _get_numpy_layout(matrix_b)
    # the call runs in A's layout; a B stored the other way round is the transpose of a matrix in A's layout
    op_b = CBLAS_NO_TRANS if layout_b == layout_a else CBLAS_TRANS
    order, ld_out = ("C", n) if layout_a == LAYOUT_CODE_C else ("F", m)
    output_arr = _out_matrix((m, n), _output_dtypes[(dbl, cplx)], order=order, out_arr=out, overwritten=True)
    alpha, beta = _mi_scalar(scalar, cplx, dbl), _mi_beta(out, out_scalar, cplx, dbl)
    if cplx:  # CBLAS convention: complex scalars by pointer
        alpha, beta = _ct.byref(alpha), _ct.byref(beta)
    name = "mi_cblas_%sgemm" % _type_letters[(dbl, cplx)]
    ret = MI.call(name, layout_a, CBLAS_NO_TRANS, op_b, m, n, k, alpha, matrix_a.ctypes.data, ld_a,
                  matrix_b.ctypes.data, ld_b, beta, output_arr.ctypes.data, ld_out)
    _check_return_value(ret, name)
    return output_arr.ravel() if vector_b else output_arr


def _dense_dot_dense(matrix_a, matrix_b, cast=False, scalar=1.0, out=None, out_scalar=None):
    _sanity_check(matrix_a, matrix_b, allow_vector=True)
    if _empty_output_check(matrix_a, matrix_b):
        debug_print("Skipping multiplication because A (dot) B must yield an empty matrix")
        both_single = matrix_a.dtype == matrix_b.dtype and matrix_a.dtype == _np.float32
        return _out_matrix((matrix_a.shape[0], matrix_b.shape[1]), _np.float32 if both_single else _np.float64,
                           out_arr=out)
    matrix_a, matrix_b = _type_check(matrix_a, matrix_b, cast=cast)
    return _dense_matmul(matrix_a, matrix_b, scalar=scalar, out=out, out_scalar=out_scalar)
