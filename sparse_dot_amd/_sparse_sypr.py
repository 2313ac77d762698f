"""
Symmetric triple product  op(A) @ B @ op(A)^T  with symmetric B, on the MI355X backend.

Operator interface of the reference module (reference sparse_dot_mkl/_sparse_sypr.py:29-181, dead upstream: nothing
imports it and `MKL._mkl_sparse_sypr` is never bound, so there is no reference behaviour to pin beyond the
signatures): sparse B -> mi_sparse_sypr (replacement of mkl_sparse_sypr), dense B -> mi_sparse_?_syprd
(replacement of mkl_sparse_?_syprd).  B is given by its UPPER triangle; the result holds the upper triangle.
"""
import ctypes as _ct
import warnings

import numpy as _np
from scipy import sparse as _sps

from ._mi_interface import (MI, LAYOUT_CODE_C, SPARSE_DIAG_NON_UNIT, SPARSE_FILL_MODE_UPPER, SPARSE_MATRIX_TYPE_SYMMETRIC,
                            SPARSE_OPERATION_NON_TRANSPOSE, SPARSE_OPERATION_TRANSPOSE, SPARSE_STAGE_FULL_MULT, SparseHandle,
                            _check_return_value, _get_numpy_layout, _is_double, _out_matrix, _output_dtypes, _type_check,
                            _type_letters, is_bsr, is_csr, matrix_descr, sparse_matrix_t)


def _sypr_sparse_A_dense_B(matrix_a, matrix_b, transpose_a=False, out=None, out_scalar=None, a_scalar=None):
    """alpha * op(A) @ B @ op(A)^T + beta * out with dense symmetric B; dense result (upper triangle defined)."""
    dbl, cplx = _is_double(matrix_a)
    if cplx:
        raise ValueError("sypr does not support complex datatypes")
    n_out = matrix_a.shape[1] if transpose_a else matrix_a.shape[0]
    k = matrix_a.shape[0] if transpose_a else matrix_a.shape[1]
    if matrix_b.shape != (k, k):
        raise ValueError("Matrix alignment error: B must be %d x %d, %s provided" % (k, k, matrix_b.shape))
    layout_b, ld_b = _get_numpy_layout(matrix_b, second_arr=out)
    output_arr = _out_matrix((n_out, n_out), _output_dtypes[(dbl, cplx)], order="C" if layout_b == LAYOUT_CODE_C else "F",
                             out_arr=out)
    output_layout, output_ld = _get_numpy_layout(output_arr, second_arr=matrix_b)
    name = "mi_sparse_%s_syprd" % _type_letters[(dbl, cplx)]
    alpha = 1.0 if a_scalar is None else float(a_scalar)
    beta = (1.0 if out_scalar is None else float(out_scalar)) if out is not None else 0.0
    with SparseHandle.from_scipy(matrix_a) as ha:
        ret = MI.call(name, SPARSE_OPERATION_TRANSPOSE if transpose_a else SPARSE_OPERATION_NON_TRANSPOSE, ha.ptr,
                      matrix_b.ctypes.data, layout_b, ld_b, alpha, beta, output_arr.ctypes.data, output_layout, output_ld)
        _check_return_value(ret, name)
    return output_arr


def _sypr_sparse_A_sparse_B(matrix_a, matrix_b, transpose_a=False):
    """triu(op(A) @ B @ op(A)^T) as a csr_matrix; B symmetric, its upper triangle is what is read."""
    descr_b = matrix_descr(SPARSE_MATRIX_TYPE_SYMMETRIC, SPARSE_FILL_MODE_UPPER, SPARSE_DIAG_NON_UNIT)
    with SparseHandle.from_scipy(matrix_a) as ha, SparseHandle.from_scipy(matrix_b) as hb:
        out = sparse_matrix_t()
        ret = MI.call("mi_sparse_sypr", SPARSE_OPERATION_TRANSPOSE if transpose_a else SPARSE_OPERATION_NON_TRANSPOSE, ha.ptr,
                      hb.ptr, descr_b, _ct.byref(out), SPARSE_STAGE_FULL_MULT)
        _check_return_value(ret, "mi_sparse_sypr")
        with SparseHandle(out, ha.letter) as hc:
            return hc.export("csr_matrix")


def _sparse_sypr(matrix_a, matrix_b, transpose_a=False, cast=False, out=None, out_scalar=None, scalar=None):
    matrix_a, matrix_b = _type_check(matrix_a, matrix_b, cast=cast)
    if _np.iscomplexobj(matrix_a):
        raise ValueError("sypr does not support complex datatypes")
    if not (is_csr(matrix_a) or is_bsr(matrix_a)) or not (is_csr(matrix_b) or is_bsr(matrix_b) or not _sps.issparse(matrix_b)):
        raise ValueError("Input matrices to spyr must be CSR or BSR; CSC and COO are not supported")
    if _sps.issparse(matrix_b):
        if out is not None or out_scalar is not None or scalar is not None:
            warnings.warn("out, out_scalar, and scalar have no effect if matrix B is not sparse", RuntimeWarning)
        return _sypr_sparse_A_sparse_B(matrix_a, matrix_b, transpose_a=transpose_a)
    return _sypr_sparse_A_dense_B(matrix_a, matrix_b, transpose_a=transpose_a, out=out, out_scalar=out_scalar,
                                  a_scalar=scalar)
