"""
Gram matrices A^T A / A A^T on the MI355X backend (upper triangle).

Operator interface of the reference module (reference sparse_dot_mkl/_gram_matrix.py:43-335):
  sparse -> sparse : mi_sparse_syrk      (replacement of mkl_sparse_syrk)
  sparse -> dense  : mi_sparse_?_syrkd   (replacement of mkl_sparse_?_syrkd)
  dense  -> dense  : mi_cblas_?syrk      (replacement of cblas_?syrk)
Only the upper triangle of a dense result is defined; for a fresh output the strict lower triangle
is zero (the backend never writes it, so no O(n^2) index arrays are built to clear it -- the
reference clears it with np.tril_indices, _gram_matrix.py:168-169).
Deliberate deviation: an empty input returns the mathematically correct (n, n) shape; the
reference's empty shortcut swaps the two cases (_gram_matrix.py:288-292 vs 134).
"""
import ctypes as _ct

import numpy as _np
from scipy import sparse as _sps

from ._mi_interface import (MI, SparseHandle, CBLAS_NO_TRANS, CBLAS_TRANS, LAYOUT_CODE_C, MI_UPPER,
                            SPARSE_OPERATION_NON_TRANSPOSE, SPARSE_OPERATION_TRANSPOSE, _check_return_value,
                            _empty_output_check, _get_numpy_layout, _is_double, _mi_scalar, _out_matrix,
                            _output_dtypes, _type_check, _type_letters, debug_print, is_csc, is_csr, sparse_matrix_t)


def _op(aat):
    # same codes as the reference: A A^T is op 10 on A, A^T A is op 11
    return SPARSE_OPERATION_NON_TRANSPOSE if aat else SPARSE_OPERATION_TRANSPOSE


def _gram_matrix_sparse(matrix_a, aat=False, reorder_output=False):
    """Upper-triangular sparse gram matrix; always a csr_matrix, like the reference."""
    with SparseHandle.from_scipy(matrix_a) as ha:
        out = sparse_matrix_t()
        ret = MI.call("mi_sparse_syrk", _op(aat), ha.ptr, _ct.byref(out))
        _check_return_value(ret, "mi_sparse_syrk")
        with SparseHandle(out, ha.letter) as hc:
            if reorder_output:
                hc.order()
            return hc.export("csr_matrix")


def _gram_matrix_sparse_to_dense(matrix_a, aat=False, scalar=1.0, out=None, out_scalar=None):
    dbl, cplx = _is_double(matrix_a)
    n = matrix_a.shape[0 if aat else 1]
    output_arr = _out_matrix((n, n), _output_dtypes[(dbl, cplx)], order="C", out_arr=out)
    if _empty_output_check(matrix_a, matrix_a):
        return output_arr
    _, ld = _get_numpy_layout(output_arr)
    name = "mi_sparse_%s_syrkd" % _type_letters[(dbl, cplx)]
    with SparseHandle.from_scipy(matrix_a) as ha:
        ret = MI.call(name, _op(aat), ha.ptr, _mi_scalar(scalar, cplx, dbl), _mi_scalar(out_scalar, cplx, dbl),
                      output_arr.ctypes.data, LAYOUT_CODE_C, ld)
        _check_return_value(ret, name)
    return output_arr


def _gram_rows_dense(matrix_a, row0, row1, aat=False):
    """Rows [row0, row1) of the dense upper-triangular gram matrix as a fresh (row1 - row0, n) array
    (mi_sparse_?_syrkd_rows): the building block of the multi-GPU gram, which splits the OUTPUT by rows so that
    no reduction is needed (sparse_dot_amd.distributed.sharded_gram_matrix).  Entries left of the diagonal stay 0."""
    matrix_a = _type_check(matrix_a)
    dbl, cplx = _is_double(matrix_a)
    if cplx:
        raise ValueError("gram_matrix_mkl does not support complex datatypes")
    n = matrix_a.shape[0 if aat else 1]
    if not (0 <= row0 <= row1 <= n):
        raise ValueError("bad output row range [%d, %d) of %d" % (row0, row1, n))
    output_arr = _np.zeros((row1 - row0, n), dtype=_output_dtypes[(dbl, cplx)])
    if row1 == row0 or _empty_output_check(matrix_a, matrix_a):
        return output_arr
    name = "mi_sparse_%s_syrkd_rows" % _type_letters[(dbl, cplx)]
    with SparseHandle.from_scipy(matrix_a) as ha:
        ret = MI.call(name, _op(aat), ha.ptr, _mi_scalar(1.0, cplx, dbl), _mi_scalar(0.0, cplx, dbl),
                      output_arr.ctypes.data, LAYOUT_CODE_C, n, row0, row1)
        _check_return_value(ret, name)
    return output_arr


def _gram_matrix_dense_to_dense(matrix_a, aat=False, scalar=1.0, out=None, out_scalar=None):
    n, k = matrix_a.shape if aat else matrix_a.shape[::-1]
    layout_a, ld_a = _get_numpy_layout(matrix_a)
    dbl, cplx = _is_double(matrix_a)
    output_arr = _out_matrix((n, n), _output_dtypes[(dbl, cplx)], order="C" if layout_a == LAYOUT_CODE_C else "F",
                             out_arr=out)
    name = "mi_cblas_%ssyrk" % _type_letters[(dbl, cplx)]
    ret = MI.call(name, layout_a, MI_UPPER, CBLAS_NO_TRANS if aat else CBLAS_TRANS, n, k,
                  _mi_scalar(scalar, cplx, dbl), matrix_a.ctypes.data, ld_a, _mi_scalar(out_scalar, cplx, dbl),
                  output_arr.ctypes.data, n)
    _check_return_value(ret, name)
    return output_arr


def _gram_matrix(matrix, transpose=False, cast=False, dense=False, reorder_output=False, out=None, out_scalar=None):
    if _empty_output_check(matrix, matrix):
        debug_print("Skipping multiplication because AT (dot) A must yield an empty matrix")
        n = matrix.shape[0] if transpose else matrix.shape[1]
        if _sps.issparse(matrix) and not dense:
            return _sps.csr_matrix((n, n), dtype=matrix.dtype)
        return _np.zeros((n, n), dtype=matrix.dtype)

    if _np.iscomplexobj(matrix):
        raise ValueError("gram_matrix_mkl does not support complex datatypes")
    matrix = _type_check(matrix, cast=cast)

    if not _sps.issparse(matrix):
        return _gram_matrix_dense_to_dense(matrix, aat=transpose, out=out, out_scalar=out_scalar)
    if not (is_csr(matrix) or is_csc(matrix)):
        raise ValueError("gram_matrix requires sparse matrix to be CSR or CSC format")
    if is_csc(matrix) and not cast:
        raise ValueError("gram_matrix cannot use a CSC matrix unless cast=True")
    if dense:
        return _gram_matrix_sparse_to_dense(matrix, aat=transpose, out=out, out_scalar=out_scalar)
    if out is not None:
        raise ValueError("out argument cannot be used with sparse (dot) sparse matrix multiplication")
    return _gram_matrix_sparse(matrix, aat=transpose, reorder_output=reorder_output)
