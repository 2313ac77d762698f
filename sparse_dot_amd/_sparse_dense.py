"""
Sparse x dense and dense x sparse products (SpMM) on the MI355X backend.

Same operator interface as the reference module (reference sparse_dot_mkl/_sparse_dense.py:34-208):
`_sparse_dense_matmul` is one call of the C-ABI routine mi_sparse_?_mm (the replacement of
mkl_sparse_?_mm); `_sparse_dot_dense` is the dispatcher-facing wrapper that validates, handles the
trivially-empty cases and turns dense x sparse into (B^T A^T)^T.
"""
import numpy as _np
from scipy import sparse as _sps

from ._mi_interface import (MI, LAYOUT_CODE_C, LAYOUT_CODE_F, DeviceMatrix, SparseHandle, _check_return_value,
                            _empty_output_check, _get_numpy_layout, _is_double, _mi_beta, _mi_scalar, _out_matrix,
                            _output_dtypes, _sanity_check, _type_check, _type_letters, debug_print, matrix_descr)


def _sparse_dense_matmul(matrix_a, matrix_b, scalar=1.0, transpose=False, out=None, out_scalar=None, out_t=None):
    """scalar * op(A) @ B + out_scalar * out, A sparse (CSR / CSC / BSR), B a contiguous 2-d array.
    The result has B's memory order.  Unlike MKL, no CSR conversion is needed for column-major B:
    the backend keeps whichever of CSR(A) / CSR(A^T) a call needs cached on the handle."""
    dbl, cplx = _is_double(matrix_a)
    out_rows = matrix_a.shape[1] if transpose else matrix_a.shape[0]
    out_shape = (out_rows, matrix_b.shape[1])
    layout_b, ld_b = _get_numpy_layout(matrix_b, second_arr=out)
    order = "C" if layout_b == LAYOUT_CODE_C else "F"
    output_arr = _out_matrix(out_shape, _output_dtypes[(dbl, cplx)], order, out_arr=out, out_t=out_t, overwritten=True)
    _, ld_out = _get_numpy_layout(output_arr, second_arr=matrix_b)
    name = "mi_sparse_%s_mm" % _type_letters[(dbl, cplx)]

    def run(handle):
        ret = MI.call(name, 11 if transpose else 10, _mi_scalar(scalar, cplx, dbl), handle.ptr, matrix_descr(),
                      layout_b, matrix_b.ctypes.data, out_shape[1], ld_b, _mi_beta(out, out_scalar, cplx, dbl),
                      output_arr.ctypes.data, ld_out)
        _check_return_value(ret, name)

    if isinstance(matrix_a, DeviceMatrix):   # resident handle: no upload, cached plan
        run(matrix_a.handle)
    else:
        with SparseHandle.from_scipy(matrix_a) as handle:
            run(handle)
    return output_arr


def _sparse_dot_dense(matrix_a, matrix_b, cast=False, scalar=1.0, out=None, out_scalar=None):
    """A @ B where exactly one operand is sparse; returns a dense array (or `out`)."""
    _sanity_check(matrix_a, matrix_b)

    if _empty_output_check(matrix_a, matrix_b):
        debug_print("Skipping multiplication because A (dot) B must yield an empty matrix")
        both_single = matrix_a.dtype == matrix_b.dtype and matrix_a.dtype == _np.float32
        return _out_matrix((matrix_a.shape[0], matrix_b.shape[1]), _np.float32 if both_single else _np.float64,
                           out_arr=out)

    if isinstance(matrix_a, DeviceMatrix) or isinstance(matrix_b, DeviceMatrix):
        if matrix_a.dtype != matrix_b.dtype:
            raise ValueError("a DeviceMatrix cannot be cast: operands must share its dtype (%s & %s provided)"
                             % (matrix_a.dtype, matrix_b.dtype))
    else:
        matrix_a, matrix_b = _type_check(matrix_a, matrix_b, cast=cast)
    a_sparse = _sps.issparse(matrix_a) or isinstance(matrix_a, DeviceMatrix)
    b_sparse = _sps.issparse(matrix_b) or isinstance(matrix_b, DeviceMatrix)
    if a_sparse == b_sparse:
        raise ValueError("_sparse_dot_dense takes one sparse and one dense array")
    if a_sparse:
        return _sparse_dense_matmul(matrix_a, matrix_b, scalar=scalar, out=out, out_scalar=out_scalar)
    # dense @ sparse  ==  (sparse^T @ dense^T)^T ; transposing a numpy array only flips its order flag
    if out is None:
        return _sparse_dense_matmul(matrix_b, matrix_a.T, scalar=scalar, transpose=True).T
    _sparse_dense_matmul(matrix_b, matrix_a.T, scalar=scalar, transpose=True, out=out.T, out_scalar=out_scalar,
                         out_t=True)
    return out
