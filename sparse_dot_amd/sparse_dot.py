"""
Public entry points with the reference's exact signatures
(reference sparse_dot_mkl/sparse_dot.py:18-28 dot_product_mkl, 155-164 gram_matrix_mkl),
dispatching to the MI355X (gfx950) backend instead of Intel MKL.
"""
import warnings as _warnings

import numpy as _np
from scipy import sparse as _sps

from ._dense_dense import _dense_dot_dense as _ddd
from ._gram_matrix import _gram_matrix as _gm
from ._mi_interface import DeviceMatrix, _is_dense_vector, print_mi_debug, set_debug_mode  # noqa: F401
from ._sparse_dense import _sparse_dot_dense as _sdd
from ._sparse_sparse import _sparse_dot_sparse as _sds
from ._sparse_vector import _sparse_dot_vector as _sdv

_DEBUG_MSG = "Set debug mode with sparse_dot_amd.set_debug_mode(True)"


def dot_product_mkl(matrix_a, matrix_b, cast=False, copy=True, reorder_output=False, dense=False, debug=False,
                    out=None, out_scalar=None):
    """
    matrix_a @ matrix_b computed on the GPU.

    :param matrix_a, matrix_b: scipy CSR / CSC / BSR matrix (or *_array) or contiguous numpy array
        (1-d or 2-d); float32, float64, complex64 or complex128.
    :param cast: allow dtype conversion (to float64 / complex128) when the operands differ or are
        not floating point; otherwise a ValueError is raised.
    :param copy: deprecated, ignored.
    :param reorder_output: sort the column indices of a sparse result (unordered by default, as with
        scipy and MKL).
    :param dense: with two sparse operands, produce a dense array instead of a sparse matrix.
    :param debug: deprecated, use set_debug_mode(True).
    :param out: dense output array to accumulate into: out := a @ b + out_scalar * out.  Must have
        the exact shape, dtype and memory order of the result; the same object is returned.
    :param out_scalar: scaling of `out` (default 1.0).
    :return: sparse matrix of A's class when both operands are sparse (and not dense=True), else ndarray.
    """
    if debug:
        _warnings.warn(_DEBUG_MSG, DeprecationWarning)
    print_mi_debug()

    a_dev, b_dev = isinstance(matrix_a, DeviceMatrix), isinstance(matrix_b, DeviceMatrix)
    if a_dev or b_dev:
        # GPU-resident sparse operand (sparse_dot_amd.to_device): SpMM / SpMV against a dense operand
        other = matrix_b if a_dev else matrix_a
        if isinstance(other, DeviceMatrix) or _sps.issparse(other):
            raise ValueError("a DeviceMatrix can only be multiplied with a dense numpy operand")
        if other.ndim == 1:   # vector: run it as a one-column / one-row matrix and restore the shape
            o2 = other.reshape(-1, 1) if a_dev else other.reshape(1, -1)
            out2 = None if out is None else (out.reshape(-1, 1) if a_dev else out.reshape(1, -1))
            r = _sdd(matrix_a if a_dev else o2, o2 if a_dev else matrix_b, cast=cast, out=out2, out_scalar=out_scalar)
            return out if out is not None else r.ravel()
        return _sdd(matrix_a, matrix_b, cast=cast, out=out, out_scalar=out_scalar)

    a_sparse, b_sparse = _sps.issparse(matrix_a), _sps.issparse(matrix_b)

    if a_sparse and b_sparse:
        return _sds(matrix_a, matrix_b, cast=cast, reorder_output=reorder_output, dense=dense, out=out)

    if a_sparse or b_sparse:
        # a dense operand that is a vector on the contracted side goes through SpMV
        a_vec = _is_dense_vector(matrix_a) and (matrix_a.ndim == 1 or matrix_a.shape[0] == 1)
        b_vec = _is_dense_vector(matrix_b) and (matrix_b.ndim == 1 or matrix_b.shape[1] == 1)
        if a_vec or b_vec:
            return _sdv(matrix_a, matrix_b, cast=cast, out=out, out_scalar=out_scalar)
        return _sdd(matrix_a, matrix_b, cast=cast, out=out, out_scalar=out_scalar)

    # two dense operands
    if _is_dense_vector(matrix_a) and _is_dense_vector(matrix_b) and (matrix_a.ndim == 1 or matrix_b.ndim == 1):
        # vector . vector: numpy does this edge case, exactly like the reference (sparse_dot.py:135-142)
        if out_scalar is not None:
            out *= out_scalar
        return _np.dot(matrix_a, matrix_b, out=out)
    return _ddd(matrix_a, matrix_b, cast=cast, out=out, out_scalar=out_scalar)


def gram_matrix_mkl(matrix, transpose=False, cast=False, dense=False, debug=False, reorder_output=False, out=None,
                    out_scalar=None):
    """
    Upper triangle of the gram matrix A^T A (or A A^T with transpose=True), computed on the GPU.

    :param matrix: scipy CSR matrix (CSC with cast=True) or numpy array, float32 / float64.
    :param dense: dense ndarray output instead of a sparse csr_matrix (always dense for dense input).
    :param reorder_output: sort the column indices of a sparse result.
    :param out, out_scalar: accumulate into a dense `out` (only its upper triangle is read / written).
    """
    if debug:
        _warnings.warn(_DEBUG_MSG, DeprecationWarning)
    print_mi_debug()
    return _gm(matrix, transpose=transpose, cast=cast, dense=dense, reorder_output=reorder_output, out=out,
               out_scalar=out_scalar)


# backwards-compatible alias kept by the reference (sparse_dot.py:252)
dot_product_transpose_mkl = gram_matrix_mkl
