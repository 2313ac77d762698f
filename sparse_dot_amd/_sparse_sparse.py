"""
Sparse x sparse products (SpGEMM) on the MI355X backend.

Operator interface of the reference module (reference sparse_dot_mkl/_sparse_sparse.py:21-244):
`_matmul_mi` is mi_sparse_spmm (replacement of mkl_sparse_spmm, sparse result handle),
`_matmul_mi_dense` is mi_sparse_?_spmmd (dense row-major result), `_sparse_dot_sparse` is the
dispatcher-facing wrapper.  The sparse result has the class of A; its column indices are unordered
unless reorder_output=True; entries that cancel to 0.0 are kept (as MKL keeps them).
"""
import ctypes as _ct

from ._mi_interface import (MI, SparseHandle, _check_return_value, _empty_output_check, _is_allowed_sparse_format,
                            _out_matrix, _output_dtypes, _sanity_check, _type_check, _type_letters, debug_timer,
                            is_bsr, sparse_matrix_t, sparse_output_type)


def _matmul_mi(handle_a, handle_b, ordered=False):
    """Sparse handle of A @ B (library-owned; the caller destroys it).  `ordered`: the rows come back with their column
    indices in increasing order (mi_sparse_spmm_ordered: the reference's mkl_sparse_spmm + mkl_sparse_order,
    _sparse_sparse.py:226-230, as one call -- the library then accumulates the long rows in order to begin with)."""
    if handle_a is None or handle_b is None:
        raise ValueError("mi_sparse_spmm returned 1 (SPARSE_STATUS_NOT_INITIALIZED)")
    out = sparse_matrix_t()
    name = "mi_sparse_spmm_ordered" if ordered else "mi_sparse_spmm"
    ret = MI.call(name, 10, handle_a.ptr, handle_b.ptr, _ct.byref(out))
    _check_return_value(ret, name)
    return SparseHandle(out, handle_a.letter)


def _matmul_mi_dense(handle_a, handle_b, output_shape, double_precision, out=None, complex_type=False):
    """Dense row-major A @ B from two sparse handles; a user `out` is OVERWRITTEN (spmmd has no beta)."""
    output_arr = _out_matrix(output_shape, _output_dtypes[(double_precision, complex_type)], out_arr=out)
    name = "mi_sparse_%s_spmmd" % _type_letters[(double_precision, complex_type)]
    ret = MI.call(name, 10, handle_a.ptr, handle_b.ptr, 101, output_arr.ctypes.data, output_shape[1])
    _check_return_value(ret, name)
    return output_arr


def _sparse_dot_sparse(matrix_a, matrix_b, cast=False, reorder_output=False, dense=False, out=None):
    if not (_is_allowed_sparse_format(matrix_a) and _is_allowed_sparse_format(matrix_b)):
        raise ValueError("Input matrices to dot_product_mkl must be CSR, CSC, or BSR; COO is not supported")
    if out is not None and not dense:
        raise ValueError("out argument cannot be used with sparse (dot) sparse matrix multiplication unless dense=True")
    make_output, output_type = sparse_output_type(matrix_a)
    _sanity_check(matrix_a, matrix_b)
    out_shape = (matrix_a.shape[0], matrix_b.shape[1])

    if _empty_output_check(matrix_a, matrix_b):
        if dense:
            return _out_matrix(out_shape, matrix_a.dtype, out_arr=out)
        return make_output(out_shape, dtype=matrix_a.dtype)

    matrix_a, matrix_b = _type_check(matrix_a, matrix_b, cast=cast)
    t = debug_timer()
    with SparseHandle.from_scipy(matrix_a) as ha, SparseHandle.from_scipy(matrix_b) as hb:
        t = debug_timer("Created sparse handles", t)
        dbl = ha.letter in "dz"
        cplx = ha.letter in "cz"
        if dense:
            result = _matmul_mi_dense(ha, hb, out_shape, dbl, out=out, complex_type=cplx)
            debug_timer("Multiplied matrices", t)
            return result
        with _matmul_mi(ha, hb, ordered=reorder_output) as hc:
            t = debug_timer("Multiplied matrices" + (" (rows ordered)" if reorder_output else ""), t)
            if is_bsr(matrix_a):
                if is_bsr(matrix_b) and matrix_a.blocksize == matrix_b.blocksize:
                    result = hc.export_bsr(output_type)  # re-blocked on the device (mi_sparse_?_export_bsr)
                else:
                    # the backend computes on the expanded CSR; blocks of A's size are cut on the host
                    csr = hc.export("csr_array" if output_type.endswith("array") else "csr_matrix")
                    result = make_output(csr, blocksize=matrix_a.blocksize)
            else:
                result = hc.export(output_type)
            debug_timer("Created python handle", t)
    return result
