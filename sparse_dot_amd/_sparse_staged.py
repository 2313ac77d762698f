"""
Two-stage sparse x sparse product (MKL's mkl_sparse_sp2m): the symbolic phase (pattern: row
pointer + nnz of C) and the numeric phase (indices + values) of the two-phase hash SpGEMM are separate calls, so
that a pattern analysed once serves many numeric products:

    p = StagedProduct(a, b)            # uploads A and B (CSR / CSC), nothing computed yet
    p.count()                          # NNZ_COUNT: nnz(C) and the row pointer are final
    c0 = p.finalize()                  # FINALIZE_MULT: values -> scipy matrix
    p.set_values(a=new_a_data)         # same patterns, new numbers
    c1 = p.finalize()                  # numeric phase only
    p.close()
"""
import ctypes as _ct

import numpy as _np
from scipy import sparse as _sps

from ._mi_interface import (MI, SPARSE_OPERATION_NON_TRANSPOSE, SPARSE_OPERATION_TRANSPOSE, SPARSE_STAGE_FINALIZE_MULT,
                            SPARSE_STAGE_FULL_MULT, SPARSE_STAGE_NNZ_COUNT, SparseHandle, _check_return_value,
                            _is_allowed_sparse_format, matrix_descr, sparse_matrix_t, sparse_output_type)


class StagedProduct:
    def __init__(self, matrix_a, matrix_b, transpose_a=False, transpose_b=False, reorder_output=False):
        for m in (matrix_a, matrix_b):
            if not _sps.issparse(m) or not _is_allowed_sparse_format(m):
                raise ValueError("StagedProduct needs scipy CSR / CSC / BSR operands")
        if matrix_a.dtype != matrix_b.dtype:
            raise ValueError("operands must share one dtype (%s & %s provided)" % (matrix_a.dtype, matrix_b.dtype))
        self._ops = (SPARSE_OPERATION_TRANSPOSE if transpose_a else SPARSE_OPERATION_NON_TRANSPOSE,
                     SPARSE_OPERATION_TRANSPOSE if transpose_b else SPARSE_OPERATION_NON_TRANSPOSE)
        ra = matrix_a.shape[::-1] if transpose_a else matrix_a.shape
        rb = matrix_b.shape[::-1] if transpose_b else matrix_b.shape
        if ra[1] != rb[0]:
            raise ValueError("Matrix alignment error: %s * %s is not valid" % (ra, rb))
        self.shape = (ra[0], rb[1])
        self._make_output, self._out_type = sparse_output_type(matrix_a)
        self._blocksize = matrix_a.blocksize if self._out_type.startswith("bsr") else None
        self._reorder = reorder_output
        self._ha = SparseHandle.from_scipy(matrix_a)
        self._hb = SparseHandle.from_scipy(matrix_b)
        self._hc = None
        self._nnz = (matrix_a.nnz, matrix_b.nnz)
        self._dtype = _np.dtype(matrix_a.dtype)

    def _call(self, request):
        c = self._hc.ptr if self._hc is not None else sparse_matrix_t()
        ret = MI.call("mi_sparse_sp2m", self._ops[0], matrix_descr(), self._ha.ptr, self._ops[1], matrix_descr(),
                      self._hb.ptr, request, _ct.byref(c))
        _check_return_value(ret, "mi_sparse_sp2m")
        if self._hc is None:
            self._hc = SparseHandle(c, self._ha.letter)

    def count(self):
        """Symbolic phase.  Returns nnz(C)."""
        if self._hc is not None:
            self._hc.destroy()
            self._hc = None
        self._call(SPARSE_STAGE_NNZ_COUNT)
        return self._hc.info()[2]

    def finalize(self):
        """Numeric phase on the counted pattern (runs count() first when needed).  Returns the product."""
        if self._hc is None:
            self.count()
        self._call(SPARSE_STAGE_FINALIZE_MULT)
        return self._export()

    def _export(self):
        if self._reorder:
            self._hc.order()
        if self._blocksize is not None:  # the backend computes on the expanded CSR; re-block on the way out
            csr = self._hc.export("csr_array" if self._out_type.endswith("array") else "csr_matrix")
            return self._make_output(csr, blocksize=self._blocksize)
        return self._hc.export(self._out_type)

    def full(self):
        """Both phases in one call (== dot_product_mkl(a, b))."""
        if self._hc is not None:
            self._hc.destroy()
            self._hc = None
        self._call(SPARSE_STAGE_FULL_MULT)
        return self._export()

    def set_values(self, a=None, b=None):
        """New values (same storage order and count as the matrices given at construction) for A and / or B."""
        for h, vals, n in ((self._ha, a, self._nnz[0]), (self._hb, b, self._nnz[1])):
            if vals is None:
                continue
            vals = _np.ascontiguousarray(vals, dtype=self._dtype)
            if vals.shape != (n,):
                raise ValueError("value array must hold %d entries, %s provided" % (n, vals.shape))
            name = "mi_sparse_%s_set_values" % h.letter
            _check_return_value(MI.call(name, h.ptr, vals.ctypes.data), name)

    def close(self):
        for h in (self._hc, self._ha, self._hb):
            if h is not None:
                h.destroy()
        self._hc = self._ha = self._hb = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
