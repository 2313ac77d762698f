"""
Minimal ctypes binding of Intel MKL's mkl_sparse_?_mm / mkl_sparse_spmm / mkl_sparse_?_syrkd, used ONLY
by bench.py's `cpu_baseline` legs to time the arithmetic engine the reference itself calls, on the GPU
box's host cores.

TEST / MEASUREMENT INFRASTRUCTURE -- never imported by the product package.  This is the build's
own shim (the reference's Python does not travel to the GPU box); it binds the three symbols the
reference's SpMM path uses (reference sparse_dot_mkl/_sparse_dense.py:111-123,
_mkl_interface/_common.py:310-319, 671-680), plus mkl_sparse_spmm (_sparse_sparse.py:35-40),
mkl_sparse_?_syrkd (_gram_matrix.py:149-157), mkl_sparse_order / mkl_sparse_?_export_csr (_common.py:387-500, 683) and
MKL_Set_Num_Threads.  LP64 (32-bit index) interface by default; `interface=1` selects ILP64
(MKL_Set_Interface_Layer(1), reference _cfunctions.py:774-782).

The same binding drives `sparse_dot_amd/libmi_mkl_rt.so` -- the MKL-named face of the build -- in
tests/test_gpu_mkl_alias.py: a library that answers these calls like MKL does is a drop-in behind `$MKL_RT`.
"""
import ctypes as _ct
import ctypes.util as _ctu
import os as _os

import numpy as _np


class _Descr(_ct.Structure):
    _fields_ = [("type", _ct.c_int), ("mode", _ct.c_int), ("diag", _ct.c_int)]


def find_mkl():
    """Path of a loadable libmkl_rt, or None."""
    cands = []
    if _os.environ.get("MKL_RT"):
        cands.append(_os.environ["MKL_RT"])
    cands += ["/opt/conda/lib/libmkl_rt.so", "/opt/conda/lib/libmkl_rt.so.1", "/opt/conda/lib/libmkl_rt.so.2"]
    found = _ctu.find_library("mkl_rt")
    if found:
        cands.append(found)
    for c in cands:
        try:
            _ct.CDLL(c)
            return c
        except OSError:
            continue
    return None


class MklSpmm:
    """C := A @ B with A CSR (int32 indices), B / C row-major, via mkl_sparse_?_mm."""

    def __init__(self, path=None, interface=0):
        path = path or find_mkl()
        if path is None:
            raise OSError("libmkl_rt not found")
        self.path = path
        self.lib = _ct.CDLL(path)
        self.interface = int(interface)
        try:
            self.lib.MKL_Set_Interface_Layer(_ct.c_int(self.interface))  # 0 = LP64, 1 = ILP64
        except AttributeError:
            pass
        self.idt = _np.int64 if self.interface else _np.int32       # MKL_INT as numpy dtype ...
        self.cint = _ct.c_longlong if self.interface else _ct.c_int  # ... and as ctypes argument
        self.lib.MKL_Get_Max_Threads.restype = _ct.c_int

    def threads(self):
        return int(self.lib.MKL_Get_Max_Threads())

    def version(self):
        buf = _ct.create_string_buffer(256)
        self.lib.MKL_Get_Version_String(buf, 256)
        return buf.value.decode().strip()

    def make(self, a):
        dt = _np.dtype(a.dtype)
        letter = {"float32": "s", "float64": "d"}[dt.name]
        indptr = _np.ascontiguousarray(a.indptr, dtype=self.idt)
        indices = _np.ascontiguousarray(a.indices, dtype=self.idt)
        data = _np.ascontiguousarray(a.data)
        h = _ct.c_void_p()
        create = getattr(self.lib, "mkl_sparse_%s_create_csr" % letter)
        create.restype = _ct.c_int
        st = create(_ct.byref(h), _ct.c_int(0), self.cint(a.shape[0]), self.cint(a.shape[1]),
                    _ct.c_void_p(indptr.ctypes.data), _ct.c_void_p(indptr.ctypes.data + indptr.itemsize),
                    _ct.c_void_p(indices.ctypes.data), _ct.c_void_p(data.ctypes.data))
        if st:
            raise RuntimeError("mkl_sparse_%s_create_csr returned %d" % (letter, st))
        return (h, letter, (indptr, indices, data))

    def mm(self, handle, b, out):
        h, letter, _keep = handle
        fn = getattr(self.lib, "mkl_sparse_%s_mm" % letter)
        ct = _ct.c_float if letter == "s" else _ct.c_double
        fn.restype = _ct.c_int
        fn.argtypes = [_ct.c_int, ct, _ct.c_void_p, _Descr, _ct.c_int, _ct.c_void_p, self.cint, self.cint, ct,
                       _ct.c_void_p, self.cint]
        st = fn(10, 1.0, h, _Descr(20, 0, 0), 101, b.ctypes.data, b.shape[1], b.shape[1], 0.0, out.ctypes.data,
                out.shape[1])
        if st:
            raise RuntimeError("mkl_sparse_%s_mm returned %d" % (letter, st))
        return out

    def destroy(self, handle):
        self.lib.mkl_sparse_destroy(handle[0])

    def set_threads(self, n):
        """MKL_Set_Num_Threads (reference _mkl_interface/__init__.py:62-77); returns the resulting maximum."""
        self.lib.MKL_Set_Num_Threads(_ct.c_int(int(n)))
        return self.threads()

    def spmm_handle(self, ha, hb):
        """C := A @ B as a new library-owned handle (mkl_sparse_spmm); pair with order / export_csr / destroy."""
        c = _ct.c_void_p()
        self.lib.mkl_sparse_spmm.restype = _ct.c_int
        st = self.lib.mkl_sparse_spmm(_ct.c_int(10), ha[0], hb[0], _ct.byref(c))
        if st:
            raise RuntimeError("mkl_sparse_spmm returned %d" % st)
        return (c, ha[1], None)

    def export_csr(self, handle):
        """(indptr, indices, data, shape) COPIED out of a handle with mkl_sparse_?_export_csr, the way the reference
        does it (_common.py:387-500): library-owned pointers, rows_end[-1] closes the row pointer, base must be 0."""
        h, letter, _keep = handle
        fn = getattr(self.lib, "mkl_sparse_%s_export_csr" % letter)
        fn.restype = _ct.c_int
        base, rows, cols = _ct.c_int(), self.cint(), self.cint()
        rs, re, ci, va = (_ct.c_void_p() for _ in range(4))
        st = fn(h, _ct.byref(base), _ct.byref(rows), _ct.byref(cols), _ct.byref(rs), _ct.byref(re), _ct.byref(ci),
                _ct.byref(va))
        if st:
            raise RuntimeError("mkl_sparse_%s_export_csr returned %d" % (letter, st))
        assert base.value == 0
        m = int(rows.value)
        isz = _np.dtype(self.idt).itemsize
        vdt = _np.float32 if letter == "s" else _np.float64

        def arr(ptr, count, dtype):
            if count == 0:
                return _np.zeros(0, dtype=dtype)
            buf = (_ct.c_char * (count * _np.dtype(dtype).itemsize)).from_address(ptr.value)
            return _np.frombuffer(buf, dtype=dtype).copy()
        start = arr(rs, m, self.idt)
        end = arr(re, m, self.idt)
        nnz = int(end[-1]) if m else 0
        indptr = _np.concatenate([start[:1] if m else _np.zeros(1, self.idt), end]).astype(self.idt)
        return indptr, arr(ci, nnz, self.idt), arr(va, nnz, vdt), (m, int(cols.value))

    def spmm(self, ha, hb):
        """C := A @ B, both sparse: mkl_sparse_spmm, then the handle is destroyed (the multiply is what is timed;
        the reference's export adds Python-side copies on top).  Returns nothing."""
        c = _ct.c_void_p()
        self.lib.mkl_sparse_spmm.restype = _ct.c_int
        st = self.lib.mkl_sparse_spmm(_ct.c_int(10), ha[0], hb[0], _ct.byref(c))
        if st:
            raise RuntimeError("mkl_sparse_spmm returned %d" % st)
        self.lib.mkl_sparse_destroy(c)

    def syrkd(self, handle, out):
        """out(upper triangle) := A^T A, dense row-major: mkl_sparse_?_syrkd(op = 11)."""
        h, letter, _keep = handle
        fn = getattr(self.lib, "mkl_sparse_%s_syrkd" % letter)
        ct = _ct.c_float if letter == "s" else _ct.c_double
        fn.restype = _ct.c_int
        fn.argtypes = [_ct.c_int, _ct.c_void_p, ct, ct, _ct.c_void_p, _ct.c_int, self.cint]
        st = fn(11, h, 1.0, 0.0, out.ctypes.data, 101, out.shape[1])
        if st:
            raise RuntimeError("mkl_sparse_%s_syrkd returned %d" % (letter, st))
        return out

    def mv(self, handle, x, y):
        """y := A x: mkl_sparse_?_mv (reference _sparse_vector.py:87-95)."""
        h, letter, _keep = handle
        fn = getattr(self.lib, "mkl_sparse_%s_mv" % letter)
        ct = _ct.c_float if letter == "s" else _ct.c_double
        fn.restype = _ct.c_int
        fn.argtypes = [_ct.c_int, ct, _ct.c_void_p, _Descr, _ct.c_void_p, ct, _ct.c_void_p]
        st = fn(10, 1.0, h, _Descr(20, 0, 0), x.ctypes.data, 0.0, y.ctypes.data)
        if st:
            raise RuntimeError("mkl_sparse_%s_mv returned %d" % (letter, st))
        return y

    def syrk(self, handle):
        """Upper triangle of A^T A as a sparse handle: mkl_sparse_syrk(op = 11) (reference _gram_matrix.py:70-74: the rows
        are ordered first, as the reference does), result destroyed at once -- the multiply is what is timed."""
        c = _ct.c_void_p()
        self.lib.mkl_sparse_syrk.restype = _ct.c_int
        st = self.lib.mkl_sparse_syrk(_ct.c_int(11), handle[0], _ct.byref(c))
        if st:
            raise RuntimeError("mkl_sparse_syrk returned %d" % st)
        self.lib.mkl_sparse_destroy(c)

    def spmmd(self, ha, hb, out):
        """out := A @ B, dense row-major: mkl_sparse_?_spmmd (reference _sparse_sparse.py:94-101)."""
        letter = ha[1]
        fn = getattr(self.lib, "mkl_sparse_%s_spmmd" % letter)
        fn.restype = _ct.c_int
        fn.argtypes = [_ct.c_int, _ct.c_void_p, _ct.c_void_p, _ct.c_int, _ct.c_void_p, self.cint]
        st = fn(10, ha[0], hb[0], 101, out.ctypes.data, out.shape[1])
        if st:
            raise RuntimeError("mkl_sparse_%s_spmmd returned %d" % (letter, st))
        return out

    def order(self, handle):
        self.lib.mkl_sparse_order.restype = _ct.c_int
        return self.lib.mkl_sparse_order(handle[0])
