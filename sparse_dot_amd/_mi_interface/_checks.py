"""
Input validation, dtype / layout rules and output-array contract of the public functions.

Behavioural contract follows the reference's marshalling helpers
(reference sparse_dot_mkl/_mkl_interface/_common.py:158-213, 725-1024): every user error is a
ValueError; operands with the same dtype in {float32, float64, complex64, complex128} pass
through by identity; otherwise cast=True converts (to float64 / complex128 / the complex operand's
dtype) and cast=False raises; a user `out` must match shape, dtype, order and be contiguous and
is returned as the same object.
"""
import time as _time

import numpy as _np
from scipy import sparse as _sps

from ._constants import LAYOUT_CODE_C, LAYOUT_CODE_F, STATUS_NAMES
from ._library import MI, Complex8, Complex16, mi_library_name

NUMPY_FLOAT_DTYPES = (_np.dtype(_np.float32), _np.dtype(_np.float64))
NUMPY_COMPLEX_DTYPES = (_np.dtype(_np.complex64), _np.dtype(_np.complex128))

# keyed by (double_precision, complex)
_output_dtypes = {
    (False, False): _np.float32,
    (True, False): _np.float64,
    (False, True): _np.complex64,
    (True, True): _np.complex128,
}
_type_letters = {(False, False): "s", (True, False): "d", (False, True): "c", (True, True): "z"}

_CSR_TYPES = (_sps.csr_matrix, _sps.csr_array)
_CSC_TYPES = (_sps.csc_matrix, _sps.csc_array)
_BSR_TYPES = (_sps.bsr_matrix, _sps.bsr_array)
_NAMED_TYPES = {
    "csr_matrix": _sps.csr_matrix, "csr_array": _sps.csr_array,
    "csc_matrix": _sps.csc_matrix, "csc_array": _sps.csc_array,
    "bsr_matrix": _sps.bsr_matrix, "bsr_array": _sps.bsr_array,
}


# ---- debug plumbing (reference _common.py:97-155) --------------------------------------------
def set_debug_mode(debug_bool):
    """Turn the library's diagnostic printing on or off."""
    MI.DEBUG = bool(debug_bool)


def print_mi_debug():
    if not MI.DEBUG:
        return
    from . import mi_get_version_string, mi_get_device_count
    print(mi_get_version_string())
    print("HIP devices visible: %d" % mi_get_device_count())
    print("backend library: %s" % mi_library_name())
    print("index arrays: int32 and int64 are both accepted as-is (no interface-layer switch)")


def debug_print(msg):
    if MI.DEBUG:
        print(msg)


def debug_timer(msg=None, old_time=None):
    if not MI.DEBUG:
        return None
    now = _time.time()
    if msg is not None and old_time is not None:
        print("%s: %.6f seconds" % (msg, now - old_time))
    return now


# ---- format predicates -------------------------------------------------------------------------
def is_csr(x):
    return isinstance(x, _CSR_TYPES)


def is_csc(x):
    return isinstance(x, _CSC_TYPES)


def is_bsr(x):
    return isinstance(x, _BSR_TYPES)


def _is_allowed_sparse_format(matrix):
    """Dense arrays are fine; sparse ones must be CSR, CSC or BSR."""
    return (not _sps.issparse(matrix)) or is_csr(matrix) or is_csc(matrix) or is_bsr(matrix)


def sparse_output_type(x):
    """(constructor, name) of the scipy class a sparse result must have: the class of `x`."""
    for name, ctor in _NAMED_TYPES.items():
        if isinstance(x, ctor):
            return ctor, name
    raise ValueError("Input matrices to dot_product_mkl must be CSR, CSC, or BSR; COO is not supported")


def _is_dense_vector(m_or_v):
    if _sps.issparse(m_or_v):
        return False
    return m_or_v.ndim == 1 or (m_or_v.ndim == 2 and min(m_or_v.shape) == 1)


def _is_double(arr):
    """(double precision?, complex?) of an array's dtype; anything else is a ValueError."""
    dt = _np.dtype(arr.dtype)
    if dt == _np.float32:
        return False, False
    if dt == _np.float64:
        return True, False
    if dt == _np.complex64:
        return False, True
    if dt == _np.complex128:
        return True, True
    raise ValueError("Only float32, float64, csingle, and cdouble dtypes are supported")


# ---- shape checks ---------------------------------------------------------------------------------
def _sanity_check(matrix_a, matrix_b, allow_vector=False):
    """Dimensionality and inner-dimension agreement of a product a @ b."""
    a2, b2 = matrix_a.ndim == 2, matrix_b.ndim == 2
    if not allow_vector and not (a2 and b2):
        raise ValueError("Matrices must be 2d: %s * %s is not valid" % (matrix_a.shape, matrix_b.shape))
    a_ok = a2 or _is_dense_vector(matrix_a)
    b_ok = b2 or _is_dense_vector(matrix_b)
    inner_a = matrix_a.shape[0] if matrix_a.ndim == 1 else matrix_a.shape[1]
    if not (a_ok and b_ok) or inner_a != matrix_b.shape[0]:
        raise ValueError("Matrix alignment error: %s * %s is not valid" % (matrix_a.shape, matrix_b.shape))


def _empty_output_check(matrix_a, matrix_b):
    """True when the product is trivially all-zero: a zero-length dimension or an empty sparse operand."""
    if min(tuple(matrix_a.shape) + tuple(matrix_b.shape)) == 0:
        return True
    for m in (matrix_a, matrix_b):
        if _sps.issparse(m) and min(m.data.size, m.indices.size) == 0:
            return True
        if getattr(m, "nnz", None) == 0 and not _sps.issparse(m) and not isinstance(m, _np.ndarray):
            return True  # an empty DeviceMatrix
    return False


# ---- dtype rules ----------------------------------------------------------------------------------
def _valid(dtype, kinds="fc"):
    dt = _np.dtype(dtype)
    return (("f" in kinds and dt in NUMPY_FLOAT_DTYPES) or ("c" in kinds and dt in NUMPY_COMPLEX_DTYPES))


def _cast_to(matrix, dtype):
    return matrix if matrix.dtype == dtype else matrix.astype(dtype)


def _type_check(matrix_a, matrix_b=None, cast=False, allow_complex=True):
    """Return operands whose dtypes the backend can multiply (see module docstring for the rule)."""
    n_complex = int(_np.iscomplexobj(matrix_a)) + int(_np.iscomplexobj(matrix_b))
    if not allow_complex and n_complex:
        raise ValueError("Complex datatypes are not supported")

    if matrix_b is None:
        if _valid(matrix_a.dtype):
            return matrix_a
        if cast:
            return _cast_to(matrix_a, _np.complex128 if n_complex else _np.float64)
        raise ValueError(
            "Matrix data type must be float32, float64, csingle, or cdouble; %s provided" % matrix_a.dtype)

    if _valid(matrix_a.dtype) and matrix_a.dtype == matrix_b.dtype:
        return matrix_a, matrix_b
    if not cast:
        raise ValueError(
            "Matrix data type must be float32, float64, csingle, or cdouble, and must be the same if "
            "cast=False; %s & %s provided" % (matrix_a.dtype, matrix_b.dtype))
    if n_complex == 0:
        target = _np.float64
    elif n_complex == 2:
        target = _np.complex128
    elif _valid(matrix_a.dtype, "c"):
        target = matrix_a.dtype
    elif _valid(matrix_b.dtype, "c"):
        target = matrix_b.dtype
    else:
        target = _np.complex128
    debug_print("Recasting matrix data types %s and %s to %s" % (matrix_a.dtype, matrix_b.dtype, _np.dtype(target)))
    return _cast_to(matrix_a, target), _cast_to(matrix_b, target)


def _mi_scalar(scalar, complex_type, double_precision):
    """Python scalar -> what the C ABI takes by value (None means 1.0)."""
    scalar = 1.0 if scalar is None else scalar
    if complex_type:
        return Complex16(scalar) if double_precision else Complex8(scalar)
    return float(scalar)


def _mi_beta(out, out_scalar, complex_type, double_precision):
    """beta of `alpha * op(A) @ B + beta * C`: with no user `out` there is nothing to accumulate into, so
    beta = 0 -- the library then neither uploads nor reads C (on CPU MKL `1.0 * zeros` was free, reference
    _common.py:869-882; here it would be a C-sized PCIe transfer and a read-modify-write per call)."""
    return _mi_scalar(0.0 if out is None else out_scalar, complex_type, double_precision)


# ---- dense layout / output array ----------------------------------------------------------------
def _get_numpy_layout(numpy_arr, second_arr=None):
    """(layout code, leading dimension) of a contiguous 2-d array.  An array that is both C and F
    contiguous (one row / one column) takes its order from `second_arr` when that is given."""
    is_c, is_f = numpy_arr.flags.c_contiguous, numpy_arr.flags.f_contiguous
    if not (is_c or is_f):
        raise ValueError("Array is not contiguous")
    if is_c and is_f and second_arr is not None:
        if second_arr.flags.c_contiguous:
            is_f = False
        elif second_arr.flags.f_contiguous:
            is_c = False
        else:
            raise ValueError("Array is not contiguous")
    if is_c:
        return LAYOUT_CODE_C, numpy_arr.shape[1]
    return LAYOUT_CODE_F, numpy_arr.shape[0]


def _describe(shape, dtype, order, contiguous=True):
    name = getattr(dtype, "__name__", None) or _np.dtype(dtype).name
    return "%s %s [%s_%s]" % (tuple(shape), name, order, "CONTIGUOUS" if contiguous else "NONCONTIGUOUS")


def _out_matrix(shape, dtype, order="C", out_arr=None, out_t=False, overwritten=False):
    """Fresh output, or the user's `out` after checking shape / dtype / order / contiguity.
    `out_t` says the caller handed us out.T, so the message is phrased in the caller's orientation.
    `overwritten`: the executor writes every element (beta = 0 products), so a fresh array need not be zeroed."""
    if out_arr is None:
        return _np.empty(shape, dtype=dtype, order=order) if overwritten else _np.zeros(shape, dtype=dtype, order=order)
    shape = tuple(shape)
    order_ok = out_arr.flags["C_CONTIGUOUS" if order == "C" else "F_CONTIGUOUS"]
    contiguous = out_arr.flags["C_CONTIGUOUS"] or out_arr.flags["F_CONTIGUOUS"]
    if tuple(out_arr.shape) == shape and out_arr.dtype == dtype and order_ok and contiguous:
        return out_arr
    have_order = "C" if out_arr.flags["C_CONTIGUOUS"] else "F"
    have_shape, need_shape, need_order = tuple(out_arr.shape), shape, order
    if out_t and out_arr.ndim != 1:
        have_shape, need_shape = have_shape[::-1], need_shape[::-1]
        have_order = "F" if have_order == "C" and not out_arr.flags["F_CONTIGUOUS"] else "C"
        need_order = "F" if order == "C" else "C"
    raise ValueError("Provided out array is %s and product requires %s" % (
        _describe(have_shape, out_arr.dtype, have_order, contiguous), _describe(need_shape, dtype, need_order)))


def _check_return_value(ret_val, func_name):
    """Non-zero status -> ValueError("<fn> returned <n> (<NAME>)"), like the reference
    (_common.py:645-668), with the backend's own reason appended."""
    if ret_val != 0:
        msg = "%s returned %d (%s)" % (func_name, ret_val, STATUS_NAMES.get(ret_val, "UNKNOWN"))
        detail = MI.last_error()
        if detail:
            msg += ": " + detail
        raise ValueError(msg)
    if MI.DEBUG:
        print("%s returned %d (%s)" % (func_name, ret_val, STATUS_NAMES[0]))
