"""ctypes boundary of the sparse_dot_amd package: library loader, symbol table, handle and
validation helpers, service functions (the analogue of the reference's _mkl_interface package)."""
import ctypes as _ct

import numpy as _np

from ._constants import *  # noqa: F401,F403
from ._library import MI, matrix_descr, sparse_matrix_t, mi_library_name, Complex8, Complex16  # noqa: F401
from ._checks import (  # noqa: F401
    set_debug_mode, print_mi_debug, debug_print, debug_timer, is_csr, is_csc, is_bsr,
    _is_allowed_sparse_format, sparse_output_type, _is_dense_vector, _is_double, _sanity_check,
    _empty_output_check, _type_check, _cast_to, _mi_scalar, _mi_beta, _get_numpy_layout, _out_matrix,
    _check_return_value, _output_dtypes, _type_letters, NUMPY_FLOAT_DTYPES, NUMPY_COMPLEX_DTYPES,
)
from ._handles import (  # noqa: F401
    SparseHandle, _create_mi_sparse, _export_mi, _destroy_mi_handle, _order_mi_handle, _convert_to_csr,
    DeviceMatrix, to_device,
)


# ---- service functions (analogues of mkl_get_version_string / mkl_get_max_threads / ...) ----------
def mi_get_version_string():
    buf = _ct.create_string_buffer(512)
    _check_return_value(MI.call("mi_sparse_get_version_string", buf, 512), "mi_sparse_get_version_string")
    return buf.value.decode("utf-8", "replace")


def mi_get_device_count():
    return int(MI.call("mi_sparse_get_device_count"))


def mi_set_device(device):
    _check_return_value(MI.call("mi_sparse_set_device", int(device)), "mi_sparse_set_device")


def mi_get_device():
    """Device this thread's library context is bound to (-1 before the first device call)."""
    return int(MI.call("mi_sparse_get_device"))


def mi_get_stream():
    """The hipStream_t (as an int, 0 = default stream) this thread's work is enqueued on."""
    p = _ct.c_void_p()
    _check_return_value(MI.call("mi_sparse_get_stream", _ct.byref(p)), "mi_sparse_get_stream")
    return p.value or 0


def mi_set_stream(stream_ptr):
    """Enqueue this thread's work on a caller-owned hipStream_t (e.g. torch.cuda.current_stream().cuda_stream)."""
    _check_return_value(MI.call("mi_sparse_set_stream", _ct.c_void_p(stream_ptr or 0)), "mi_sparse_set_stream")


def mi_synchronize():
    _check_return_value(MI.call("mi_sparse_synchronize"), "mi_sparse_synchronize")


def mi_set_option(name, value):
    _check_return_value(MI.call("mi_sparse_set_option", name.encode(), int(value)), "mi_sparse_set_option")


def mi_get_counter(name):
    v = _ct.c_double()
    _check_return_value(MI.call("mi_sparse_get_counter", name.encode(), _ct.byref(v)), "mi_sparse_get_counter")
    return v.value


def mi_probe_copy_gbs(nbytes=1 << 31, reps=3):
    """The device's copy rate in GB/s (read + write) by the library's tuned copy kernel (mi_sparse_probe_copy)."""
    v = _ct.c_double()
    _check_return_value(MI.call("mi_sparse_probe_copy", int(nbytes), int(reps), _ct.byref(v)), "mi_sparse_probe_copy")
    return v.value


def mi_get_last_kernel():
    """Name of the dominant kernel this thread launched last, as the library instantiated it."""
    buf = _ct.create_string_buffer(256)
    _check_return_value(MI.call("mi_sparse_get_last_kernel", buf, 256), "mi_sparse_get_last_kernel")
    return buf.value.decode()


def mi_interface_integer_dtype():
    """Index dtype of results built from int32 inputs (int64 inputs / huge results give int64)."""
    return _np.int32
