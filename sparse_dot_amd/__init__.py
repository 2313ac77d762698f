"""
sparse_dot_amd -- MI355X (gfx950 / CDNA4) backend for the dot_product_mkl / gram_matrix_mkl hot
path of flatironinstitute/sparse_dot, behind the reference's own Python signatures.

    from sparse_dot_amd import dot_product_mkl, gram_matrix_mkl

All arithmetic runs in hand-written HIP kernels inside libmi_sparse.so (C ABI: include/mi_sparse.h).
There is no CPU fallback: without the library the import fails, without a HIP device the first
product raises.
"""
__version__ = "0.1.0"

from .sparse_dot import dot_product_mkl, dot_product_transpose_mkl, gram_matrix_mkl, set_debug_mode  # noqa: F401
from ._mi_interface import (  # noqa: F401
    mi_get_version_string, mi_get_device_count, mi_set_device, mi_set_stream, mi_synchronize, mi_set_option,
    mi_get_counter, mi_get_last_kernel, mi_probe_copy_gbs, mi_interface_integer_dtype, DeviceMatrix, to_device,
)

from ._sparse_sypr import _sparse_sypr as sparse_sypr  # noqa: F401,E402  (reference _sparse_sypr.py, dead upstream)
from ._sparse_staged import StagedProduct  # noqa: F401,E402

get_version_string = mi_get_version_string
