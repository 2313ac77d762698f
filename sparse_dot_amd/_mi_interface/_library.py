"""
Locate and bind libmi_sparse.so (the HIP / gfx950 backend) with ctypes.

Mirrors the role of the reference's loader + symbol table
(reference sparse_dot_mkl/_mkl_interface/_load_library.py:31-96 and _cfunctions.py:32-705):
the library is chosen by an environment variable first ($MI_SPARSE_RT, the analogue of $MKL_RT),
then the in-tree build next to this package.  A missing library is an ImportError -- there is no
CPU fallback.  HIP itself is only initialised by the first call that needs the device.
"""
import ctypes as _ct
import os as _os

_PKG_DIR = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
_DEFAULT_LIB = _os.path.join(_PKG_DIR, "libmi_sparse.so")


def _find_library():
    env = _os.environ.get("MI_SPARSE_RT")
    if env:
        if not _os.path.exists(env):
            raise ImportError("MI_SPARSE_RT=%r does not exist" % env)
        return env
    if _os.path.exists(_DEFAULT_LIB):
        return _DEFAULT_LIB
    raise ImportError(
        "libmi_sparse.so not found next to the sparse_dot_amd package (%s); build it with "
        "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C sparse_dot_amd/csrc`, "
        "or point MI_SPARSE_RT at it.  There is no CPU fallback." % _DEFAULT_LIB
    )


def _share_torch_hip_runtime():
    """When PyTorch-ROCm is installed it bundles its own copy of the HIP runtime (libamdhip64).
    Two copies of that runtime in one process cannot both own the GPU (the second one reports
    "no devices"), and which copy wins would depend on import order.  So, if a torch install is
    present, its copy is loaded FIRST (without importing torch); libmi_sparse.so's
    `NEEDED libamdhip64.so.7` then binds to that same object and torch may be imported before or
    after this package.  Without torch the system runtime (/opt/rocm) is used.  Opt out with
    MI_SPARSE_NO_TORCH_HIP=1."""
    if _os.environ.get("MI_SPARSE_NO_TORCH_HIP"):
        return None
    try:
        import importlib.util as _ilu
        spec = _ilu.find_spec("torch")
        if spec is None or not spec.origin:
            return None
        cand = _os.path.join(_os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if _os.path.exists(cand):
            return _ct.CDLL(cand, mode=_ct.RTLD_GLOBAL)
    except Exception:
        return None
    return None


_torch_hip = _share_torch_hip_runtime()


class _OpaqueMatrix(_ct.Structure):
    pass


sparse_matrix_t = _ct.POINTER(_OpaqueMatrix)


class matrix_descr(_ct.Structure):
    """struct mi_matrix_descr, passed by value (general matrix = {20, 0, 0})."""
    _fields_ = [("type", _ct.c_int), ("mode", _ct.c_int), ("diag", _ct.c_int)]

    def __init__(self, type=20, mode=0, diag=0):
        super().__init__(type, mode, diag)


class Complex8(_ct.Structure):
    _fields_ = [("real", _ct.c_float), ("imag", _ct.c_float)]

    def __init__(self, z=0j):
        z = complex(z)
        super().__init__(z.real, z.imag)


class Complex16(_ct.Structure):
    _fields_ = [("real", _ct.c_double), ("imag", _ct.c_double)]

    def __init__(self, z=0j):
        z = complex(z)
        super().__init__(z.real, z.imag)


# letter -> (C scalar type passed by value)
SCALAR_CTYPE = {"s": _ct.c_float, "d": _ct.c_double, "c": Complex8, "z": Complex16}

_i64 = _ct.c_int64
_vp = _ct.c_void_p
_int = _ct.c_int


def _bind(lib):
    """Attach argtypes / restype to every entry point declared in include/mi_sparse.h."""
    H = sparse_matrix_t
    HP = _ct.POINTER(sparse_matrix_t)
    table = {}

    def add(name, argtypes, restype=_int):
        fn = getattr(lib, name)  # AttributeError here == the .so does not match the header
        fn.argtypes = argtypes
        fn.restype = restype
        table[name] = fn

    for t in "sdcz":
        sc = SCALAR_CTYPE[t]
        for sfx in ("", "_64"):
            add("mi_sparse_%s_create_csr%s" % (t, sfx), [HP, _int, _i64, _i64, _vp, _vp, _vp, _vp])
            add("mi_sparse_%s_create_csc%s" % (t, sfx), [HP, _int, _i64, _i64, _vp, _vp, _vp, _vp])
            add("mi_sparse_%s_create_bsr%s" % (t, sfx), [HP, _int, _int, _i64, _i64, _i64, _vp, _vp, _vp, _vp])
            add("mi_sparse_%s_export_bsr%s" % (t, sfx),
                [H, _ct.POINTER(_int), _ct.POINTER(_int), _vp, _vp, _vp, _ct.POINTER(_vp), _ct.POINTER(_vp), _ct.POINTER(_vp),
                 _ct.POINTER(_vp)])
            for fmt in ("csr", "csc"):
                add("mi_sparse_%s_export_%s%s" % (t, fmt, sfx),
                    [H, _ct.POINTER(_int), _vp, _vp, _ct.POINTER(_vp), _ct.POINTER(_vp), _ct.POINTER(_vp),
                     _ct.POINTER(_vp)])
        add("mi_sparse_%s_mm" % t, [_int, sc, H, matrix_descr, _int, _vp, _i64, _i64, sc, _vp, _i64])
        add("mi_sparse_%s_mv" % t, [_int, sc, H, matrix_descr, _vp, sc, _vp])
        add("mi_sparse_%s_spmmd" % t, [_int, H, H, _int, _vp, _i64])
        add("mi_sparse_%s_set_values" % t, [H, _vp])
    for t in "sd":
        sc = SCALAR_CTYPE[t]
        add("mi_sparse_%s_syrkd" % t, [_int, H, sc, sc, _vp, _int, _i64])
        add("mi_sparse_%s_syprd" % t, [_int, H, _vp, _int, _i64, sc, sc, _vp, _int, _i64])
        add("mi_sparse_%s_syrkd_rows" % t, [_int, H, sc, sc, _vp, _int, _i64, _i64, _i64])
        add("mi_cblas_%sgemm" % t, [_int, _int, _int, _i64, _i64, _i64, sc, _vp, _i64, _vp, _i64, sc, _vp, _i64])
        add("mi_cblas_%ssyrk" % t, [_int, _int, _int, _i64, _i64, sc, _vp, _i64, sc, _vp, _i64])
    for t in "cz":
        add("mi_cblas_%sgemm" % t, [_int, _int, _int, _i64, _i64, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _i64])
    add("mi_sparse_destroy", [H])
    add("mi_sparse_order", [H])
    add("mi_sparse_optimize", [H])
    add("mi_sparse_convert_csr", [H, _int, HP])
    add("mi_sparse_spmm", [_int, H, H, HP])
    add("mi_sparse_spmm_ordered", [_int, H, H, HP])
    add("mi_sparse_syrk", [_int, H, HP])
    add("mi_sparse_sp2m", [_int, matrix_descr, H, _int, matrix_descr, H, _int, HP])
    add("mi_sparse_sypr", [_int, H, H, matrix_descr, HP, _int])
    add("mi_sparse_get_info", [H, _ct.POINTER(_i64), _ct.POINTER(_i64), _ct.POINTER(_i64), _ct.c_char_p,
                               _ct.POINTER(_int)])
    add("mi_sparse_copy_out", [H, _int, _int, _vp, _vp, _vp])
    add("mi_sparse_get_device_csr", [H, _ct.POINTER(_vp), _ct.POINTER(_vp), _ct.POINTER(_vp)])
    add("mi_sparse_get_version_string", [_ct.c_char_p, _int])
    add("mi_sparse_get_device_count", [], _int)
    add("mi_sparse_set_device", [_int])
    add("mi_sparse_get_device", [], _int)
    add("mi_sparse_set_stream", [_vp])
    add("mi_sparse_get_stream", [_ct.POINTER(_vp)])
    add("mi_sparse_synchronize", [])
    add("mi_sparse_last_error", [], _ct.c_char_p)
    add("mi_sparse_set_option", [_ct.c_char_p, _i64])
    add("mi_sparse_get_counter", [_ct.c_char_p, _ct.POINTER(_ct.c_double)])
    add("mi_sparse_probe_copy", [_i64, _int, _ct.POINTER(_ct.c_double)])
    add("mi_sparse_get_last_kernel", [_ct.c_char_p, _int])
    return table


class MI:
    """Symbol table of the backend (the analogue of the reference's `class MKL`)."""
    DEBUG = False
    lib_path = _find_library()
    lib = _ct.CDLL(lib_path)  # cdll: the GIL is released for the duration of every call
    fn = _bind(lib)

    @classmethod
    def call(cls, name, *args):
        return cls.fn[name](*args)

    @classmethod
    def last_error(cls):
        msg = cls.fn["mi_sparse_last_error"]()
        return msg.decode("utf-8", "replace") if msg else ""


def mi_library_name():
    return MI.lib_path
