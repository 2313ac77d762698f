"""Loader for the golden vectors written by oracle/make_golden.py: tests/golden/golden_v1.npz (rounds 1-2) and
tests/golden/golden_spmv_v1.npz (`--spmv`, the sparse x vector cases added in round 3)."""
import json
import os

import numpy as np
import scipy.sparse as sps

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GOLDEN = os.path.join(_DIR, "golden_v1.npz")
GOLDEN_FILES = [GOLDEN, os.path.join(_DIR, "golden_spmv_v1.npz")]

_CLS = {
    ("csr", "matrix"): sps.csr_matrix, ("csr", "array"): sps.csr_array,
    ("csc", "matrix"): sps.csc_matrix, ("csc", "array"): sps.csc_array,
    ("bsr", "matrix"): sps.bsr_matrix, ("bsr", "array"): sps.bsr_array,
}

_cache = {}


def _load():
    if "z" not in _cache:
        arrays, merged = {}, None
        for path in GOLDEN_FILES:
            z = np.load(path)
            m = json.loads(bytes(z["__manifest__"]).decode("utf-8"))
            for k in z.files:
                if k != "__manifest__":
                    assert k not in arrays, k  # array keys carry the case name: unique across files
                    arrays[k] = z[k]
            if merged is None:
                merged = m
            else:
                merged["cases"] = merged["cases"] + m["cases"]
        _cache["z"] = arrays
        _cache["manifest"] = merged
    return _cache["z"], _cache["manifest"]


def manifest():
    return _load()[1]


def cases(prefix=None, raises=None):
    """List of manifest entries, optionally filtered by name prefix / whether they raise."""
    out = []
    for c in manifest()["cases"]:
        if prefix is not None and not c["name"].startswith(prefix):
            continue
        if raises is not None and bool(c["raises"]) != raises:
            continue
        out.append(c)
    return out


def operand(desc):
    """Rebuild a fresh (writable, correctly ordered) numpy / scipy object from its descriptor."""
    if desc is None:
        return None
    z, _ = _load()
    if desc["kind"] == "dense":
        x = np.array(z[desc["key"]])
        if desc["order"] == "F":
            x = np.asfortranarray(x.T)
        assert list(x.shape) == desc["shape"]
        return x
    if desc["fmt"] == "coo":
        return sps.coo_matrix((np.array(z[desc["data"]]), (np.array(z[desc["row"]]), np.array(z[desc["col"]]))),
                              shape=tuple(desc["shape"]))
    cls = _CLS[(desc["fmt"], desc["cls"])]
    parts = (np.array(z[desc["data"]]), np.array(z[desc["indices"]]), np.array(z[desc["indptr"]]))
    if desc["fmt"] == "bsr":
        return cls(parts, shape=tuple(desc["shape"]), blocksize=tuple(desc["blocksize"]))
    return cls(parts, shape=tuple(desc["shape"]))


def out_array(case):
    o = case["out"]
    if o is None:
        return None
    return np.full(tuple(o["shape"]), o["fill"], dtype=np.dtype(o["dtype"]), order=o["order"])


def tolerances(dtype):
    """(rtol, atol) the north_star states: fp64 1e-12 rel, fp32 1e-5 rel (atol scaled likewise)."""
    dt = np.dtype(dtype)
    if dt in (np.dtype(np.float32), np.dtype(np.complex64)):
        return 1e-5, 1e-5
    return 1e-12, 1e-12
