"""
Generate tests/golden/golden_v1.npz: inputs + outputs of the UNMODIFIED reference
(sparse_dot_mkl 0.9.6 on Intel oneMKL) for the dot_product_mkl / gram_matrix_mkl hot path,
plus scipy's answer for the same inputs.

Run in the build container only (the reference and MKL do not exist on the GPU box):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference \
        MKL_RT=/opt/conda/lib/libmkl_rt.so python /root/repo/oracle/make_golden.py

Only DATA is written (arrays and a JSON manifest describing each call); no reference source,
bytecode or test text is stored.  `scipy.sparse.random(random_state=int)` streams depend on the
scipy version, so the inputs are stored as arrays and never regenerated from seeds.

Manifest entry:  {"name", "fn": "dot"|"gram", "a": <operand>, "b": <operand>|null,
                  "kwargs": {...}, "out": null | {"fill": 1.0, "order": "C"|"F", "dtype": ...},
                  "result": <operand>|null, "scipy": <operand>|null, "raises": null|"ValueError"}
Operand:         {"kind": "dense", "key", "order", "shape"}  (F-order arrays are stored transposed) or
                 {"kind": "sparse", "fmt", "cls": "matrix"|"array", "shape", "blocksize",
                  "data", "indices", "indptr"}  (values are npz keys; identical arrays are stored once)
"""
import json
import os
import sys

import numpy as np
import scipy.sparse as sps

import sparse_dot_mkl as ref  # the reference (PYTHONPATH=/root/reference)

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
OUT = os.path.join(GOLDEN_DIR, "golden_v1.npz")
OUT_SPMV = os.path.join(GOLDEN_DIR, "golden_spmv_v1.npz")  # `make_golden.py --spmv`: the sparse x vector cases (round 3)

ARR = {}
CASES = []
_SEEN = {}


def _store(key, x):
    """Store array once; identical content is aliased to the first key that held it."""
    import hashlib
    x = np.ascontiguousarray(x)
    h = (hashlib.sha1(x.tobytes()).hexdigest(), x.dtype.str, x.shape)
    if h in _SEEN:
        return _SEEN[h]
    _SEEN[h] = key
    ARR[key] = x
    return key


def put_dense(key, x):
    x = np.asarray(x)
    order = "F" if (x.ndim == 2 and x.flags.f_contiguous and not x.flags.c_contiguous) else "C"
    key = _store(key, x if order == "C" else x.T)
    return {"kind": "dense", "key": key, "order": order, "shape": list(x.shape)}


def put_sparse(key, m):
    if m.format == "coo":
        return {"kind": "sparse", "fmt": "coo", "cls": "matrix", "shape": list(m.shape),
                "blocksize": None, "data": _store(key + "__data", m.data),
                "row": _store(key + "__row", m.row), "col": _store(key + "__col", m.col)}
    return {
        "kind": "sparse", "fmt": m.format,
        "data": _store(key + "__data", m.data), "indices": _store(key + "__indices", m.indices),
        "indptr": _store(key + "__indptr", m.indptr),
        "cls": "array" if isinstance(m, sps.sparray) else "matrix",
        "shape": list(m.shape),
        "blocksize": list(m.blocksize) if m.format == "bsr" else None,
    }


def put(key, x):
    if x is None:
        return None
    return put_sparse(key, x) if sps.issparse(x) else put_dense(key, x)


def canon(m):
    """Canonical form of a sparse result: sorted indices (values untouched, zeros kept)."""
    m = m.copy()
    m.sort_indices()
    return m


def scipy_answer(fn, a, b, kwargs):
    if fn == "dot":
        r = a @ b
        if sps.issparse(r):
            return canon(r) if not kwargs.get("dense") else np.asarray(r.todense())
        return np.asarray(r)
    # gram: upper triangle of A^T A (or A A^T)
    aa = a if sps.issparse(a) else np.asarray(a)
    full = (aa @ aa.T) if kwargs.get("transpose") else (aa.T @ aa)
    full = np.asarray(full.todense()) if sps.issparse(full) else np.asarray(full)
    return np.triu(full)


def add(name, fn, a, b=None, out=None, raises=None, **kwargs):
    entry = {"name": name, "fn": fn, "kwargs": kwargs, "out": None, "raises": raises}
    entry["a"] = put(name + "/a", a)
    entry["b"] = put(name + "/b", b)
    call_kwargs = dict(kwargs)
    if out is not None:
        fill, order, dtype = out[:3]
        shape = tuple(out[3]) if len(out) > 3 else out_shape(fn, a, b, kwargs)
        call_kwargs["out"] = np.full(shape, fill, dtype=dtype, order=order)
        entry["out"] = {"fill": fill, "order": order, "dtype": np.dtype(dtype).name, "shape": list(shape)}
    # the reference may mutate index dtypes / ordering of its inputs in place: hand it copies
    ra = a.copy() if sps.issparse(a) else a.copy(order="K")
    rb = None if b is None else (b.copy() if sps.issparse(b) else b.copy(order="K"))
    try:
        res = ref.dot_product_mkl(ra, rb, **call_kwargs) if fn == "dot" else ref.gram_matrix_mkl(ra, **call_kwargs)
    except ValueError as e:
        if raises != "ValueError":
            raise
        entry["result"] = None
        entry["scipy"] = None
        entry["message"] = str(e)[:200]
        CASES.append(entry)
        return
    assert raises is None, name
    if sps.issparse(res):
        res = canon(res)
    entry["result"] = put(name + "/result", res)
    try:
        sc = scipy_answer(fn, a, b, kwargs)
        if out is not None and not sps.issparse(sc):
            beta = kwargs.get("out_scalar", None)
            beta = 1.0 if beta is None else beta
            if fn == "dot" and sps.issparse(a) and sps.issparse(b):
                pass  # spmmd overwrites `out`
            elif fn == "gram":
                sc = sc + np.triu(np.full(sc.shape, out[0] * beta))
            else:
                sc = sc + out[0] * beta
        # dense answers are recomputed by the tests from the inputs; only sparse structure is stored
        entry["scipy"] = put(name + "/scipy", sc) if sps.issparse(sc) else None
        # flag cases where the reference's own answer is wrong (vs scipy/numpy in float64); the
        # tests do not pin the build to those (e.g. float64 CSC + cast + dense gram returns zeros:
        # use-after-free of the converted handle, reference _gram_matrix.py:126-127)
        if not sps.issparse(sc) and not sps.issparse(res):
            r2, s2 = np.asarray(res), np.asarray(sc)
            if fn == "gram":
                r2, s2 = np.triu(r2), np.triu(s2)
            if r2.shape != s2.shape or not np.allclose(r2, s2, rtol=1e-4, atol=1e-4):
                entry["reference_deviates"] = True
                print("NOTE: reference deviates from scipy on", name)
    except Exception:  # scipy cannot do it (never expected)
        entry["scipy"] = None
    CASES.append(entry)


def out_shape(fn, a, b, kwargs):
    if fn == "gram":
        n = a.shape[0] if kwargs.get("transpose") else a.shape[1]
        return (n, n)
    return (a.shape[0], b.shape[1])


def main():
    SEED = 86
    m1 = sps.random(60, 90, density=0.08, format="csr", dtype=np.float64, random_state=SEED)
    m2 = sps.random(90, 40, density=0.08, format="csr", dtype=np.float64, random_state=SEED + 1)
    d1 = np.asarray(m1.todense())
    d2 = np.asarray(m2.todense())

    # ---- SpMM: sparse x dense and dense x sparse -------------------------------------------
    for dt in (np.float32, np.float64):
        tag = np.dtype(dt).name
        for order in ("C", "F"):
            for fmt in ("csr", "csc", "bsr"):
                a = m1.astype(dt).asformat(fmt) if fmt != "bsr" else m1.astype(dt).tobsr(blocksize=(10, 10))
                b = np.asarray(d2.astype(dt), order=order)
                add(f"spmm/{tag}/{order}/{fmt}/a_sparse", "dot", a, b)
                add(f"spmm/{tag}/{order}/{fmt}/a_sparse_out", "dot", a, b, out=(1.0, order, dt), out_scalar=3.0)
                s = m2.astype(dt).asformat(fmt) if fmt != "bsr" else m2.astype(dt).tobsr(blocksize=(10, 10))
                d = np.asarray(d1.astype(dt), order=order)
                add(f"spmm/{tag}/{order}/{fmt}/b_sparse", "dot", d, s)
                add(f"spmm/{tag}/{order}/{fmt}/b_sparse_out", "dot", d, s, out=(1.0, order, dt), out_scalar=3.0)
    # csr_array class, mixed precision with cast, integer data with cast
    add("spmm/cast/f32xf64", "dot", m1.astype(np.float32), d2, cast=True)
    add("spmm/cast/int", "dot", sps.csr_matrix((m1 * 10).astype(np.int32)), d2.astype(np.float64), cast=True)
    add("spmm/array_cls", "dot", sps.csr_array(m1), d2)
    add("spmm/nocast_raises", "dot", m1.astype(np.float32), d2, raises="ValueError")
    add("spmm/misaligned_raises", "dot", m1, d2[:-1], raises="ValueError")
    add("spmm/bad_out_raises", "dot", m1, d2, out=(1.0, "F", np.float64), raises="ValueError")
    # one-row / one-column sparse operands (kept 2-D dense so the SpMM branch is taken)
    add("spmm/one_row", "dot", m1[0:1, :].tocsr(), d2)
    add("spmm/wide_n", "dot", m1, np.asarray(np.random.default_rng(5).random((90, 257))))
    add("spmm/empty_sparse", "dot", sps.csr_matrix((60, 90), dtype=np.float64), d2)
    # complex
    rng = np.random.default_rng(11)
    mc = m1.astype(np.complex128)
    mc.data = mc.data + 1j * rng.random(mc.data.shape)
    dc = d2 + 1j * rng.random(d2.shape) * (d2 != 0)
    for order in ("C", "F"):
        add(f"spmm/complex128/{order}/a_sparse", "dot", mc, np.asarray(dc, order=order))
        add(f"spmm/complex128/{order}/b_sparse", "dot", np.asarray(np.asarray(mc.todense()), order=order),
            sps.csr_matrix(dc))
    add("spmm/complex64/C/a_sparse", "dot", mc.astype(np.complex64), dc.astype(np.complex64))

    # ---- SpGEMM ------------------------------------------------------------------------------
    for dt in (np.float32, np.float64):
        tag = np.dtype(dt).name
        a, b = m1.astype(dt), m2.astype(dt)
        add(f"spgemm/{tag}/csr", "dot", a, b, reorder_output=True)
        add(f"spgemm/{tag}/csr_unordered", "dot", a, b)
        add(f"spgemm/{tag}/csc", "dot", a.tocsc(), b.tocsc(), reorder_output=True)
        add(f"spgemm/{tag}/csr_x_csc", "dot", a, b.tocsc(), reorder_output=True)
        add(f"spgemm/{tag}/bsr", "dot", a.tobsr(blocksize=(10, 10)), b.tobsr(blocksize=(10, 10)))
        add(f"spgemm/{tag}/dense", "dot", a, b, dense=True)
        add(f"spgemm/{tag}/dense_out", "dot", a, b, dense=True, out=(7.0, "C", dt))
    lo_a = sps.random(2000, 3000, density=5e-4, format="csr", random_state=SEED)
    lo_b = sps.random(3000, 1000, density=5e-4, format="csr", random_state=SEED + 1)
    add("spgemm/low_density", "dot", lo_a, lo_b, reorder_output=True)
    vlo_a = sps.random(2000, 3000, density=5e-6, format="csr", random_state=SEED)
    vlo_b = sps.random(3000, 1000, density=5e-6, format="csr", random_state=SEED + 1)
    add("spgemm/very_low_density", "dot", vlo_a, vlo_b, reorder_output=True)
    fa = sps.random(10, 50, density=1.0, format="csr", random_state=SEED)
    fb = sps.random(50, 20, density=1.0, format="csr", random_state=SEED + 1)
    add("spgemm/full_density", "dot", fa, fb, reorder_output=True)
    add("spgemm/all_zero", "dot", sps.csr_matrix((50, 100), dtype=np.float64),
        sps.csr_matrix((100, 20), dtype=np.float64))
    # cancellation: MKL keeps the explicit 0.0, scipy prunes it
    ca = sps.csr_matrix(np.array([[1.0, -1.0, 0.0], [0.0, 2.0, 0.0]]))
    cb = sps.csr_matrix(np.array([[1.0, 0.0], [1.0, 0.0], [0.0, 3.0]]))
    add("spgemm/cancellation", "dot", ca, cb, reorder_output=True)
    # unsorted indices with duplicates in the inputs
    ua = sps.csr_matrix((np.array([1.0, 2.0, 3.0, 4.0, 5.0]), np.array([2, 0, 2, 1, 0]),
                         np.array([0, 3, 5])), shape=(2, 3))
    ub = sps.csr_matrix((np.array([1.0, 2.0, 3.0, 4.0]), np.array([1, 0, 1, 1]), np.array([0, 2, 3, 4])),
                        shape=(3, 2))
    add("spgemm/unsorted_dup", "dot", ua, ub, reorder_output=True)
    add("spgemm/coo_raises", "dot", m1.tocoo(), m2, raises="ValueError")
    add("spgemm/out_without_dense_raises", "dot", m1, m2, out=(0.0, "C", np.float64), raises="ValueError")
    add("spgemm/cast/f32xf64", "dot", m1.astype(np.float32), m2, cast=True, reorder_output=True)
    add("spgemm/complex128", "dot", mc, sps.csr_matrix(dc), reorder_output=True)
    add("spgemm/array_cls", "dot", sps.csr_array(m1), sps.csr_array(m2), reorder_output=True)

    # ---- Gram --------------------------------------------------------------------------------
    for dt in (np.float32, np.float64):
        tag = np.dtype(dt).name
        g = m2.astype(dt)  # 300 x 100
        for tr in (False, True):
            t = "aat" if tr else "ata"
            add(f"gram/{tag}/{t}/sparse", "gram", g, transpose=tr, reorder_output=True)
            add(f"gram/{tag}/{t}/dense", "gram", g, transpose=tr, dense=True)
            add(f"gram/{tag}/{t}/dense_out", "gram", g, transpose=tr, dense=True,
                out=(1.0, "C", dt), out_scalar=1.0)
            add(f"gram/{tag}/{t}/dense_in_C", "gram", np.asarray(g.todense(), order="C"), transpose=tr)
            add(f"gram/{tag}/{t}/dense_in_F", "gram", np.asarray(g.todense(), order="F"), transpose=tr)
        add(f"gram/{tag}/csc_cast", "gram", g.tocsc(), cast=True, reorder_output=True)
        add(f"gram/{tag}/csc_cast_dense", "gram", g.tocsc(), cast=True, dense=True)
    add("gram/csc_nocast_raises", "gram", m2.tocsc(), raises="ValueError")
    add("gram/out_sparse_raises", "gram", m2, out=(0.0, "C", np.float64), raises="ValueError")
    add("gram/complex_raises", "gram", mc, raises="ValueError")

    # ---- dense x dense -----------------------------------------------------------------------
    for dt in (np.float32, np.float64):
        tag = np.dtype(dt).name
        for oa in ("C", "F"):
            for ob in ("C", "F"):
                add(f"gemm/{tag}/{oa}{ob}", "dot", np.asarray(d1.astype(dt), order=oa),
                    np.asarray(d2.astype(dt), order=ob))
        add(f"gemm/{tag}/out", "dot", d1.astype(dt), d2.astype(dt), out=(1.0, "C", dt), out_scalar=3.0)

    write(OUT)


def write(path):
    manifest = json.dumps({"version": 1, "cases": CASES, "generator": "oracle/make_golden.py",
                           "reference": "sparse_dot_mkl " + ref.__version__,
                           "mkl": ref.mkl_get_version_string(),
                           "numpy": np.__version__, "scipy": __import__("scipy").__version__})
    ARR["__manifest__"] = np.frombuffer(manifest.encode("utf-8"), dtype=np.uint8)
    np.savez_compressed(path, **ARR)
    print("wrote", os.path.abspath(path), "cases:", len(CASES), "bytes:", os.path.getsize(path))


def main_spmv():
    """Sparse x dense VECTOR (reference _sparse_vector.py:28-174 -> mkl_sparse_?_mv; dispatcher sparse_dot.py:96-121):
    1-D vs (n, 1) vs (1, n) vectors, vector on the left (op = T), out / out_scalar, CSR / CSC / BSR, s / d / c / z."""
    SEED = 86
    m1 = sps.random(60, 90, density=0.08, format="csr", dtype=np.float64, random_state=SEED)
    rng = np.random.default_rng(11)
    mc = (m1 + 1j * sps.random(60, 90, density=0.08, format="csr", dtype=np.float64, random_state=SEED + 7)).tocsr()
    for dt in (np.float32, np.float64, np.complex64, np.complex128):
        tag = np.dtype(dt).name
        cplx = np.dtype(dt).kind == "c"
        base = (mc if cplx else m1).astype(dt)
        vr = rng.random(90) + (1j * rng.random(90) if cplx else 0)   # right-hand vector, len = columns
        vl = rng.random(60) + (1j * rng.random(60) if cplx else 0)   # left-hand vector, len = rows
        vr, vl = vr.astype(dt), vl.astype(dt)
        for fmt in ("csr", "csc", "bsr"):
            a = base.asformat(fmt) if fmt != "bsr" else base.tobsr(blocksize=(10, 10))
            add(f"spmv/{tag}/{fmt}/right_1d", "dot", a, vr.copy())
            add(f"spmv/{tag}/{fmt}/right_col", "dot", a, vr.reshape(-1, 1).copy())
            add(f"spmv/{tag}/{fmt}/left_1d", "dot", vl.copy(), a)
            add(f"spmv/{tag}/{fmt}/left_row", "dot", vl.reshape(1, -1).copy(), a)
            if fmt != "bsr":
                add(f"spmv/{tag}/{fmt}/right_1d_out", "dot", a, vr.copy(), out=(1.0, "C", dt, (60,)), out_scalar=3.0)
                add(f"spmv/{tag}/{fmt}/right_col_out", "dot", a, vr.reshape(-1, 1).copy(), out=(1.0, "C", dt, (60, 1)),
                    out_scalar=3.0)
                add(f"spmv/{tag}/{fmt}/left_1d_out", "dot", vl.copy(), a, out=(2.0, "C", dt, (90,)), out_scalar=-0.5)
                add(f"spmv/{tag}/{fmt}/left_row_out", "dot", vl.reshape(1, -1).copy(), a, out=(2.0, "C", dt, (1, 90)),
                    out_scalar=-0.5)
    v64 = rng.random(90)
    add("spmv/cast/f32xf64", "dot", m1.astype(np.float32), v64.copy(), cast=True)
    add("spmv/cast/f64xf32_left", "dot", rng.random(60).astype(np.float32), m1, cast=True)
    add("spmv/nocast_raises", "dot", m1.astype(np.float32), v64.copy(), raises="ValueError")
    add("spmv/misaligned_raises", "dot", m1, v64[:-1].copy(), raises="ValueError")
    add("spmv/array_cls", "dot", sps.csr_array(m1), v64.copy())
    add("spmv/empty_sparse", "dot", sps.csr_matrix((60, 90), dtype=np.float64), v64.copy())
    add("spmv/one_row", "dot", m1[0:1, :].tocsr(), v64.copy())
    add("spmv/one_col", "dot", m1[:, 0:1].tocsr(), v64[:1].copy())
    hub = sps.random(300, 2000, density=0.004, format="lil", dtype=np.float64, random_state=5)
    hub[7, :] = rng.random(2000)          # one row holding every column (cut across chunks on the GPU)
    hub = hub.tocsr()
    add("spmv/hub_row", "dot", hub, rng.random(2000))
    add("spmv/hub_row_left", "dot", rng.random(300), hub)
    write(OUT_SPMV)


if __name__ == "__main__":
    sys.exit(main_spmv() if "--spmv" in sys.argv else main())
