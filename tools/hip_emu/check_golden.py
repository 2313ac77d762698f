"""Developer tool (NOT a test, NOT a product path): replay the golden cases through the
host-thread emulation build of the kernels (tools/hip_emu/build_emu.sh) to debug kernel logic
without a GPU.     MI_SPARSE_RT=/tmp/libmi_sparse_emu.so python tools/hip_emu/check_golden.py [prefix]"""
import os
import sys
import traceback

if not os.environ.get("CHECK_GOLDEN_REAL"):  # default: the emulation build; CHECK_GOLDEN_REAL=1 replays on the real GPU library
    os.environ.setdefault("MI_SPARSE_RT", "/tmp/libmi_sparse_emu.so")
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import scipy.sparse as sps

import golden_util as G
import sparse_dot_amd as sda

print(sda.mi_get_version_string())
assert os.environ.get("CHECK_GOLDEN_REAL") or "EMULATED" in sda.mi_get_version_string()
prefix = sys.argv[1] if len(sys.argv) > 1 else None
bad = n = 0
for c in G.cases(prefix=prefix):
    a, b, out = G.operand(c["a"]), G.operand(c["b"]), G.out_array(c)
    kw = dict(c["kwargs"])
    if out is not None:
        kw["out"] = out
    try:
        got = sda.dot_product_mkl(a, b, **kw) if c["fn"] == "dot" else sda.gram_matrix_mkl(a, **kw)
    except ValueError as e:
        if c["raises"]:
            n += 1
            continue
        print("UNEXPECTED ValueError", c["name"], e)
        bad += 1
        continue
    except Exception:
        print("EXC", c["name"])
        traceback.print_exc()
        bad += 1
        continue
    if c["raises"]:
        print("DID NOT RAISE", c["name"])
        bad += 1
        continue
    n += 1
    if c.get("reference_deviates"):
        continue
    exp = G.operand(c["result"])
    rt, at = G.tolerances(exp.dtype)
    if sps.issparse(exp):
        ok = sps.issparse(got) and got.format == exp.format and type(got) is type(exp) and got.shape == exp.shape
        if ok:
            g = got.copy()
            g.sort_indices()
            ok = (np.array_equal(g.indptr, exp.indptr) and np.array_equal(g.indices, exp.indices)
                  and np.allclose(g.data, exp.data, rtol=rt, atol=at))
    else:
        ok = isinstance(got, np.ndarray) and got.shape == exp.shape and got.dtype == exp.dtype
        if ok:
            if c["fn"] == "gram":
                iu = np.triu_indices(exp.shape[0])
                ok = np.allclose(got[iu], exp[iu], rtol=rt, atol=at)
            else:
                ok = np.allclose(got, exp, rtol=rt, atol=at)
        if ok and out is not None:
            ok = got is out
    if not ok:
        print("MISMATCH", c["name"], type(got), getattr(got, "shape", None), getattr(got, "dtype", None))
        if isinstance(got, np.ndarray) and isinstance(exp, np.ndarray) and got.shape == exp.shape:
            d = np.abs(got - exp)
            bad_idx = np.argwhere(d > 1e-4 * np.maximum(np.abs(exp), 1))
            print("   max abs diff", d.max(), "n bad", len(bad_idx), "first bad", bad_idx[:6].tolist())
        bad += 1
print("%d cases, %d bad" % (n, bad))
sys.exit(1 if bad else 0)
