#!/usr/bin/env python
"""Child process of bench.py's cpu_baseline legs for the MKL entry points added in round 6 (mkl_sparse_?_mv, mkl_sparse_syrk,
mkl_sparse_?_spmmd): oneMKL 2021.4's mkl_sparse_syrk aborted with heap corruption ("corrupted size vs. prev_size") when it was
called inside bench.py's own process -- next to torch's OpenMP runtime -- and ran cleanly on the same operand in a process of its
own, so these calls are timed here: operands from .npz files, one JSON line out.

    python tools/mkl_child.py mv a.npz [reps]      |  syrk a.npz [reps]  |  spmmd a.npz b.npz [reps]"""
import json
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)


def load(path):
    import numpy as np
    import scipy.sparse as sps
    z = np.load(path)
    return sps.csr_matrix((z["data"], z["indices"], z["indptr"]), shape=tuple(z["shape"]))


def main():
    import numpy as np
    from oracle import mkl_shim
    op = sys.argv[1]
    files = [a for a in sys.argv[2:] if a.endswith(".npz")]
    reps = [int(a) for a in sys.argv[2:] if a.isdigit()]
    reps = reps[0] if reps else 3
    mkl = mkl_shim.MklSpmm()
    a = load(files[0])
    h = mkl.make(a)
    if op == "syrk":
        fn = lambda: mkl.syrk(h)  # noqa: E731
    elif op == "mv":
        x = np.random.default_rng(0).random(a.shape[1]).astype(a.dtype)
        y = np.zeros(a.shape[0], dtype=a.dtype)
        fn = lambda: mkl.mv(h, x, y)  # noqa: E731
    elif op == "spmmd":
        b = load(files[1])
        hb = mkl.make(b)
        out = np.zeros((a.shape[0], b.shape[1]), dtype=a.dtype)
        fn = lambda: mkl.spmmd(h, hb, out)  # noqa: E731
    else:
        raise SystemExit("unknown op " + op)
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    print(json.dumps({"ms": ts[len(ts) // 2] * 1e3, "cores": mkl.threads(), "version": mkl.version()}), flush=True)


if __name__ == "__main__":
    main()
