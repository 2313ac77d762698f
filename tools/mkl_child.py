#!/usr/bin/env python
"""Child process of bench.py's cpu_baseline legs for MKL entry points that can take the whole process down (mkl_sparse_syrk of
oneMKL 2021.4 aborts with heap corruption on the 2^20 x 2^18 operand of secondary.gram_sparse): times the call on an operand read
from an .npz and prints one JSON line.   python tools/mkl_child.py syrk operand.npz [reps]"""
import json
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import scipy.sparse as sps
    from oracle import mkl_shim
    op, path = sys.argv[1], sys.argv[2]
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    z = np.load(path)
    a = sps.csr_matrix((z["data"], z["indices"], z["indptr"]), shape=tuple(z["shape"]))
    mkl = mkl_shim.MklSpmm()
    h = mkl.make(a)
    fn = {"syrk": lambda: mkl.syrk(h)}[op]
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    mkl.destroy(h)
    ts.sort()
    print(json.dumps({"ms": ts[len(ts) // 2] * 1e3, "cores": mkl.threads(), "version": mkl.version()}), flush=True)


if __name__ == "__main__":
    main()
