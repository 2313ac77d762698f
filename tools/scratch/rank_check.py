import os, sys, time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, scipy.sparse as sps
import sparse_dot_amd as sda
print(sda.mi_get_version_string())
rng = np.random.default_rng(0); OPTS = [int(x) for x in os.environ.get("OPTS","1,0").split(",")]; pass
def hubby(n, m, per, hubs, hub_len, dtype):
    a = sps.random(n, m, density=per / m, format="csr", random_state=rng, dtype=np.float64)
    # hub rows
    rows = rng.choice(n, hubs, replace=False)
    l = a.tolil()
    for r in rows:
        cols = rng.choice(m, hub_len, replace=False)
        l[r, cols] = rng.random(hub_len) + 0.5
    a = l.tocsr().astype(dtype); a.sort_indices()
    return a
for dtype, tol in ((np.float64, 1e-12), (np.float32, 1e-5), (np.complex128, 1e-12)):
    n = 3000
    a = hubby(n, n, 6, 12, 700, np.float64)
    b = hubby(n, n, 6, 12, 900, np.float64)
    if dtype == np.complex128:
        a = (a + 1j * a).astype(dtype); b = (b - 0.5j * b).astype(dtype)
    else:
        a = a.astype(dtype); b = b.astype(dtype)
    for opt in OPTS:
        sda.mi_set_option("spgemm_rank", opt) if hasattr(sda, "mi_set_option") else None
        t = time.time()
        c = sda.dot_product_mkl(a, b)
        dt = time.time() - t
        ref = (a.astype(np.complex128 if dtype == np.complex128 else np.float64) @ b.astype(np.complex128 if dtype == np.complex128 else np.float64)).tocsr()
        ref.sort_indices(); c = c.tocsr(); c.sort_indices()
        ok_struct = np.array_equal(c.indptr, ref.indptr) and np.array_equal(c.indices, ref.indices)
        err = np.max(np.abs(c.data - ref.data) / np.maximum(np.abs(ref.data), 1e-300)) if ok_struct else None
        print(dtype.__name__, "rank" if opt else "hash", "struct", ok_struct, "err", err, "nnz", c.nnz, "max row", np.diff(c.indptr).max(), "%.1fs" % dt)
# larger: rows well beyond one item
for dtype in (np.float64, np.float32):
    n = 60000
    a = hubby(n, n, 8, 40, 6000, np.float64).astype(dtype)
    b = hubby(n, n, 8, 40, 9000, np.float64).astype(dtype)
    for opt in OPTS:
        sda.mi_set_option("spgemm_rank", opt)
        t = time.time(); c = sda.dot_product_mkl(a, b); dt = time.time() - t
        ref = (a.astype(np.float64) @ b.astype(np.float64)).tocsr(); ref.sort_indices(); c = c.tocsr(); srt = c.has_sorted_indices; c.sort_indices()
        ok = np.array_equal(c.indptr, ref.indptr) and np.array_equal(c.indices, ref.indices)
        err = np.max(np.abs(c.data - ref.data) / np.abs(ref.data)) if ok else None
        print(dtype.__name__, "rank" if opt else "hash", "struct", ok, "err", err, "nnz", c.nnz, "max row", np.diff(c.indptr).max(), "%.2fs" % dt)
