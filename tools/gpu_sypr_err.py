"""Developer tool (round 5): measured relative error of the fp32 triple product mi_sparse_sypr / mi_sparse_s_syprd (two chained
fp32 products) against an fp64 evaluation, positive data, over shapes and densities (the test's tolerance is the north_star's 1e-5)."""
import sys, json
sys.path.insert(0, "/root/repo")
import numpy as np, scipy.sparse as sps
import sparse_dot_amd as sda
def pos(m, n, d, seed):
    a = sps.random(m, n, density=d, format="csr", dtype=np.float64, random_state=seed)
    a.data[:] = np.random.default_rng(seed + 7).uniform(0.5, 1.5, a.nnz)
    return a.astype(np.float32)
worst = 0.0
for (m, k, d, seed) in [(300, 200, 0.05, 3), (200, 300, 0.05, 5), (1000, 800, 0.02, 7), (2000, 1500, 0.01, 9), (500, 500, 0.2, 11), (3000, 3000, 0.01, 13)]:
    for tr in (False, True):
        x = pos(m, k, d, seed)
        kk = x.shape[0] if tr else x.shape[1]
        s0 = pos(kk, kk, d, seed + 1)
        sym = (s0 + s0.T).tocsr()
        bu = sps.triu(sym).tocsr()
        opx = (x.T if tr else x).astype(np.float64)
        ref = np.triu((opx @ sym.astype(np.float64) @ opx.T).toarray())
        got = sda.sparse_sypr(x, bu, transpose_a=tr).toarray()
        mask = ref != 0
        e1 = float(np.max(np.abs(got[mask] - ref[mask]) / ref[mask]))
        bd = np.triu(sym.toarray()).astype(np.float32)
        gd = np.triu(sda.sparse_sypr(x, bd, transpose_a=tr))
        e2 = float(np.max(np.abs(gd[mask] - ref[mask]) / ref[mask]))
        terms = float((opx != 0).sum(1).mean() * (sym != 0).sum(1).mean())
        worst = max(worst, e1, e2)
        print(json.dumps({"op(A)": [opx.shape[0], opx.shape[1]], "density": d, "transpose": tr, "products_per_entry_about": round(terms),
                          "max_rel_err_sparse": e1, "max_rel_err_dense": e2}), flush=True)
print(json.dumps({"worst": worst}))
