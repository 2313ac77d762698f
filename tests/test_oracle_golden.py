"""CPU: pin the oracle (oracle/sparse_oracle.c via oracle/cpu_oracle.py) to the golden vectors
captured from the unmodified reference running on MKL (tests/golden/golden_v1.npz, generator
oracle/make_golden.py), and to scipy / numpy on the same inputs."""
import numpy as np
import pytest
import scipy.sparse as sps

import golden_util as G

CASES = G.cases(raises=False)


def _call(oracle, c):
    a, b, out = G.operand(c["a"]), G.operand(c["b"]), G.out_array(c)
    kw = dict(c["kwargs"])
    if c["fn"] == "dot":
        return oracle.dot_product(a, b, out=out, **kw), a, b
    return oracle.gram_matrix(a, out=out, **kw), a, b


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle_matches_reference_golden(oracle, case):
    if case.get("reference_deviates"):
        pytest.skip("the reference's own answer is wrong for this input (see make_golden.py)")
    got, _, _ = _call(oracle, case)
    exp = G.operand(case["result"])
    rtol, atol = G.tolerances(exp.dtype)
    if sps.issparse(exp):
        assert sps.issparse(got)
        g = got.asformat(exp.format) if exp.format != "bsr" else got.tobsr(blocksize=exp.blocksize)
        if exp.format == "bsr":
            np.testing.assert_allclose(g.toarray(), exp.toarray(), rtol=rtol, atol=atol)
            return
        g.sort_indices()
        # bit-exact structure (after ordering), values within the north_star tolerance
        assert np.array_equal(g.indptr, exp.indptr)
        assert np.array_equal(g.indices, exp.indices)
        np.testing.assert_allclose(g.data, exp.data, rtol=rtol, atol=atol)
    else:
        assert got.shape == exp.shape and got.dtype == exp.dtype
        if case["fn"] == "gram":  # only the upper triangle is defined
            iu = np.triu_indices(exp.shape[0])
            np.testing.assert_allclose(got[iu], exp[iu], rtol=rtol, atol=atol)
        else:
            np.testing.assert_allclose(got, exp, rtol=rtol, atol=atol)


@pytest.mark.parametrize("case", [c for c in CASES if c["scipy"] is not None],
                         ids=[c["name"] for c in CASES if c["scipy"] is not None])
def test_oracle_structure_matches_scipy(oracle, case):
    """Sparse results: indptr / indices equal scipy's (after sort_indices) unless an entry cancels
    to exactly zero, which scipy prunes and MKL (and the oracle, and the build) keep."""
    got, _, _ = _call(oracle, case)
    sc = G.operand(case["scipy"])
    if case["name"].endswith("cancellation") or case["fn"] == "gram":
        # scipy's triu / pruning changes structure only by dropping explicit zeros
        np.testing.assert_allclose(got.toarray() if case["fn"] == "dot" else np.triu(got.toarray()),
                                   sc.toarray() if sps.issparse(sc) else sc, rtol=1e-5, atol=1e-6)
        return
    g = got.tocsr() if got.format != "bsr" else got.tocsr()
    s = sc.tocsr()
    g.sort_indices()
    s.sort_indices()
    if got.format == "bsr":
        np.testing.assert_allclose(g.toarray(), s.toarray(), rtol=1e-5, atol=1e-6)
        return
    assert np.array_equal(g.indptr, s.indptr) and np.array_equal(g.indices, s.indices)


def test_oracle_transpose_order_roundtrip(oracle):
    rng = np.random.default_rng(3)
    a = sps.random(37, 53, density=0.2, format="csr", dtype=np.float64, random_state=4)
    perm_a = a.copy()
    # shuffle entries inside rows
    for i in range(a.shape[0]):
        lo, hi = a.indptr[i], a.indptr[i + 1]
        p = rng.permutation(hi - lo)
        perm_a.indices[lo:hi] = a.indices[lo:hi][p]
        perm_a.data[lo:hi] = a.data[lo:hi][p]
    o = oracle.order(perm_a)
    assert np.array_equal(o.indices, a.indices) and np.array_equal(o.data, a.data)
    t = oracle.transpose(a)
    assert np.array_equal(t.toarray(), a.toarray().T)
    tt = a.T.tocsr()
    tt.sort_indices()
    assert np.array_equal(t.indices, tt.indices) and np.array_equal(t.indptr, tt.indptr)


def test_oracle_spmm_alpha_beta_layouts(oracle):
    rng = np.random.default_rng(0)
    a = sps.random(40, 30, density=0.2, format="csr", dtype=np.float64, random_state=1)
    for order in ("C", "F"):
        b = np.asarray(rng.random((30, 7)), order=order)
        c = np.asarray(rng.random((40, 7)), order=order)
        got = oracle.spmm(a, b, alpha=2.5, beta=-0.5, c=c)
        np.testing.assert_allclose(got, 2.5 * (a @ b) - 0.5 * c, rtol=1e-13)
        bt = np.asarray(rng.random((40, 5)), order=order)
        got = oracle.spmm(a, bt, op=oracle.OP_T)
        np.testing.assert_allclose(got, a.T @ bt, rtol=1e-13)
