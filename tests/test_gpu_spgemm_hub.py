"""GPU: the hub path of the sparse x sparse product (csrc/spgemm_hub.inc, round 6) -- B's columns relabelled by popularity,
the leading column blocks of every hub row accumulated in dense LDS accumulators (k_hub_sym / k_hub_num), the rest of the row
through the range-partitioned path with its extents cut at the row's floor -- forced onto test-sized power-law operands and
compared with scipy / the CPU oracle: structure bit-exact after ordering, values at the north_star's bars.
Replaces one mkl_sparse_spmm (reference sparse_dot_mkl/_sparse_sparse.py:21-44)."""
import numpy as np
import pytest
import scipy.sparse as sps

from test_gpu_parity import _check_spgemm, rel_err, tol

pytestmark = pytest.mark.gpu

DEFAULTS = dict(spgemm_hub=0, spgemm_hub_fill_pct=20, spgemm_hub_acc_kb=64, spgemm_hub_block_kb=1 << 20, spgemm_slice_table=1)


class hub_forced:
    def __init__(self, gpu, **opts):
        self.gpu, self.opts = gpu, dict(spgemm_hub=2, **opts)

    def __enter__(self):
        for k, v in self.opts.items():
            self.gpu.mi_set_option(k, v)
        self.before = self.gpu.mi_get_counter("spgemm_hub_items")
        return self

    def items(self):
        return self.gpu.mi_get_counter("spgemm_hub_items") - self.before

    def __exit__(self, *a):
        for k in self.opts:
            self.gpu.mi_set_option(k, DEFAULTS[k])


def power_law(n, m, seed, dtype=np.float64, exponent=0.8, cols_skew=1.0, signed=False):
    """m x n: row i has ~ n / (i + 1)^exponent entries (the first rows are hubs); column j is drawn with probability
    ~ 1 / (j' + 1)^cols_skew under a random relabelling j -> j' (popular columns are NOT the first ones)."""
    r = np.random.default_rng(seed)
    deg = np.minimum((n / (np.arange(m) + 1.0) ** exponent).astype(np.int64) + 2, n // 2)
    r.shuffle(deg)
    p = 1.0 / (np.arange(n) + 1.0) ** cols_skew
    p /= p.sum()
    where = r.permutation(n)
    ptr = np.concatenate([[0], np.cumsum(deg)])
    ind = np.concatenate([np.sort(where[r.choice(n, d, replace=False, p=p)]) for d in deg]).astype(np.int32)
    val = r.standard_normal(ind.size) if signed else r.uniform(0.5, 1.5, ind.size)
    if np.dtype(dtype).kind == "c":
        val = val + 1j * r.uniform(0.5, 1.5, ind.size)
    return sps.csr_matrix((val.astype(dtype), ind, ptr), shape=(m, n))


def wide(dtype):
    return np.complex128 if np.dtype(dtype).kind == "c" else np.float64


def reference(a, b):
    w = wide(a.dtype)
    want = (a.astype(w) @ b.astype(w)).tocsr()
    want.sort_indices()
    return want


@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.complex64, np.complex128])
def test_hub_path_matches_scipy(gpu, dtype):
    """Power-law rows and columns: hub rows get dense blocks, every other row and the tails of the hub rows the range /
    hash paths, all on the relabelled copy of B; the result comes back in B's own column ids."""
    n = 1 << 13
    a, b = power_law(n, 3000, 1, dtype), power_law(n, n, 2, dtype)
    want = reference(a, b)
    assert np.diff(want.indptr).max() > 5000
    with hub_forced(gpu) as h:
        got = gpu.dot_product_mkl(a, b)
        assert h.items() > 0
    _check_spgemm(got, want, dtype)
    with hub_forced(gpu):
        got = gpu.dot_product_mkl(a, b, reorder_output=True)  # mkl_sparse_order on a result that is [ranges][dense blocks] per row
    assert np.array_equal(got.indptr, want.indptr) and np.array_equal(got.indices, want.indices)
    assert rel_err(got.data, want.data) <= tol(dtype)


@pytest.mark.parametrize("opts", [dict(spgemm_hub_acc_kb=2), dict(spgemm_hub_acc_kb=4, spgemm_hub_block_kb=64), dict(spgemm_hub_acc_kb=16), dict(spgemm_hub_acc_kb=32),
                                  dict(spgemm_hub_fill_pct=1), dict(spgemm_hub_fill_pct=60), dict(spgemm_hub_block_kb=16),
                                  dict(spgemm_slice_table=0)])
def test_hub_path_block_shapes(gpu, opts):
    """Narrow accumulators (many blocks per row), blocks bounded by entries instead of width, low / high fill thresholds
    (almost everything dense / almost nothing), the range path without its slice table: same product."""
    n = 1 << 13
    a, b = power_law(n, 2500, 3), power_law(n, n, 4, cols_skew=1.3)
    want = reference(a, b)
    with hub_forced(gpu, **opts) as h:
        got = gpu.dot_product_mkl(a, b)
        assert h.items() > 0
    _check_spgemm(got, want, np.float64)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_hub_path_keeps_cancelled_entries(gpu, oracle, dtype):
    """Signed, integer-valued operands: sums cancel EXACTLY to 0.0 all over the dense blocks; MKL keeps such entries
     -- the accumulators' own 'touched' marker (-0.0) must not lose them, nor entries whose products are
    all zero (explicit zeros stored in the operands)."""
    n = 1 << 12
    a, b = power_law(n, 1500, 5, dtype, signed=True), power_law(n, n, 6, dtype, signed=True)
    a.data[:] = np.sign(a.data) * (1 + (np.abs(a.data) > 1))   # +-1, +-2
    b.data[:] = np.sign(b.data)
    b.data[::7] = 0.0                                           # explicit zeros: products +0.0 / -0.0
    a.data[::11] = -0.0
    want = oracle.spgemm(a.astype(np.float64), b.astype(np.float64))  # sorted rows, zeros kept
    assert (want.data == 0).sum() > 1000
    with hub_forced(gpu) as h:
        got = gpu.dot_product_mkl(a, b)
        assert h.items() > 0
    got.sort_indices()
    assert np.array_equal(got.indptr, want.indptr) and np.array_equal(got.indices, want.indices)
    assert np.array_equal(got.data.astype(np.float64), want.data)  # small integers: exact in fp32 too


def test_hub_path_unsorted_b_duplicates_and_wide_indices(gpu):
    """B with shuffled rows and repeated column entries (summed), int64 index arrays; B's own arrays are not touched."""
    n = 1 << 12
    a, b0 = power_law(n, 1200, 7), power_law(n, n, 8)
    rng = np.random.default_rng(9)
    ind = np.repeat(b0.indices, 2)
    dat = np.repeat(b0.data, 2) * np.tile([0.25, 0.75], b0.nnz)
    ptr = b0.indptr.astype(np.int64) * 2
    for i in range(n):
        lo, hi = ptr[i], ptr[i + 1]
        o = rng.permutation(hi - lo)
        ind[lo:hi], dat[lo:hi] = ind[lo:hi][o], dat[lo:hi][o]
    b = sps.csr_matrix((dat, ind.astype(np.int64), ptr), shape=b0.shape)
    keep = b.indices.copy(), b.data.copy()
    want = reference(a, b0)
    with hub_forced(gpu) as h:
        got = gpu.dot_product_mkl(a, b)
        assert h.items() > 0
    _check_spgemm(got, want, np.float64)
    assert np.array_equal(b.indices, keep[0]) and np.array_equal(b.data, keep[1])


def test_hub_path_declines_politely(gpu):
    """No hub rows / uniform columns / the upper-triangle product / option off: the range path as before, same result."""
    n = 1 << 12
    a, b = power_law(n, 800, 10), power_law(n, n, 11)
    want = reference(a, b)
    before = gpu.mi_get_counter("spgemm_hub_items")
    _check_spgemm(gpu.dot_product_mkl(a, b), want, np.float64)  # library default: option off
    assert gpu.mi_get_counter("spgemm_hub_items") == before
    gpu.mi_set_option("spgemm_hub", 3)  # the relabelling alone: the range path on the relabelled copy, columns mapped back
    try:
        _check_spgemm(gpu.dot_product_mkl(a, b), want, np.float64)
    finally:
        gpu.mi_set_option("spgemm_hub", 0)
    # sparse gram (upper triangle): never relabelled
    x = power_law(n, 2000, 12)
    with hub_forced(gpu) as h:
        g = gpu.gram_matrix_mkl(x, reorder_output=True)
        assert h.items() == 0
    ref = sps.triu(x.T @ x).tocsr()
    ref.sort_indices()
    assert np.array_equal(g.indices, ref.indices) and rel_err(g.data, ref.data) <= 1e-12


def test_hub_path_bsr_operands_and_deterministic_option(gpu):
    """BSR operands reach the product as their element CSR (the hub path included); option deterministic keeps the range path
    (it re-forms the values in a fixed order on the ordered pattern) -- same structure either way."""
    n = 1 << 12
    a, b = power_law(n, 1024, 13), power_law(n, n, 14)
    want = reference(a, b)
    ab, bb = sps.bsr_matrix(a, blocksize=(2, 2)), sps.bsr_matrix(b, blocksize=(2, 2))
    with hub_forced(gpu) as h:
        got = gpu.dot_product_mkl(ab, bb)  # comes back re-blocked (the class of A): compared as a dense array
        assert h.items() > 0
    assert sps.isspmatrix_bsr(got) and rel_err(got.toarray(), np.maximum(want.toarray(), 0)) <= 1e-12
    gpu.mi_set_option("deterministic", 1)
    try:
        with hub_forced(gpu) as h:
            r1 = gpu.dot_product_mkl(a, b, reorder_output=True)
            r2 = gpu.dot_product_mkl(a, b, reorder_output=True)
            assert h.items() == 0
    finally:
        gpu.mi_set_option("deterministic", 0)
    assert np.array_equal(r1.indices, want.indices) and np.array_equal(r1.data, r2.data) and rel_err(r1.data, want.data) <= 1e-12


def test_copy_probe_reports_a_plausible_rate(gpu):
    """mi_sparse_probe_copy: the 'measured HBM roofline' of bench.py -- between 1 and 8 TB/s on an MI355X."""
    gbs = gpu.mi_probe_copy_gbs(1 << 28, 1)
    assert 1000.0 < gbs < 8000.0
