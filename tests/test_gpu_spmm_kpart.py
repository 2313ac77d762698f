"""GPU: the column-partitioned SpMM (SpmmKpart, csrc/spmm.hip: long rows split by part(column) and gathered by one set of
XCDs per partition, partial rows summed in a fixed order) and the streaming walk of k_spmm against the CPU oracle --
forced on for test-sized matrices.  Replaces one mkl_sparse_?_mm (reference sparse_dot_mkl/_sparse_dense.py:111-123)."""
import numpy as np
import pytest
import scipy.sparse as sps

from test_gpu_parity import F32_TOL, dense, rel_err, tol

pytestmark = pytest.mark.gpu


def skewed_csr(m, k, dtype, seed, hubs=((7, 9000), (8, 1), (300, 2500), (301, 130), (1999, 4000)), max_len=70):
    """Row lengths 0..max_len with runs of empty rows and a few hub rows; power-law columns (so partitions are uneven
    unless the hash spreads them)."""
    rng = np.random.default_rng(seed)
    lens = rng.integers(0, max_len, m)
    lens[:40] = 0
    lens[-30:] = 0
    lens[1000:1100] = 0
    for r, l in hubs:
        if r < m:
            lens[r] = min(l, k)
    p = 1.0 / np.arange(1, k + 1) ** 0.9
    p /= p.sum()
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    indices = np.concatenate([np.sort(rng.choice(k, l, replace=False, p=p)) for l in lens]).astype(np.int32)
    data = rng.uniform(0.5, 1.5, indices.size)
    if np.dtype(dtype).kind == "c":
        data = data + 1j * rng.uniform(0.5, 1.5, indices.size)
    return sps.csr_matrix((data.astype(dtype), indices, indptr), shape=(m, k))


def wide(dtype):
    return np.complex128 if np.dtype(dtype).kind == "c" else np.float64


class kpart_forced:
    def __init__(self, gpu, min_row=16, parts=8, **extra):
        self.gpu, self.opts = gpu, dict(spmm_kpart=2, spmm_kpart_min_row=min_row, spmm_kpart_parts=parts, **extra)
        self.defaults = dict(spmm_kpart=1, spmm_kpart_min_row=64, spmm_kpart_parts=8, spmm_chunk=256, spmm_kpart_chunk=128,
                             spmm_slices=0, deterministic=0)

    def __enter__(self):
        for k, v in self.opts.items():
            self.gpu.mi_set_option(k, v)

    def __exit__(self, *a):
        for k in self.opts:
            self.gpu.mi_set_option(k, self.defaults[k])


@pytest.mark.parametrize("parts", [8, 4, 2])
@pytest.mark.parametrize("dtype,n", [(np.float32, 128), (np.float32, 64), (np.float32, 256), (np.float64, 64), (np.float64, 128),
                                     (np.complex64, 64), (np.complex128, 32)])
def test_kpart_matches_oracle(gpu, oracle, dtype, n, parts):
    """Long rows by column partition + short rows row-owned == the oracle; empty rows zero; alpha / beta."""
    a = skewed_csr(2500, 12000, dtype, 3)
    b = dense((12000, n), dtype, 4)
    want = oracle.spmm(a.astype(wide(dtype)), b.astype(wide(dtype)))
    with kpart_forced(gpu, 16, parts):
        got = gpu.dot_product_mkl(a, b)
        assert gpu.mi_get_counter("spmm_last_kpart") == float(parts)
        assert 0.3 < gpu.mi_get_counter("spmm_kpart_long_share") < 1.0
        out = np.full_like(got, 2.0)
        got2 = gpu.dot_product_mkl(a, b, out=out, out_scalar=-1.5)
    assert got.dtype == dtype and rel_err(got, want) <= tol(dtype)
    assert not got[9:40].any() and not got[-30:].any() and not got[1000:1100].any()
    assert got2 is out
    np.testing.assert_allclose(got2, want - 3.0, rtol=10 * tol(dtype), atol=10 * tol(dtype))


@pytest.mark.parametrize("chunk", [128, 256, 1024])
def test_kpart_chunks(gpu, oracle, chunk):
    """Partition boundaries against chunk boundaries (a chunk straddling two partitions), hub sub-rows cut into carries."""
    a = skewed_csr(2200, 9000, np.float32, 5, hubs=((0, 9000), (1, 8000), (2, 17), (2100, 6000)))
    b = dense((9000, 128), np.float32, 6)
    want = oracle.spmm(a.astype(np.float64), b.astype(np.float64))
    with kpart_forced(gpu, 24, 8, spmm_chunk=chunk, spmm_kpart_chunk=chunk):  # (the partitioned kernels have their own chunk option)
        got = gpu.dot_product_mkl(a, b)
        assert gpu.mi_get_counter("spmm_last_kpart") == 8.0
        again = gpu.dot_product_mkl(a, b)
    assert rel_err(got, want) <= F32_TOL
    assert np.array_equal(got, again)  # fixed summation order: the same bits on every call


def test_kpart_adopted_on_third_product_and_dropped_by_set_values(gpu, oracle):
    """Library defaults on a resident handle: two row-owned products, then the partitioned plan; mi_sparse_?_set_values
    drops the plan's copy of the values; option deterministic keeps the row-owned kernel."""
    a = skewed_csr(3000, 20000, np.float32, 9, max_len=40)
    b = dense((20000, 128), np.float32, 10)
    want = oracle.spmm(a.astype(np.float64), b.astype(np.float64))
    gpu.mi_set_option("spmm_kpart_min_row", 16)
    try:
        A = gpu.to_device(a)
        seen = []
        for _ in range(4):
            got = gpu.dot_product_mkl(A, b)
            seen.append(gpu.mi_get_counter("spmm_last_kpart"))
            assert rel_err(got, want) <= F32_TOL
        assert seen == [0.0, 0.0, 0.0, 0.0]  # test-sized: below the size gate of option value 1
        gpu.mi_set_option("spmm_kpart", 2)
        assert rel_err(gpu.dot_product_mkl(A, b), want) <= F32_TOL
        assert gpu.mi_get_counter("spmm_last_kpart") == 8.0
        # new values through the API: the plan is rebuilt from them
        a2 = a.copy()
        a2.data = (a.data * np.float32(1.25)).astype(np.float32)
        from sparse_dot_amd._mi_interface import MI, _check_return_value
        _check_return_value(MI.call("mi_sparse_s_set_values", A.handle.ptr, a2.data.ctypes.data), "set_values")
        got = gpu.dot_product_mkl(A, b)
        assert rel_err(got, 1.25 * want) <= F32_TOL
        assert gpu.mi_get_counter("spmm_last_kpart") == 8.0
        # another chunk size on a handle that already holds the partitioned plan: its chunk ranges are rebuilt
        for chunk in (128, 1024, 256):
            gpu.mi_set_option("spmm_chunk", chunk)
            gpu.mi_set_option("spmm_kpart_chunk", chunk)
            got = gpu.dot_product_mkl(A, b)
            assert gpu.mi_get_counter("spmm_last_kpart") == 8.0 and rel_err(got, 1.25 * want) <= F32_TOL, chunk
        gpu.mi_set_option("deterministic", 1)
        got = gpu.dot_product_mkl(A, b)
        assert gpu.mi_get_counter("spmm_last_kpart") == 0.0 and rel_err(got, 1.25 * want) <= F32_TOL
        A.free()
    finally:
        gpu.mi_set_option("deterministic", 0)
        gpu.mi_set_option("spmm_chunk", 256)
        gpu.mi_set_option("spmm_kpart_chunk", 128)
        gpu.mi_set_option("spmm_kpart", 1)
        gpu.mi_set_option("spmm_kpart_min_row", 64)


def test_kpart_transposed_and_column_major(gpu, oracle):
    """op(A) = A^T goes through the partitioned plan of the transposed orientation; column-major operands are re-laid and
    take the same path."""
    a = skewed_csr(2500, 6000, np.float64, 11)
    at = a.T.tocsr()
    b = dense((2500, 64), np.float64, 12)
    want = oracle.spmm(at.astype(np.float64), b)
    with kpart_forced(gpu, 16, 8):
        got = gpu.dot_product_mkl(np.ascontiguousarray(b.T), a)  # dense x sparse = (A^T b)^T: op = transpose on the handle of A
        assert gpu.mi_get_counter("spmm_last_kpart") == 8.0
        assert got.shape == (64, 6000) and rel_err(got.T, want) <= 1e-12
        bf = np.asfortranarray(b)
        got = gpu.dot_product_mkl(at, bf)
        assert got.flags["F_CONTIGUOUS"] and rel_err(got, want) <= 1e-12


def test_kpart_all_rows_long_and_no_rows_long(gpu, oracle):
    a = skewed_csr(600, 3000, np.float32, 13, hubs=(), max_len=60)
    a = a[np.diff(a.indptr) >= 20]  # every row long at min_row 16
    b = dense((3000, 128), np.float32, 14)
    want = oracle.spmm(a.astype(np.float64), b.astype(np.float64))
    with kpart_forced(gpu, 16, 8):
        got = gpu.dot_product_mkl(a, b)
        assert gpu.mi_get_counter("spmm_last_kpart") == 8.0 and gpu.mi_get_counter("spmm_kpart_long_share") == 1.0
        assert rel_err(got, want) <= F32_TOL
    with kpart_forced(gpu, 100000, 8):  # nothing reaches the threshold: declined, row-owned
        got = gpu.dot_product_mkl(a, b)
        assert gpu.mi_get_counter("spmm_last_kpart") == 0.0
        assert rel_err(got, want) <= F32_TOL


@pytest.mark.parametrize("dtype,n", [(np.float32, 128), (np.float32, 96), (np.float32, 20), (np.float64, 64), (np.float64, 7),
                                     (np.complex64, 32), (np.complex128, 16)])
def test_streaming_walk_every_lane_group_width(gpu, oracle, dtype, n):
    """k_spmm's walk (batches of loads across row ends, wave-uniform row state): every lane-group width, scalar path included;
    rows of exactly the cut length, hub rows, empty rows; alpha / beta."""
    a = skewed_csr(1800, 5000, dtype, 21, hubs=((5, 3000), (6, 129), (7, 128), (900, 257)))
    b = dense((5000, n), dtype, 22)
    want = oracle.spmm(a.astype(wide(dtype)), b.astype(wide(dtype)))
    got = gpu.dot_product_mkl(a, b)
    out = np.full_like(got, 1.0)
    got2 = gpu.dot_product_mkl(a, b, out=out, out_scalar=0.5)
    assert rel_err(got, want) <= tol(dtype)
    np.testing.assert_allclose(got2, want + 0.5, rtol=10 * tol(dtype), atol=10 * tol(dtype))
    assert not got[8:40].any()


@pytest.mark.parametrize("seed", list(range(24)))
def test_spmm_randomised_shapes_row_owned_and_partitioned(gpu, oracle, seed):
    """Seeded random sweep over what the SpMM kernels branch on: rows / columns / widths (vector and scalar paths, every
    lane-group width), row-length mixes (empty runs, rows of exactly the cut length 128 / 129, hub rows over many chunks),
    chunk size, alpha / beta, dtype -- the row-owned and the column-partitioned form must both
    equal the oracle."""
    rng = np.random.default_rng(1000 + seed)
    dtype = [np.float32, np.float64, np.complex64, np.complex128][seed % 4]
    m = int(rng.integers(1, 3000))
    k = int(rng.integers(1, 5000))
    n = int(rng.choice([1, 2, 3, 4, 5, 8, 12, 16, 24, 32, 48, 64, 100, 128, 192, 256, 320, 512]))
    lens = rng.integers(0, min(k, 40) + 1, m)
    for _ in range(int(rng.integers(0, 6))):
        lens[int(rng.integers(0, m))] = min(k, int(rng.choice([127, 128, 129, 255, 256, 257, 1000, 4000])))
    if m > 50:
        s = int(rng.integers(0, m - 40))
        lens[s:s + 40] = 0
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    indices = np.concatenate([np.sort(rng.choice(k, l, replace=False)) for l in lens] + [np.empty(0, np.int64)]).astype(np.int32)
    data = rng.uniform(0.5, 1.5, indices.size)
    if np.dtype(dtype).kind == "c":
        data = data + 1j * rng.uniform(0.5, 1.5, indices.size)
    a = sps.csr_matrix((data.astype(dtype), indices, indptr), shape=(m, k))
    b = dense((k, n), dtype, 2000 + seed)
    want = oracle.spmm(a.astype(wide(dtype)), b.astype(wide(dtype)))
    chunk = int(rng.choice([128, 256, 512, 1024]))
    beta = float(rng.choice([0.0, 0.5, -1.25]))
    c0 = dense((m, n), dtype, 3000 + seed)
    for opts in (dict(spmm_kpart=0), dict(spmm_kpart=2, spmm_kpart_min_row=int(rng.choice([2, 8, 32, 128])),
                                                            spmm_kpart_parts=int(rng.choice([8, 4, 2])))):
        gpu.mi_set_option("spmm_chunk", chunk)
        gpu.mi_set_option("spmm_kpart_chunk", chunk)
        for name, value in opts.items():
            gpu.mi_set_option(name, value)
        try:
            got = gpu.dot_product_mkl(a, b)
            out = c0.copy()
            got2 = gpu.dot_product_mkl(a, b, out=out, out_scalar=beta) if beta else None
        finally:
            for name, value in dict(spmm_chunk=256, spmm_kpart_chunk=128, spmm_kpart=1, spmm_kpart_min_row=64, spmm_kpart_parts=8).items():
                gpu.mi_set_option(name, value)
        if n == 1:
            got = got.reshape(m, 1)
        assert got.shape == (m, n) and rel_err(got, want) <= tol(dtype), (opts, rel_err(got, want))
        if got2 is not None:
            np.testing.assert_allclose(got2, want + beta * c0.astype(wide(dtype)), rtol=20 * tol(dtype), atol=20 * tol(dtype))


def test_optimize_builds_the_partitioned_plan_up_front(gpu, oracle):
    """mi_sparse_optimize / to_device(a, optimize=True) -- the mkl_sparse_optimize analogue: the FIRST product of the handle
    already runs the column-partitioned kernels (without it: the third); same result as the row-owned product to tolerance,
    same bits on every later call; harmless on a matrix without long rows."""
    a = skewed_csr(3000, 20000, np.float32, 19, max_len=40)
    b = dense((20000, 128), np.float32, 20)
    want = oracle.spmm(a.astype(np.float64), b.astype(np.float64))
    with kpart_forced(gpu, 16, 8):
        A = gpu.to_device(a, optimize=True)
        got = gpu.dot_product_mkl(A, b)
        assert gpu.mi_get_counter("spmm_last_kpart") == 8.0
        again = gpu.dot_product_mkl(A, b)
        A.free()
    assert rel_err(got, want) <= F32_TOL and np.array_equal(got, again)
    u = sps.random(500, 400, density=0.01, format="csr", dtype=np.float64, random_state=3)
    U = gpu.to_device(u).optimize()
    x = dense((400, 8), np.float64, 4)
    assert rel_err(gpu.dot_product_mkl(U, x), oracle.spmm(u, x)) <= 1e-12
    U.free()
