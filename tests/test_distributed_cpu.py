"""CPU, world_size 2, gloo: the partition + collective logic of sparse_dot_amd.distributed and of bench.py's
partitioned step.  The local products are INJECTED (the CPU oracle / scipy) because this box has no GPU; on
GPUs the default local product is the HIP kernel and the backend is nccl (= RCCL)."""
import os
import socket

import numpy as np
import pytest
import scipy.sparse as sps


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup(rank, world, port):
    import sys
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    sys.path.insert(0, os.path.abspath(root))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    return dist


def _skewed_matrix():
    a = sps.random(301, 127, density=0.05, format="csr", dtype=np.float64, random_state=11)
    # skew: one very long row and a run of empty rows
    return sps.vstack([a, sps.csr_matrix(np.ones((1, 127))), sps.csr_matrix((40, 127))]).tocsr()


def _worker_spmm(rank, world, port, q):
    dist = _setup(rank, world, port)
    try:
        import torch
        from sparse_dot_amd import distributed as D
        from oracle import cpu_oracle
        rng = np.random.default_rng(5)
        a = _skewed_matrix()
        b = rng.random((127, 9))
        bounds = D.partition_rows(a.indptr, world)
        blocks = [int(bounds[i + 1] - bounds[i]) for i in range(world)]
        mine = D.row_block(a, int(bounds[rank]), int(bounds[rank + 1]))
        b_in = b if rank == 0 else np.zeros_like(b)  # only the root has B
        spmm = lambda x, y: cpu_oracle.spmm(x, y)  # noqa: E731
        want = a @ b
        ok = True
        for mode in ("bcast", "padded"):
            full = D.sharded_dot_product(mine, b_in, blocks, src=0, local_spmm=spmm, gather_mode=mode)
            ok = ok and np.allclose(full, want, rtol=1e-12, atol=1e-12)
        part = D.sharded_dot_product(mine, b_in, blocks, src=0, gather=False, local_spmm=spmm)
        ok = ok and np.allclose(part, want[bounds[rank]:bounds[rank + 1]], rtol=1e-12, atol=1e-12)
        # resident form: the same ShardedCSR serves several products, tensors in / tensors out
        with D.ShardedCSR(mine, bounds, local_spmm=spmm) as sh:
            for k in range(2):
                bk = torch.from_numpy(b * (k + 1) if rank == 0 else np.zeros_like(b))
                got = sh.dot(bk, src=0)
                ok = ok and isinstance(got, torch.Tensor) and np.allclose(got.numpy(), want * (k + 1), rtol=1e-12, atol=1e-12)
        q.put((rank, bool(ok), blocks))
    finally:
        dist.destroy_process_group()


def _worker_bench_partition(rank, world, port, q):
    """Drives bench.run_partitioned -- the code the multi-GPU bench times -- on CPU tensors."""
    dist = _setup(rank, world, port)
    try:
        import torch
        import bench
        a = _skewed_matrix().astype(np.float32)
        rng = np.random.default_rng(3)
        b = rng.random((127, 8)).astype(np.float32)
        n_rows = a.shape[0]
        indptr = torch.from_numpy(a.indptr.astype(np.int32))
        indices = torch.from_numpy(a.indices.astype(np.int32))
        vals = torch.from_numpy(a.data)
        B = torch.from_numpy(b.copy() if rank == 0 else np.zeros_like(b))
        made = []

        def make_local(bp, bi, bv, rows, cols):
            blk = sps.csr_matrix((bv.numpy(), bi.numpy(), bp.numpy()), shape=(rows, 127))
            made.append(blk)

            def mm(Bt, Ct):
                Ct.copy_(torch.from_numpy((blk @ Bt.numpy()).astype(np.float32)))
            return mm, (lambda: None)

        # bench passes the ROW count as `n` (square matrices there); C has n rows
        res = bench.run_partitioned(torch, dist, torch.device("cpu"), indptr, indices, vals, n_rows, B, steps=2, warmup=1,
                                    make_local=make_local, gather_mode="bcast")
        want = a @ b
        ok = np.allclose(res["C"].numpy(), want, rtol=1e-5, atol=1e-6)
        ok = ok and res["t_end_to_end"] > 0 and res["t_compute"] > 0
        ok = ok and all(k in res for k in ("t_bcast", "t_gather_bcast", "t_gather_padded"))
        ok = ok and int(res["bounds"][-1]) == n_rows and made[0].shape[0] == res["block_rows"]
        q.put((rank, bool(ok), res["block_rows"]))
    finally:
        dist.destroy_process_group()


def _worker_spgemm_gram(rank, world, port, q):
    dist = _setup(rank, world, port)
    try:
        from sparse_dot_amd import distributed as D
        a = _skewed_matrix()
        b = sps.random(127, 83, density=0.08, format="csr", dtype=np.float64, random_state=4)
        bounds = D.partition_rows(a.indptr, world)
        blocks = [int(bounds[i + 1] - bounds[i]) for i in range(world)]
        mine = D.row_block(a, int(bounds[rank]), int(bounds[rank + 1]))

        def local_spgemm(x, y):
            c = (x @ y).tocsr()
            c.sort_indices()
            return c
        got = D.sharded_sparse_dot_product(mine, b if rank == 0 else None, blocks, src=0, local_spgemm=local_spgemm)
        want = (a @ b).tocsr()
        want.sort_indices()
        ok = (got.shape == want.shape and np.array_equal(got.indptr, want.indptr) and np.array_equal(got.indices, want.indices)
              and np.allclose(got.data, want.data, rtol=1e-12, atol=0))
        # gram by output rows: every rank holds A, produces a band of equal triangle area
        x = sps.random(150, 61, density=0.1, format="csr", dtype=np.float64, random_state=8)
        ref = np.triu((x.T @ x).toarray())
        full = D.sharded_gram_matrix(x, local_gram_rows=lambda m, c0, c1: ref[c0:c1].copy())
        band, (c0, c1) = D.sharded_gram_matrix(x, gather=False, local_gram_rows=lambda m, c0, c1: ref[c0:c1].copy())
        ok = ok and np.allclose(full, ref) and np.allclose(band, ref[c0:c1])
        q.put((rank, bool(ok), (c0, c1)))
    finally:
        dist.destroy_process_group()


def _worker_p2p_and_pipeline(rank, world, port, q):
    """The xGMI-shaped collectives (grouped point-to-point all-gatherv, scatter + all-gather broadcast) and the
    panel-pipelined product, on world = 3 so that the "forward my slab to the other non-root ranks" leg exists."""
    dist = _setup(rank, world, port)
    try:
        import torch
        from sparse_dot_amd import distributed as D
        from oracle import cpu_oracle
        rng = np.random.default_rng(5)
        a = _skewed_matrix()
        b = rng.random((127, 12))
        want = a @ b
        bounds = D.partition_rows(a.indptr, world)
        mine = D.row_block(a, int(bounds[rank]), int(bounds[rank + 1]))
        spmm = lambda x, y: cpu_oracle.spmm(x, y)  # noqa: E731
        ok = True
        # broadcast_rows from a non-zero root, both forms (tall and shorter-than-world tensors)
        for src in (0, 2):
            for mode in ("scatter_allgather", "bcast"):
                for shape in ((127, 12), (2, 5), (0, 4)):
                    ref = torch.arange(shape[0] * shape[1], dtype=torch.float64).reshape(shape) + 7.0
                    t = ref.clone() if rank == src else torch.zeros(shape, dtype=torch.float64)
                    D.broadcast_rows(t, src=src, mode=mode).wait()
                    ok = ok and torch.equal(t, ref)
        # all-gatherv, all three forms agree
        for mode in ("p2p", "bcast", "padded"):
            full = torch.zeros((a.shape[0], 12), dtype=torch.float64)
            full[int(bounds[rank]):int(bounds[rank + 1])] = torch.from_numpy(want[bounds[rank]:bounds[rank + 1]])
            D.gather_rows(full, bounds, None, mode)
            ok = ok and np.array_equal(full.numpy(), want)
        with D.ShardedCSR(mine, bounds, local_spmm=spmm) as sh:
            bt = torch.from_numpy(b.copy() if rank == 1 else np.zeros_like(b))
            got = sh.dot(bt, src=1, gather_mode="p2p", bcast_mode="scatter_allgather")
            ok = ok and np.allclose(got.numpy(), want, rtol=1e-12, atol=1e-12)
            # panel-major operands: 3 panels of 4 columns, root 0, a second group for the gathers
            g2 = dist.new_group(list(range(world)))
            panels = np.ascontiguousarray(b.reshape(127, 3, 4).transpose(1, 0, 2))
            bp = torch.from_numpy(panels.copy() if rank == 0 else np.zeros_like(panels))
            for depth in (1, 2, 5):
                for gmode, gg in (("p2p", g2), ("p2p", None), ("bcast", None)):
                    if rank != 0:
                        bp.zero_()
                    cp = sh.dot_pipelined(bp, src=0, gather_mode=gmode, gather_group=gg, depth=depth)
                    got = cp.numpy().transpose(1, 0, 2).reshape(a.shape[0], 12)
                    ok = ok and np.allclose(got, want, rtol=1e-12, atol=1e-12)
                    ok = ok and np.array_equal(bp.numpy(), panels)  # every rank ends up holding B
        q.put((rank, bool(ok), int(bounds[rank + 1] - bounds[rank])))
    finally:
        dist.destroy_process_group()


def _run(worker, world=2):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    return sorted(res)


def test_row_sharded_spmm_world2_gloo():
    res = _run(_worker_spmm)
    assert sum(res[0][2]) == 342


def test_bench_partitioned_step_world2_gloo():
    res = _run(_worker_bench_partition)
    assert res[0][2] + res[1][2] == 342


def test_sharded_spgemm_and_gram_world2_gloo():
    res = _run(_worker_spgemm_gram)
    assert res[0][2][1] == res[1][2][0]  # the two output bands are adjacent


def test_p2p_collectives_and_pipelined_product_world3_gloo():
    res = _run(_worker_p2p_and_pipeline, world=3)
    assert sum(r[2] for r in res) == 342


def test_partition_rows_balances_work():
    from sparse_dot_amd.distributed import partition_rows, partition_triangle, row_block
    rng = np.random.default_rng(1)
    lens = rng.integers(0, 50, 1000)
    lens[17] = 20000  # hub row
    indptr = np.concatenate([[0], np.cumsum(lens)])
    for parts in (1, 2, 3, 8):
        b = partition_rows(indptr, parts, model="items")
        assert b[0] == 0 and b[-1] == 1000 and np.all(np.diff(b) >= 0) and len(b) == parts + 1
        work = [(indptr[b[i + 1]] - indptr[b[i]]) + (b[i + 1] - b[i]) for i in range(parts)]
        # no block exceeds the ideal share by more than the largest single row
        assert max(work) <= (indptr[-1] + 1000) / parts + 20001
        # the default: equal COST under the SpMM model (a row 3.3, a short-row entry 1.6, a long-row entry 1.0)
        b = partition_rows(indptr, parts)
        assert b[0] == 0 and b[-1] == 1000 and np.all(np.diff(b) >= 0) and len(b) == parts + 1
        row_cost = 3.3 + lens * np.where(lens >= 128, 1.0, 1.6)
        cost = [row_cost[b[i]:b[i + 1]].sum() for i in range(parts)]
        assert max(cost) <= row_cost.sum() / parts + row_cost.max() + 1e-6
    with pytest.raises(ValueError):
        partition_rows(indptr, 2, model="nope")
    assert partition_rows(np.array([0]), 4).tolist() == [0, 0, 0, 0, 0]
    a = sps.random(50, 20, density=0.3, format="csr", random_state=0)
    blk = row_block(a, 10, 35)
    assert np.array_equal(blk.toarray(), a.toarray()[10:35])
    for n, parts in ((1000, 4), (262144, 8), (7, 3), (0, 2)):
        t = partition_triangle(n, parts)
        assert t[0] == 0 and t[-1] == n and np.all(np.diff(t) >= 0) and len(t) == parts + 1
        if n >= 1000:
            area = [(t[i + 1] - t[i]) * n - (t[i + 1] * (t[i + 1] - 1) - t[i] * (t[i] - 1)) / 2 for i in range(parts)]
            assert max(area) <= 1.05 * sum(area) / parts
