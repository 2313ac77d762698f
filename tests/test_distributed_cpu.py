"""CPU, world_size 2, gloo: the row-partition + broadcast(B) + all-gatherv(C) logic of
sparse_dot_amd.distributed.  The local product is INJECTED (the CPU oracle) because this box has no
GPU; on GPUs the default local product is the HIP kernel and the backend is nccl (= RCCL)."""
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


def _worker(rank, world, port, q):
    import sys
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    sys.path.insert(0, os.path.abspath(root))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from sparse_dot_amd import distributed as D
    from oracle import cpu_oracle
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(5)
        a = sps.random(301, 127, density=0.05, format="csr", dtype=np.float64, random_state=11)
        # skew: one very long row and a run of empty rows
        a = sps.vstack([a, sps.csr_matrix(np.ones((1, 127))), sps.csr_matrix((40, 127))]).tocsr()
        b = rng.random((127, 9))
        bounds = D.partition_rows(a.indptr, world)
        blocks = [int(bounds[i + 1] - bounds[i]) for i in range(world)]
        mine = D.row_block(a, int(bounds[rank]), int(bounds[rank + 1]))
        b_in = b if rank == 0 else np.zeros_like(b)  # only the root has B
        full = D.sharded_dot_product(mine, b_in, blocks, src=0, local_spmm=lambda x, y: cpu_oracle.spmm(x, y))
        part = D.sharded_dot_product(mine, b_in, blocks, src=0, gather=False,
                                     local_spmm=lambda x, y: cpu_oracle.spmm(x, y))
        want = a @ b
        ok = (np.allclose(full, want, rtol=1e-12, atol=1e-12)
              and np.allclose(part, want[bounds[rank]:bounds[rank + 1]], rtol=1e-12, atol=1e-12))
        q.put((rank, bool(ok), blocks))
    finally:
        dist.destroy_process_group()


def test_row_sharded_spmm_world2_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert sum(res[0][2]) == 342


def test_partition_rows_balances_work():
    from sparse_dot_amd.distributed import partition_rows, row_block
    rng = np.random.default_rng(1)
    lens = rng.integers(0, 50, 1000)
    lens[17] = 20000  # hub row
    indptr = np.concatenate([[0], np.cumsum(lens)])
    for parts in (1, 2, 3, 8):
        b = partition_rows(indptr, parts)
        assert b[0] == 0 and b[-1] == 1000 and np.all(np.diff(b) >= 0) and len(b) == parts + 1
        work = [(indptr[b[i + 1]] - indptr[b[i]]) + (b[i + 1] - b[i]) for i in range(parts)]
        # no block exceeds the ideal share by more than the largest single row
        assert max(work) <= (indptr[-1] + 1000) / parts + 20001
    assert partition_rows(np.array([0]), 4).tolist() == [0, 0, 0, 0, 0]
    a = sps.random(50, 20, density=0.3, format="csr", random_state=0)
    blk = row_block(a, 10, 35)
    assert np.array_equal(blk.toarray(), a.toarray()[10:35])
