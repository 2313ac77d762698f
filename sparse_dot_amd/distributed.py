"""
Row-partitioned SpMM across the GPUs of one node (one process per GPU, torch.distributed).

The reference has no multi-device path; this is the north_star's extension of it: the left CSR
matrix is split into contiguous, nnz-balanced row blocks (output rows are independent:
C[i, :] depends only on A[i, :] and B), the dense B is broadcast from a root rank, every rank
multiplies its block with the single-GPU kernel, and the output row blocks are all-gathered
(variable block heights -> padded all_gather).  Backend "nccl" is RCCL over xGMI on ROCm; the same
code runs on "gloo" CPU tensors, which is how the host logic is tested without GPUs.
"""
import numpy as _np
from scipy import sparse as _sps


def partition_rows(indptr, nparts):
    """Contiguous row blocks balanced by (nnz + rows) work items, the unit the SpMM kernel balances
    on.  Returns nparts + 1 boundaries b with b[0] = 0, b[-1] = nrows, non-decreasing."""
    indptr = _np.asarray(indptr, dtype=_np.int64)
    nrows = indptr.shape[0] - 1
    if nparts < 1:
        raise ValueError("nparts must be positive")
    work = indptr[1:] - indptr[0] + _np.arange(1, nrows + 1, dtype=_np.int64)  # items up to and incl. row i
    total = int(work[-1]) if nrows else 0
    bounds = [0]
    for p in range(1, nparts):
        target = total * p // nparts
        bounds.append(int(_np.searchsorted(work, target, side="left")) if nrows else 0)
    bounds.append(nrows)
    for i in range(1, len(bounds)):  # monotone even for degenerate inputs
        bounds[i] = max(bounds[i], bounds[i - 1])
    return _np.asarray(bounds, dtype=_np.int64)


def row_block(matrix, r0, r1):
    """CSR rows [r0, r1) with the row pointer rebased to 0 (shares index / value memory)."""
    if not isinstance(matrix, (_sps.csr_matrix, _sps.csr_array)):
        raise ValueError("row_block needs a CSR matrix")
    lo, hi = int(matrix.indptr[r0]), int(matrix.indptr[r1])
    return _sps.csr_matrix((matrix.data[lo:hi], matrix.indices[lo:hi], matrix.indptr[r0:r1 + 1] - lo),
                           shape=(r1 - r0, matrix.shape[1]))


def _default_local_spmm(a_block, b):
    from .sparse_dot import dot_product_mkl
    return dot_product_mkl(a_block, b)


def sharded_dot_product(a_block, matrix_b, block_rows, src=0, group=None, local_spmm=None, gather=True):
    """C = A @ B for a row-partitioned A.

    :param a_block: this rank's CSR row block of A (rows block_rows[rank]).
    :param matrix_b: dense B (k x n, C-contiguous) on rank `src`; an array of the right shape and
        dtype (contents ignored) on the other ranks.
    :param block_rows: list with every rank's block height (sum = rows of A).
    :param gather: all-gather the output row blocks so every rank returns the full C; with
        gather=False each rank returns only its own block (outputs stay row-distributed).
    :param local_spmm: callable (a_block, b) -> dense block; defaults to the GPU dot_product_mkl.
        (Tests inject a CPU function to exercise the collective logic under gloo.)
    """
    import torch
    import torch.distributed as dist

    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    if len(block_rows) != world:
        raise ValueError("block_rows must have one entry per rank")
    if a_block.shape[0] != block_rows[rank]:
        raise ValueError("rank %d holds %d rows but block_rows says %d" % (rank, a_block.shape[0], block_rows[rank]))
    local_spmm = local_spmm or _default_local_spmm
    on_gpu = dist.get_backend(group) == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")

    b_host = _np.ascontiguousarray(matrix_b)
    b_t = torch.from_numpy(b_host).to(dev)
    dist.broadcast(b_t, src=src, group=group)  # RCCL broadcast of dense B
    b_host = b_t.cpu().numpy() if rank != src else b_host

    c_block = _np.ascontiguousarray(local_spmm(a_block, b_host))
    if not gather:
        return c_block
    n = c_block.shape[1]
    hmax = int(max(block_rows))
    padded = torch.zeros((hmax, n), dtype=torch.from_numpy(c_block).dtype, device=dev)
    if c_block.shape[0]:
        padded[:c_block.shape[0]] = torch.from_numpy(c_block).to(dev)
    gathered = torch.empty((world * hmax, n), dtype=padded.dtype, device=dev)
    dist.all_gather_into_tensor(gathered, padded, group=group)  # all-gatherv via padding
    gathered = gathered.cpu().numpy().reshape(world, hmax, n)
    return _np.concatenate([gathered[r, :block_rows[r]] for r in range(world)], axis=0)
