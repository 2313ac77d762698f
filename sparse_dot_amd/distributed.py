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


class _DeviceBlock:
    """This rank's CSR row block uploaded once (C-ABI handle over the library's own HBM copy)."""

    def __init__(self, a_block):
        from ._mi_interface import SparseHandle
        self.handle = SparseHandle.from_scipy(a_block)
        self.shape = a_block.shape
        self.dtype = _np.dtype(a_block.dtype)

    def spmm_into(self, b_t, c_t):
        """c_t := A_block @ b_t with torch CUDA tensors (row-major), zero copy, on torch's current stream."""
        import torch
        from ._mi_interface import MI, matrix_descr, _check_return_value, _type_letters, _is_double, mi_set_stream
        dbl, cplx = _is_double(self)
        letter = _type_letters[(dbl, cplx)]
        if cplx:
            raise ValueError("sharded_dot_product supports real dtypes")
        mi_set_stream(torch.cuda.current_stream().cuda_stream)
        n = b_t.shape[1]
        ret = MI.call("mi_sparse_%s_mm" % letter, 10, 1.0, self.handle.ptr, matrix_descr(), 101, b_t.data_ptr(), n, n,
                      0.0, c_t.data_ptr(), n)
        _check_return_value(ret, "mi_sparse_%s_mm" % letter)

    def free(self):
        self.handle.destroy()


def sharded_dot_product(a_block, matrix_b, block_rows, src=0, group=None, local_spmm=None, gather=True):
    """C = A @ B for a row-partitioned A.

    :param a_block: this rank's CSR row block of A (rows block_rows[rank]).
    :param matrix_b: dense B (k x n, C-contiguous) on rank `src`; an array of the right shape and
        dtype (contents ignored) on the other ranks.
    :param block_rows: list with every rank's block height (sum = rows of A).
    :param gather: all-gather the output row blocks so every rank returns the full C; with
        gather=False each rank returns only its own block (outputs stay row-distributed).
    :param local_spmm: callable (a_block, b) -> dense block.  Default: on the "nccl" (RCCL) backend B,
        the local product and the gathered C stay in HBM -- broadcast -> HIP kernel through the C ABI
        with device pointers -> all-gather -> one D2H copy of the result; on other backends the GPU
        dot_product_mkl on host arrays.  (Tests inject a CPU function to exercise the collective logic
        under gloo.)
    """
    import torch
    import torch.distributed as dist

    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    if len(block_rows) != world:
        raise ValueError("block_rows must have one entry per rank")
    if a_block.shape[0] != block_rows[rank]:
        raise ValueError("rank %d holds %d rows but block_rows says %d" % (rank, a_block.shape[0], block_rows[rank]))
    on_gpu = dist.get_backend(group) == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")

    b_host = _np.ascontiguousarray(matrix_b)
    if a_block.shape[1] != b_host.shape[0]:
        raise ValueError("Matrix alignment error: %s * %s is not valid" % (a_block.shape, b_host.shape))
    b_t = torch.from_numpy(b_host).to(dev)
    dist.broadcast(b_t, src=src, group=group)  # RCCL broadcast of dense B over xGMI
    n = b_t.shape[1]
    hmax = int(max(block_rows))

    if on_gpu and local_spmm is None:
        if a_block.dtype != b_host.dtype:
            raise ValueError("operands must share one dtype (%s & %s provided)" % (a_block.dtype, b_host.dtype))
        padded = torch.zeros((hmax, n), dtype=b_t.dtype, device=dev)
        if a_block.shape[0]:
            blk = _DeviceBlock(a_block)
            try:
                blk.spmm_into(b_t, padded[:a_block.shape[0]])
                torch.cuda.current_stream().synchronize()
            finally:
                blk.free()
    else:
        fn = local_spmm or _default_local_spmm
        b_local = b_t.cpu().numpy() if rank != src else b_host
        c_block = _np.ascontiguousarray(fn(a_block, b_local))
        padded = torch.zeros((hmax, n), dtype=torch.from_numpy(c_block).dtype, device=dev)
        if c_block.shape[0]:
            padded[:c_block.shape[0]] = torch.from_numpy(c_block).to(dev)
    if not gather:
        return padded[:block_rows[rank]].cpu().numpy()
    gathered = torch.empty((world * hmax, n), dtype=padded.dtype, device=dev)
    dist.all_gather_into_tensor(gathered, padded, group=group)  # all-gatherv via padding
    gathered = gathered.cpu().numpy().reshape(world, hmax, n)
    return _np.concatenate([gathered[r, :block_rows[r]] for r in range(world)], axis=0)
