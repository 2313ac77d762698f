"""
Row-partitioned products across the GPUs of one node (one process per GPU, torch.distributed).

The reference has no multi-device path; this is the north_star's extension of it.  The left CSR
matrix is split into contiguous, nnz-balanced row blocks (output rows are independent:
C[i, :] depends only on A[i, :] and B):

  * SpMM   -- broadcast(dense B) -> single-GPU kernel on the resident block -> all-gatherv of the
              output row blocks (`ShardedCSR.dot`, `sharded_dot_product`);
  * SpGEMM -- broadcast(B's three CSR arrays) -> local SpGEMM -> all-gather of the per-rank nnz +
              gatherv of indices / data, row pointer rebuilt from the counts
              (`sharded_sparse_dot_product`);
  * gram   -- split by OUTPUT rows (= columns of A for A^T A): every rank holds A and produces rows
              [c0, c1) of the n x n result, so no reduction is needed (`sharded_gram_matrix`).

Backend "nccl" is RCCL over xGMI on ROCm; the same code runs on "gloo" CPU tensors, which is how
the host logic is tested without GPUs.

xGMI is point-to-point (every GPU has one link to every other GPU of the node), so the collectives that move the
dense operands are built for THAT topology rather than for a ring:

  * all-gatherv (`gather_rows`, `gather_mode`): "p2p" -- ONE grouped batch of sends / receives, every rank writes its
    block straight into every peer's slice, so all world - 1 links of a GPU carry traffic at once (a ring pushes the
    whole output through one link per GPU); "bcast" -- one broadcast per rank into its slice (world back-to-back
    collectives); "padded" -- blocks padded to the tallest one and one all_gather_into_tensor.
  * broadcast of B (`broadcast_rows`, `bcast_mode`): "scatter_allgather" -- the root sends slab r (1 / world of the
    rows) to rank r over link r, then every rank forwards its slab to the others directly: the root's egress per link
    drops from |B| to 2 |B| / world; "bcast" -- a single dist.broadcast.
  * `ShardedCSR.dot_pipelined`: B and C held as column PANELS (panel-major, each panel contiguous); the broadcast of
    panel p + 2, the kernel on panel p and the all-gatherv of panel p - 1 overlap (collectives on the communicator's
    own stream, the kernel on the caller's stream).
"""
import numpy as _np
from scipy import sparse as _sps


# cost of one row of the left matrix in the SpMM, in units of one entry of a long row -- fitted to the kernels' times on the
# eight row blocks of the headline R-MAT (tools/gpu_kpart_blocks.py, round 5): a row costs 3.3 units whatever it holds (its
# output row is written, its row-end is walked), an entry of a short row 1.6 (row-owned gather, L2 hit ~0.48), an entry of a
# row of >= 64 entries (the library's spmm_kpart_min_row) 1.0 (gathered by column partition, L2 hit ~0.66)
_ROW_COST, _SHORT_ENTRY_COST, _LONG_ROW = 3.3, 1.6, 64


def partition_rows(indptr, nparts, model="kernel", dense_bytes=None):
    """Contiguous row blocks of (nearly) equal COST.  model "kernel" (default): the SpMM cost model above -- what makes the
    ranks' local products take the same time; model "items": nnz + rows, the kernel's raw work items (the rule of rounds
    1-4, under which the rank holding the many short rows of a power-law matrix finished last).  `dense_bytes` = size of the
    dense operand B when known: from 4 GiB on (BASELINE configs[4]: 17 GB) hardly any row of B is found in a cache whatever
    the row of A, an entry costs the same everywhere and a row about one entry (measured on its eight blocks: long 1.0, short
    1.08, row 1.0) -- the "items" rule is used.  Returns nparts + 1 boundaries b with b[0] = 0, b[-1] = nrows, non-decreasing."""
    if dense_bytes is not None and dense_bytes >= (1 << 32) and model == "kernel":
        model = "items"
    indptr = _np.asarray(indptr, dtype=_np.int64)
    nrows = indptr.shape[0] - 1
    if nparts < 1:
        raise ValueError("nparts must be positive")
    if model not in ("kernel", "items"):
        raise ValueError("model must be 'kernel' or 'items'")
    lens = _np.diff(indptr)
    if model == "items":
        cost = lens + 1
    else:
        cost = _ROW_COST + lens * _np.where(lens >= _LONG_ROW, 1.0, _SHORT_ENTRY_COST)
    work = _np.cumsum(cost)  # cost up to and incl. row i
    total = float(work[-1]) if nrows else 0.0
    bounds = [0]
    for p in range(1, nparts):
        target = total * p / nparts
        bounds.append(int(_np.searchsorted(work, target, side="left")) if nrows else 0)
    bounds.append(nrows)
    for i in range(1, len(bounds)):  # monotone even for degenerate inputs
        bounds[i] = max(bounds[i], bounds[i - 1])
    return _np.asarray(bounds, dtype=_np.int64)


def partition_triangle(n, nparts):
    """Row boundaries that split the upper triangle of an n x n matrix into nparts bands of (nearly)
    equal area: row i of the triangle holds n - i entries."""
    if nparts < 1:
        raise ValueError("nparts must be positive")
    i = _np.arange(n + 1, dtype=_np.float64)
    area = i * n - i * (i - 1) / 2.0  # entries in rows [0, i)
    total = area[-1] if n else 0.0
    bounds = [0]
    for p in range(1, nparts):
        bounds.append(int(_np.searchsorted(area, total * p / nparts, side="left")))
    bounds.append(n)
    for k in range(1, len(bounds)):
        bounds[k] = min(max(bounds[k], bounds[k - 1]), n)
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


def _default_local_spgemm(a_block, b):
    from .sparse_dot import dot_product_mkl
    return dot_product_mkl(a_block, b, reorder_output=True)


def _group_info(group):
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    on_gpu = dist.get_backend(group) == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
    return rank, world, on_gpu, dev


def _src_global(src, group):
    """torch.distributed.broadcast takes the GLOBAL rank of the source."""
    import torch.distributed as dist
    return dist.get_global_rank(group, src) if group is not None else src


class _Pending:
    """Outstanding communication: `wait()` orders the caller behind it (NCCL: the current stream waits, the host does
    not block; gloo: blocks)."""

    def __init__(self, works=()):
        self.works = list(works)

    def wait(self):
        for w in self.works:
            w.wait()
        self.works = []
        return self


def _p2p_batch(sends, recvs, group):
    """One grouped batch of point-to-point transfers: sends = [(tensor, group rank)], recvs likewise.  Under NCCL the
    batch is a single ncclGroup -- every transfer runs concurrently on its own link."""
    import torch.distributed as dist
    ops = [dist.P2POp(dist.irecv, t, _src_global(r, group), group) for t, r in recvs if t.numel()]
    ops += [dist.P2POp(dist.isend, t, _src_global(r, group), group) for t, r in sends if t.numel()]
    return _Pending(dist.batch_isend_irecv(ops)) if ops else _Pending()


def gather_rows_p2p(full, bounds, group=None):
    """All-gatherv as one grouped send / receive batch (returns a _Pending): this rank's block goes to every peer,
    every peer's block lands in its slice of `full`."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    b = [int(x) for x in bounds]
    mine = full[b[rank]:b[rank + 1]]
    sends = [(mine, r) for r in range(world) if r != rank]
    recvs = [(full[b[r]:b[r + 1]], r) for r in range(world) if r != rank]
    return _p2p_batch(sends, recvs, group)


def broadcast_rows(t, src=0, group=None, mode="scatter_allgather"):
    """Broadcast the (k, n) contiguous tensor `t` from group rank `src` (returns a _Pending).

    "scatter_allgather": rows are cut into world slabs; phase 1 the root sends slab r to rank r (world - 1 links in
    parallel), phase 2 every rank sends its slab to every other non-root rank (all links).  Per-link volume
    2 |t| / world instead of |t|.  "bcast": dist.broadcast."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if world == 1:
        return _Pending()
    if mode == "bcast":
        w = dist.broadcast(t, src=_src_global(src, group), group=group, async_op=True)
        return _Pending([w])
    if mode != "scatter_allgather":
        raise ValueError("bcast mode must be 'bcast' or 'scatter_allgather'")
    k = t.shape[0]
    cut = [k * r // world for r in range(world + 1)]
    slab = lambda r: t[cut[r]:cut[r + 1]]  # noqa: E731
    if rank == src:
        p1 = _p2p_batch([(slab(r), r) for r in range(world) if r != src], [], group)
    else:
        p1 = _p2p_batch([], [(slab(rank), src)], group)
    # the slab must have landed before it is forwarded.  On NCCL / RCCL wait() is a stream dependency (no host block); it is
    # issued unconditionally because only coalesced batches are guaranteed to share one communicator stream with the next
    # batch -- torch builds that issue point-to-point operations on per-peer streams would otherwise read before arrival
    p1.wait()
    if rank == src:  # the root already holds everything: it only hands out its own slab
        sends, recvs = [(slab(src), r) for r in range(world) if r != src], []
    else:
        sends = [(slab(rank), r) for r in range(world) if r != rank and r != src]
        recvs = [(slab(r), r) for r in range(world) if r != rank]  # the root's own slab comes from the root
    p2 = _p2p_batch(sends, recvs, group)
    return _Pending(p1.works + p2.works)


def gather_rows(full, bounds, group=None, mode="bcast"):
    """All-gatherv of row blocks IN PLACE: on entry rank r has written rows bounds[r]:bounds[r+1] of
    `full` (a (rows, n) tensor present on every rank); on exit every rank holds every block.

    mode "p2p": one grouped batch of sends / receives (gather_rows_p2p) -- all links at once.
    mode "bcast": one broadcast per rank directly into that rank's slice -- no padding, no staging
    copy, exactly (world - 1) / world of the output crosses the links.  mode "padded": blocks are
    copied into a buffer padded to the tallest block, one all_gather_into_tensor, and copied out --
    the classic all-gatherv emulation; nnz-balanced blocks of a skewed matrix have unequal heights, so
    it moves (and copies) more."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    heights = [int(bounds[r + 1] - bounds[r]) for r in range(world)]
    if mode == "p2p":
        gather_rows_p2p(full, bounds, group).wait()
        return full
    if mode == "bcast":
        for r in range(world):
            if heights[r]:
                dist.broadcast(full[int(bounds[r]):int(bounds[r + 1])], src=_src_global(r, group), group=group)
        return full
    if mode != "padded":
        raise ValueError("gather mode must be 'p2p', 'bcast' or 'padded'")
    hmax = max(heights) if heights else 0
    if hmax == 0:
        return full
    n = full.shape[1]
    padded = torch.zeros((hmax, n), dtype=full.dtype, device=full.device)
    padded[:heights[rank]] = full[int(bounds[rank]):int(bounds[rank + 1])]
    gathered = torch.empty((world * hmax, n), dtype=full.dtype, device=full.device)
    if full.device.type == "cpu":
        parts = list(gathered.view(world, hmax, n).unbind(0))
        dist.all_gather(parts, padded, group=group)
    else:
        dist.all_gather_into_tensor(gathered, padded, group=group)
    g3 = gathered.view(world, hmax, n)
    for r in range(world):
        if r != rank and heights[r]:
            full[int(bounds[r]):int(bounds[r + 1])] = g3[r, :heights[r]]
    return full


def pipelined_panels(local_into, b_panels, c_panels, bounds, rank, src=0, group=None, bcast_mode="scatter_allgather",
                     gather_mode="p2p", gather_group=None, depth=2):
    """The panel pipeline behind ShardedCSR.dot_pipelined (bench.py drives it with its own kernel closure):
    for every column panel p:  broadcast(panel p + depth) | local_into(B_p, C_p[rows of this rank]) | all-gatherv(C_p - 1).
    `local_into(b_panel, c_slice)` enqueues the local product on the caller's stream."""
    P = b_panels.shape[0]
    r0, r1 = int(bounds[rank]), int(bounds[rank + 1])
    ggroup = gather_group if gather_group is not None else group
    ahead = [broadcast_rows(b_panels[p], src, group, bcast_mode) for p in range(min(depth, P))]
    gathers = []
    for p in range(P):
        if p + depth < P:
            ahead.append(broadcast_rows(b_panels[p + depth], src, group, bcast_mode))
        ahead[p].wait()
        if r1 > r0:
            local_into(b_panels[p], c_panels[p][r0:r1])
        if gather_mode == "p2p":
            gathers.append(gather_rows_p2p(c_panels[p], bounds, ggroup))
        else:
            gather_rows(c_panels[p], bounds, ggroup, gather_mode)
    for g in gathers:
        g.wait()
    return c_panels


class _DeviceBlock:
    """This rank's CSR row block uploaded once (C-ABI handle over the library's own HBM copy; the
    SpMM plan is cached on the handle, so repeated products pay neither the upload nor the inspector)."""

    def __init__(self, a_block):
        import torch
        from ._mi_interface import SparseHandle, mi_set_device, mi_get_device
        # bind the library context of this thread to torch's current device BEFORE anything is allocated:
        # with one process per GPU and every GPU visible, that device is LOCAL_RANK, not 0
        self.device = torch.cuda.current_device()
        if mi_get_device() != self.device:
            mi_set_device(self.device)
        self.handle = SparseHandle.from_scipy(a_block)
        self.shape = a_block.shape
        self.dtype = _np.dtype(a_block.dtype)

    def spmm_into(self, b_t, c_t):
        """c_t := A_block @ b_t with torch CUDA tensors (row-major), zero copy, on torch's current stream."""
        import torch
        from ._mi_interface import (MI, matrix_descr, _check_return_value, _type_letters, _is_double, mi_set_stream,
                                    mi_get_stream)
        dbl, cplx = _is_double(self)
        letter = _type_letters[(dbl, cplx)]
        if cplx:
            raise ValueError("sharded_dot_product supports real dtypes")
        for t in (b_t, c_t):
            if t.device.type != "cuda" or t.device.index != self.device:
                raise ValueError("operand lives on %s but this rank's block was uploaded to cuda:%d" % (t.device, self.device))
            if not t.is_contiguous():
                raise ValueError("dense operands must be C-contiguous")
        prev = mi_get_stream()
        mi_set_stream(torch.cuda.current_stream().cuda_stream)
        try:
            n = b_t.shape[1]
            ret = MI.call("mi_sparse_%s_mm" % letter, 10, 1.0, self.handle.ptr, matrix_descr(), 101, b_t.data_ptr(), n,
                          n, 0.0, c_t.data_ptr(), n)
            _check_return_value(ret, "mi_sparse_%s_mm" % letter)
        finally:
            mi_set_stream(prev)  # the stream may belong to a `with torch.cuda.stream(...)` block that ends

    def free(self):
        self.handle.destroy()


class ShardedCSR:
    """A row-partitioned left operand that stays resident: this rank's block is uploaded once and every
    later `dot` pays only the collectives and the kernel.

    :param a_block: this rank's CSR row block (rows bounds[rank]:bounds[rank + 1] of A).
    :param bounds: world + 1 row boundaries (partition_rows), identical on every rank.
    :param local_spmm: optional callable (a_block, b ndarray) -> ndarray replacing the GPU kernel (tests
        inject a CPU function to exercise the collective logic under gloo)."""

    def __init__(self, a_block, bounds, group=None, local_spmm=None):
        self.group = group
        self.rank, self.world, self.on_gpu, self.dev = _group_info(group)
        self.bounds = _np.asarray(bounds, dtype=_np.int64)
        if self.bounds.shape[0] != self.world + 1:
            raise ValueError("bounds must have world + 1 entries")
        h = int(self.bounds[self.rank + 1] - self.bounds[self.rank])
        if a_block.shape[0] != h:
            raise ValueError("rank %d holds %d rows but the partition says %d" % (self.rank, a_block.shape[0], h))
        self.a_block = a_block
        self.rows = int(self.bounds[-1])
        self.local_spmm = local_spmm
        self._blk = None
        if self.on_gpu and local_spmm is None and h:
            self._blk = _DeviceBlock(a_block)

    def close(self):
        if self._blk is not None:
            self._blk.free()
            self._blk = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def local_into(self, b_t, c_block_t):
        """c_block_t := A_block @ b_t (this rank's rows only, no collective)."""
        import torch
        if c_block_t.shape[0] == 0:
            return
        if self._blk is not None:
            self._blk.spmm_into(b_t, c_block_t)
        else:
            fn = self.local_spmm or _default_local_spmm
            c = _np.ascontiguousarray(fn(self.a_block, b_t.cpu().numpy()))
            c_block_t.copy_(torch.from_numpy(c).to(c_block_t.device))

    def dot_pipelined(self, b_panels, c_panels=None, src=0, bcast_mode="scatter_allgather", gather_mode="p2p",
                      gather_group=None, depth=2):
        """C = A @ B with the dense operands held as column PANELS: `b_panels` is a (P, k, w) contiguous tensor
        (panel p = columns p w : (p + 1) w of B, valid on rank `src`), the result a (P, rows, w) tensor of the same
        layout (every rank receives every panel).  The broadcast of panel p + depth, the kernel on panel p and the
        all-gatherv of panel p - 1 overlap: collectives run on the communicator's stream (`gather_group`, a second
        process group over the same ranks, gives the all-gathervs their own), the kernel on the caller's."""
        import torch
        if b_panels.dim() != 3 or not b_panels.is_contiguous():
            raise ValueError("b_panels must be a contiguous (panels, k, w) tensor")
        P, k, w = b_panels.shape
        if self.a_block.shape[1] != k:
            raise ValueError("Matrix alignment error: %s * %s is not valid" % (self.a_block.shape, (k, P * w)))
        if c_panels is None:
            c_panels = torch.empty((P, self.rows, w), dtype=b_panels.dtype, device=b_panels.device)
        return pipelined_panels(self.local_into, b_panels, c_panels, self.bounds, self.rank, src=src, group=self.group,
                                bcast_mode=bcast_mode, gather_mode=gather_mode, gather_group=gather_group, depth=depth)

    def dot(self, b_t, src=0, gather=True, gather_mode="bcast", out=None, broadcast=True, bcast_mode="bcast"):
        """C = A @ B.  `b_t`: (k, n) contiguous tensor on this rank's device, holding B on rank `src`
        (contents are overwritten by the broadcast elsewhere).  Returns the full (rows, n) tensor when
        gather=True (every rank), else this rank's (h, n) block; everything stays on the device."""
        import torch
        if self.a_block.shape[1] != b_t.shape[0]:
            raise ValueError("Matrix alignment error: %s * %s is not valid" % (self.a_block.shape, tuple(b_t.shape)))
        if broadcast:
            broadcast_rows(b_t, src, self.group, bcast_mode).wait()  # RCCL broadcast of dense B
        n = b_t.shape[1]
        r0, r1 = int(self.bounds[self.rank]), int(self.bounds[self.rank + 1])
        if not gather:
            blk = out if out is not None else torch.empty((r1 - r0, n), dtype=b_t.dtype, device=b_t.device)
            self.local_into(b_t, blk)
            return blk
        full = out if out is not None else torch.empty((self.rows, n), dtype=b_t.dtype, device=b_t.device)
        self.local_into(b_t, full[r0:r1])  # the kernel writes straight into this rank's slice of C
        return gather_rows(full, self.bounds, self.group, gather_mode)


def sharded_dot_product(a_block, matrix_b, block_rows, src=0, group=None, local_spmm=None, gather=True,
                        gather_mode="bcast"):
    """C = A @ B for a row-partitioned A (one-shot convenience form of ShardedCSR).

    :param a_block: this rank's CSR row block of A (rows block_rows[rank]).
    :param matrix_b: dense B (k x n, C-contiguous) on rank `src`; an array of the right shape and
        dtype (contents ignored) on the other ranks.  A torch tensor on this rank's device is used in
        place (and the result is returned as a device tensor); a numpy array is staged and the result
        comes back as numpy.
    :param block_rows: list with every rank's block height (sum = rows of A).
    :param gather: all-gather the output row blocks so every rank returns the full C; with
        gather=False each rank returns only its own block (outputs stay row-distributed).
    :param local_spmm: callable (a_block, b) -> dense block replacing the GPU kernel (tests).
    """
    import torch
    rank, world, on_gpu, dev = _group_info(group)
    if len(block_rows) != world:
        raise ValueError("block_rows must have one entry per rank")
    if a_block.shape[0] != block_rows[rank]:
        raise ValueError("rank %d holds %d rows but block_rows says %d" % (rank, a_block.shape[0], block_rows[rank]))
    bounds = _np.concatenate([[0], _np.cumsum(_np.asarray(block_rows, dtype=_np.int64))])
    as_numpy = not isinstance(matrix_b, torch.Tensor)
    if as_numpy:
        b_host = _np.ascontiguousarray(matrix_b)
        if a_block.shape[1] != b_host.shape[0]:
            raise ValueError("Matrix alignment error: %s * %s is not valid" % (a_block.shape, b_host.shape))
        if on_gpu and local_spmm is None and a_block.dtype != b_host.dtype:
            raise ValueError("operands must share one dtype (%s & %s provided)" % (a_block.dtype, b_host.dtype))
        b_t = torch.from_numpy(b_host).to(dev)
    else:
        b_t = matrix_b
    with ShardedCSR(a_block, bounds, group=group, local_spmm=local_spmm) as sh:
        res = sh.dot(b_t, src=src, gather=gather, gather_mode=gather_mode)
        if on_gpu:
            torch.cuda.current_stream().synchronize()
    return res.cpu().numpy() if as_numpy else res


# ------------------------------------------------------------------------------------------------
# SpGEMM: C = A @ B with A row-partitioned, B replicated by broadcast, CSR blocks gathered
# ------------------------------------------------------------------------------------------------
def _bcast_csr(b, src, group, dev):
    """Broadcast a scipy CSR matrix from rank `src` (other ranks pass anything); returns scipy CSR."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank(group)
    gsrc = _src_global(src, group)
    hdr = torch.zeros(5, dtype=torch.int64, device=dev)
    if rank == src:
        if not isinstance(b, (_sps.csr_matrix, _sps.csr_array)):
            raise ValueError("the right operand must be CSR on the source rank")
        code = {"float32": 0, "float64": 1, "complex64": 2, "complex128": 3}[_np.dtype(b.dtype).name]
        hdr = torch.tensor([b.shape[0], b.shape[1], b.nnz, code, 0], dtype=torch.int64, device=dev)
    dist.broadcast(hdr, src=gsrc, group=group)
    rows, cols, nnz, code = (int(x) for x in hdr[:4].tolist())
    dt = [_np.float32, _np.float64, _np.complex64, _np.complex128][code]
    if rank == src:
        ptr = torch.from_numpy(_np.ascontiguousarray(b.indptr, dtype=_np.int64)).to(dev)
        idx = torch.from_numpy(_np.ascontiguousarray(b.indices, dtype=_np.int32)).to(dev)
        val = torch.from_numpy(_np.ascontiguousarray(b.data).view(_np.uint8)).to(dev)
    else:
        ptr = torch.empty(rows + 1, dtype=torch.int64, device=dev)
        idx = torch.empty(nnz, dtype=torch.int32, device=dev)
        val = torch.empty(nnz * _np.dtype(dt).itemsize, dtype=torch.uint8, device=dev)
    for t in (ptr, idx, val):  # the three CSR arrays
        if t.numel():
            dist.broadcast(t, src=gsrc, group=group)
    if rank == src:
        return b
    return _sps.csr_matrix((val.cpu().numpy().view(dt), idx.cpu().numpy(), ptr.cpu().numpy()), shape=(rows, cols))


def sharded_sparse_dot_product(a_block, matrix_b, block_rows, src=0, group=None, local_spgemm=None):
    """C = A @ B, both sparse: A row-partitioned, B (scipy CSR) given on rank `src` and broadcast as its
    three CSR arrays; every rank multiplies its block (two-phase hash SpGEMM on its GPU), the per-rank
    nnz are all-gathered, indices / data are gathered with one broadcast per rank straight into their
    final positions, and the row pointer is rebuilt from the per-row counts.  Every rank returns the
    full C as a scipy CSR matrix with sorted rows (the reference's result class for CSR inputs)."""
    import torch
    import torch.distributed as dist
    rank, world, on_gpu, dev = _group_info(group)
    if len(block_rows) != world:
        raise ValueError("block_rows must have one entry per rank")
    if a_block.shape[0] != block_rows[rank]:
        raise ValueError("rank %d holds %d rows but block_rows says %d" % (rank, a_block.shape[0], block_rows[rank]))
    if on_gpu:
        from ._mi_interface import mi_set_device, mi_get_device
        if mi_get_device() != torch.cuda.current_device():
            mi_set_device(torch.cuda.current_device())
    b = _bcast_csr(matrix_b, src, group, dev)
    if a_block.shape[1] != b.shape[0]:
        raise ValueError("Matrix alignment error: %s * %s is not valid" % (a_block.shape, b.shape))
    fn = local_spgemm or _default_local_spgemm
    c_blk = fn(a_block, b).tocsr() if a_block.shape[0] else _sps.csr_matrix((0, b.shape[1]), dtype=_np.result_type(a_block.dtype, b.dtype))
    # one exchange step: per-rank nnz, then the variable-length arrays
    counts = torch.zeros(world, dtype=torch.int64, device=dev)
    mine = torch.tensor([c_blk.nnz], dtype=torch.int64, device=dev)
    if dev.type == "cpu":
        parts = list(counts.unbind(0))
        parts = [p.reshape(1) for p in parts]
        dist.all_gather(parts, mine, group=group)
        counts = torch.cat(parts)
    else:
        dist.all_gather_into_tensor(counts, mine, group=group)
    counts = [int(x) for x in counts.tolist()]
    off = _np.concatenate([[0], _np.cumsum(counts)]).astype(_np.int64)
    bounds = _np.concatenate([[0], _np.cumsum(_np.asarray(block_rows, dtype=_np.int64))])
    total, rows = int(off[-1]), int(bounds[-1])
    vdt = _np.dtype(c_blk.dtype)
    rowlen = torch.zeros(rows, dtype=torch.int64, device=dev)
    idx = torch.empty(total, dtype=torch.int32, device=dev)
    val = torch.empty(total * vdt.itemsize, dtype=torch.uint8, device=dev)
    r0, r1 = int(bounds[rank]), int(bounds[rank + 1])
    if r1 > r0:
        rowlen[r0:r1] = torch.from_numpy(_np.diff(c_blk.indptr).astype(_np.int64)).to(dev)
    if c_blk.nnz:
        idx[off[rank]:off[rank + 1]] = torch.from_numpy(_np.ascontiguousarray(c_blk.indices, dtype=_np.int32)).to(dev)
        val[off[rank] * vdt.itemsize:off[rank + 1] * vdt.itemsize] = torch.from_numpy(
            _np.ascontiguousarray(c_blk.data).view(_np.uint8)).to(dev)
    for r in range(world):
        g = _src_global(r, group)
        if bounds[r + 1] > bounds[r]:
            dist.broadcast(rowlen[int(bounds[r]):int(bounds[r + 1])], src=g, group=group)
        if counts[r]:
            dist.broadcast(idx[int(off[r]):int(off[r + 1])], src=g, group=group)
            dist.broadcast(val[int(off[r]) * vdt.itemsize:int(off[r + 1]) * vdt.itemsize], src=g, group=group)
    indptr = _np.concatenate([[0], _np.cumsum(rowlen.cpu().numpy())])
    index_dt = _np.int32 if total <= _np.iinfo(_np.int32).max else _np.int64
    return _sps.csr_matrix((val.cpu().numpy().view(vdt), idx.cpu().numpy().astype(index_dt, copy=False),
                            indptr.astype(index_dt)), shape=(rows, b.shape[1]))


# ------------------------------------------------------------------------------------------------
# gram: A^T A split by output rows -- no reduction
# ------------------------------------------------------------------------------------------------
def _default_local_gram_rows(matrix, c0, c1):
    """Rows [c0, c1) of the upper triangle of A^T A as a dense (c1 - c0, n) block, on the GPU."""
    from ._gram_matrix import _gram_rows_dense
    return _gram_rows_dense(matrix, c0, c1)


def sharded_gram_matrix(matrix, group=None, gather=True, local_gram_rows=None, bounds=None):
    """Dense upper-triangular A^T A (gram_matrix_mkl(matrix, dense=True)) with the n x n OUTPUT split by
    rows across the ranks: rank r produces rows bounds[r]:bounds[r+1] (bands of equal triangle area by
    default) from the full A, which every rank holds -- a row split of A itself would leave every rank
    with a partial n x n sum and need a reduce-scatter of the dense output; splitting the output needs
    none.  gather=True all-gathers the bands (every rank returns the full n x n array, strict lower
    triangle zero); gather=False returns (band, (c0, c1))."""
    import torch
    rank, world, on_gpu, dev = _group_info(group)
    if not isinstance(matrix, (_sps.csr_matrix, _sps.csr_array)):
        raise ValueError("sharded_gram_matrix needs a CSR matrix")
    n = matrix.shape[1]
    bounds = partition_triangle(n, world) if bounds is None else _np.asarray(bounds, dtype=_np.int64)
    c0, c1 = int(bounds[rank]), int(bounds[rank + 1])
    if on_gpu:
        from ._mi_interface import mi_set_device, mi_get_device
        if mi_get_device() != torch.cuda.current_device():
            mi_set_device(torch.cuda.current_device())
    fn = local_gram_rows or _default_local_gram_rows
    band = _np.ascontiguousarray(fn(matrix, c0, c1)) if c1 > c0 else _np.zeros((0, n), dtype=matrix.dtype)
    if not gather:
        return band, (c0, c1)
    full = torch.zeros((n, n), dtype=torch.from_numpy(band).dtype, device=dev)
    if c1 > c0:
        full[c0:c1] = torch.from_numpy(band).to(dev)
    gather_rows(full, bounds, group, "bcast")
    return full.cpu().numpy()
